#!/bin/bash
# Collect PMC counters for the fused kernel in separate rocprofv3 passes (no trace domains combined with --pmc).
# usage: tools/gpu_pmc.sh <tag> [rows]
set -o pipefail
TAG=${1:-pmc}
ROWS=${2:-65536}
R=${GRAFT_REPO_ROOT:-$(pwd)}
. "$R/tools/gpu_lib.sh"
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
if [ ! -f $OUT/counters_list.txt ]; then rocprofv3 -L > $OUT/counters_list.txt 2>&1; fi
i=0
for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum" ; do
  i=$((i+1))
  tos 200 rocprofv3 --pmc $SET --output-format csv -d $OUT/pass$i -- python3 $R/tests/dev/gpu_speed.py $ROWS fused > $OUT/pass$i.log 2>&1
  echo "pass $i ($SET) rc=$?"
done
python3 $R/tools/pmc_summary.py $OUT | tee $OUT/summary.txt
