#!/bin/bash
# Second set of PMC passes (stall attribution) for the fused kernel. usage: tools/gpu_pmc2.sh <tag>
set -o pipefail
TAG=${1:-pmc2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
. "$R/tools/gpu_lib.sh"
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for SET in "SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_IFETCH_LEVEL SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INSTS_BRANCH" \
           "SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL" \
           "SQ_ACTIVE_INST_VALU2 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_IOPS SQ_LEVEL_WAVES SQ_INSTS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR" ; do
  i=$((i+1))
  tos 200 rocprofv3 --pmc $SET --output-format csv -d $OUT/pass$i -- python3 $R/tests/dev/gpu_speed.py 65536 fused > $OUT/pass$i.log 2>&1
  echo "pass $i ($SET) rc=$?"
done
python3 $R/tools/pmc_summary.py $OUT | tee $OUT/summary.txt
