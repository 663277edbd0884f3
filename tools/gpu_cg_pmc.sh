#!/bin/bash
# Issue / stall / LDS counters of chosen constant-geometry variants in separate rocprofv3 --pmc passes.
# usage: tools/gpu_cg_pmc.sh <tag> variant...
TAG=${1:-cgpmc}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
. "$R/tools/gpu_lib.sh"
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE" ; do
  i=$((i+1))
  tos 200 rocprofv3 --pmc $SET --output-format csv -d $OUT/pass$i -- python3 $R/tools/gpu_cg_sweep.py ${ROWS:-65536} "$@" > $OUT/pass$i.log 2>&1
  echo "pass $i rc=$?"
done
python3 $R/tools/pmc_summary.py $OUT | tee $OUT/summary.txt
