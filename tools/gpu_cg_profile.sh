#!/bin/bash
# rocprofv3 passes for the CG (stage-sweep) kernels: timing + LDS bank-conflict counters.  usage: tools/gpu_cg_profile.sh <tag>
TAG=${1:-cg}
R=${GRAFT_REPO_ROOT:-$(pwd)}
. "$R/tools/gpu_lib.sh"
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
tos 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/gpu_cg_sweep.py ${ROWS:-65536} > $OUT/sweep.txt 2>&1; echo "trace rc=$?"
tos 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_INSTS_VALU --output-format csv -d $OUT/pass1 -- python3 $R/tools/gpu_cg_sweep.py ${ROWS:-65536} > $OUT/pmc1.txt 2>&1; echo "pmc1 rc=$?"
tos 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pass2 -- python3 $R/tools/gpu_cg_sweep.py ${ROWS:-65536} > $OUT/pmc2.txt 2>&1; echo "pmc2 rc=$?"
tos 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pass3 -- python3 $R/tools/gpu_cg_sweep.py ${ROWS:-65536} > $OUT/pmc3.txt 2>&1; echo "pmc3 rc=$?"
grep -v amdgpu.ids $OUT/sweep.txt
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -r cat | head -16
python3 $R/tools/pmc_summary.py $OUT | grep cg_kernel | tee $OUT/summary.txt
