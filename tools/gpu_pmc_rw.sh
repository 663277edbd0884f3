#!/bin/bash
# FETCH/WRITE size of a library variant: pmc_rw.sh name
R=${GRAFT_REPO_ROOT:-$(pwd)}
. "$R/tools/gpu_lib.sh"
n=$1
if [ "$n" = base ]; then L=$R/tiny_ntt_amd/lib/libtinyntt.so; else L=$R/tiny_ntt_amd/lib/libtinyntt_$n.so; fi
export TMPDIR=/tmp; cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  TINYNTT_LIB=$L tos 200 rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/rw_$n/$C -- python3 $R/tests/dev/gpu_speed.py 65536 fused > $R/gpurun_out/rw_$n.$C.log 2>&1
done
python3 $R/tools/pmc_summary.py $R/gpurun_out/rw_$n | grep polymul
