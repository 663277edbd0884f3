#!/bin/bash
# Same-box A/B of library variants on the constant-geometry sweep: tools/gpu_cg_ab.sh "variant list" name1 name2 ...  ("base" = libtinyntt.so)
# Each step writes to a file under gpurun_out/cg_ab/ (a step that hits its limit ends the whole call: gpu_lib.sh).
R=${GRAFT_REPO_ROOT:-$(pwd)}
. "$R/tools/gpu_lib.sh"
VARS=$1; shift
OUT=$R/gpurun_out/cg_ab; mkdir -p $OUT
for n in "$@"; do
  if [ "$n" = base ]; then L=$R/tiny_ntt_amd/lib/libtinyntt.so; else L=$R/tiny_ntt_amd/lib/libtinyntt_$n.so; fi
  TINYNTT_LIB=$L tos 200 python $R/tools/gpu_cg_sweep.py ${ROWS:-65536} $VARS > $OUT/$n.txt 2>&1
  echo "== $n"; grep -v amdgpu.ids $OUT/$n.txt
done
