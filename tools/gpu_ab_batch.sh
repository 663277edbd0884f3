#!/bin/bash
# Batch sweeps (launch-bound region) for several library variants
# Each step runs through `tos` (gpu_lib.sh) as a plain command writing to a file, never on the left of a pipeline: a step that
# hits its limit ends the whole call (exit 9) instead of only its pipeline subshell.
R=${GRAFT_REPO_ROOT:-$(pwd)}
. "$R/tools/gpu_lib.sh"
OUT=$R/gpurun_out/ab; mkdir -p $OUT
lib_of() { if [ "$1" = base ]; then echo $R/tiny_ntt_amd/lib/libtinyntt.so; else echo $R/tiny_ntt_amd/lib/libtinyntt_$1.so; fi; }
for n in "$@"; do
  echo "== $n"
  TINYNTT_LIB=$(lib_of $n) tos 120 python $R/tools/gpu_batch_sweep.py cfg2 1024 4096 8192 > $OUT/abb2_$n.txt 2>&1
  grep cfg $OUT/abb2_$n.txt
  TINYNTT_LIB=$(lib_of $n) tos 120 python $R/tools/gpu_batch_sweep.py cfg3 1024 65536 > $OUT/abb3_$n.txt 2>&1
  grep cfg $OUT/abb3_$n.txt
done
