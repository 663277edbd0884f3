#!/bin/bash
# Time several library variants in one GPU call: tools/gpu_ab.sh name1 name2 ...  ("base" = libtinyntt.so)
# Each step runs through `tos` (gpu_lib.sh) as a plain command writing to a file, never on the left of a pipeline: a step that
# hits its limit ends the whole call (exit 9) instead of only its pipeline subshell.
R=${GRAFT_REPO_ROOT:-$(pwd)}
. "$R/tools/gpu_lib.sh"
OUT=$R/gpurun_out/ab; mkdir -p $OUT
lib_of() { if [ "$1" = base ]; then echo $R/tiny_ntt_amd/lib/libtinyntt.so; else echo $R/tiny_ntt_amd/lib/libtinyntt_$1.so; fi; }
for n in "$@"; do
  echo "== $n"
  TINYNTT_LIB=$(lib_of $n) tos 120 python $R/tests/dev/gpu_speed.py 65536 fused > $OUT/ab_$n.txt 2>&1
  grep -E "fused:|checksum|oracle" $OUT/ab_$n.txt | tr "\n" " "; echo
done
