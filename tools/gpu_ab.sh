#!/bin/bash
# Time several library variants in one GPU call: tools/gpu_ab.sh name1 name2 ...  ("base" = libtinyntt.so)
R=${GRAFT_REPO_ROOT:-$(pwd)}
. "$R/tools/gpu_lib.sh"
for n in "$@"; do
  if [ "$n" = base ]; then L=$R/tiny_ntt_amd/lib/libtinyntt.so; else L=$R/tiny_ntt_amd/lib/libtinyntt_$n.so; fi
  echo "== $n"
  TINYNTT_LIB=$L tos 120 python $R/tests/dev/gpu_speed.py 65536 fused 2>&1 | grep -E "fused:|checksum|oracle" | tr "\n" " "; echo
done
