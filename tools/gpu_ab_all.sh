#!/bin/bash
# Time every BASELINE shape for several library variants: tools/gpu_ab_all.sh name1 name2 ...  ("base" = libtinyntt.so)
R=${GRAFT_REPO_ROOT:-$(pwd)}
. "$R/tools/gpu_lib.sh"
for n in "$@"; do
  if [ "$n" = base ]; then L=$R/tiny_ntt_amd/lib/libtinyntt.so; else L=$R/tiny_ntt_amd/lib/libtinyntt_$n.so; fi
  echo "== $n"
  TINYNTT_LIB=$L tos 200 python $R/tools/gpu_configs.py 2>&1 | grep -E "batch|fused"
done
