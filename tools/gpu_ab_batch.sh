#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
. "$R/tools/gpu_lib.sh"
for n in "$@"; do
  if [ "$n" = base ]; then L=$R/tiny_ntt_amd/lib/libtinyntt.so; else L=$R/tiny_ntt_amd/lib/libtinyntt_$n.so; fi
  echo "== $n"
  TINYNTT_LIB=$L tos 120 python $R/tools/gpu_batch_sweep.py cfg2 1024 4096 8192 2>&1 | grep cfg
  TINYNTT_LIB=$L tos 120 python $R/tools/gpu_batch_sweep.py cfg3 1024 65536 2>&1 | grep cfg
done
