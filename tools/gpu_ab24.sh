#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
. "$R/tools/gpu_lib.sh"
for n in "$@"; do
  if [ "$n" = base ]; then L=$R/tiny_ntt_amd/lib/libtinyntt.so; else L=$R/tiny_ntt_amd/lib/libtinyntt_$n.so; fi
  echo "== $n"; TINYNTT_LIB=$L tos 120 python $R/tools/gpu_speed24.py 2>&1 | grep -E "bit B="
done
