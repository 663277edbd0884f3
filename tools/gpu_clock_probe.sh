#!/bin/bash
# Sample shader clock and power with rocm-smi while the fused kernel runs: tools/gpu_clock_probe.sh name...  ("base" = libtinyntt.so)
R=${GRAFT_REPO_ROOT:-$(pwd)}
. "$R/tools/gpu_lib.sh"
mkdir -p $R/gpurun_out
for n in "$@"; do
  if [ "$n" = base ]; then L=$R/tiny_ntt_amd/lib/libtinyntt.so; else L=$R/tiny_ntt_amd/lib/libtinyntt_$n.so; fi
  echo "== $n"
  F=$R/gpurun_out/spin.flag; rm -f $F
  TINYNTT_LIB=$L tos 200 python $R/tools/gpu_spin.py $F 4 &
  PID=$!
  for i in $(seq 1 150); do [ -f $F ] && break; sleep 1; done
  for i in 1 2 3 4 5; do
    [ -f $F ] || break
    /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" | tr '\n' ' '; echo
    sleep 1
  done
  wait $PID; [ $? -eq 9 ] && exit 9
done
