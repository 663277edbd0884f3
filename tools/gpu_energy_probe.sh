#!/bin/bash
# Energy per VALU instruction: run one instruction back to back on every SIMD and sample rocm-smi power / clock.
# usage: tools/gpu_energy_probe.sh "v_mad_u64_u32" "v_mul_lo_u32" ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
. "$R/tools/gpu_lib.sh"
mkdir -p $R/gpurun_out
echo "idle: $(/opt/rocm/bin/rocm-smi --showpower 2>/dev/null | grep -o 'Power (W): [0-9.]*')"
for n in "$@"; do
  F=$R/gpurun_out/energy.flag; rm -f $F
  tos 60 $R/tools/ubench/valu_rates spin "$n" 5 $F > $R/gpurun_out/energy_spin.txt 2>&1 &
  PID=$!
  for i in $(seq 1 60); do [ -f $F ] && break; sleep 0.5; done
  sleep 1.5
  S=""
  for i in 1 2 3; do
    [ -f $F ] || break
    S="$S | $(/opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|Power' | grep -oE '\([0-9]+Mhz\)|W\): [0-9.]+' | tr '\n' ' ')"
    sleep 0.8
  done
  wait $PID; [ $? -eq 9 ] && exit 9
  echo "$(cat $R/gpurun_out/energy_spin.txt | tail -1) $S"
done
