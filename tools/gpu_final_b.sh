#!/bin/bash
# Second half of tools/gpu_final.sh (one gpurun call): the constant-geometry sweep (timing + PMC, 11 points), its phase stamps and the
# fused kernel's in-kernel clock (diagnostic builds, if present), clock / power of the CG kernels, latency, two seeds of the parity fuzz.
TAG=${1:-final}
R=${GRAFT_REPO_ROOT:-$(pwd)}
. "$R/tools/gpu_lib.sh"
cd $R
mkdir -p gpurun_out/$TAG
sub tools/gpu_cg_profile.sh ${TAG}_cg > gpurun_out/${TAG}_cg.log 2>&1; python3 tools/make_cg_table.py gpurun_out/${TAG}_cg > gpurun_out/$TAG/cg_table.txt 2>&1; cat gpurun_out/$TAG/cg_table.txt
if [ -f tiny_ntt_amd/lib/libtinyntt_stamps.so ]; then TINYNTT_LIB=$R/tiny_ntt_amd/lib/libtinyntt_stamps.so tos 200 python tools/gpu_cg_stamps.py 65536 cg8_padded cg8_swizzled cg4_padded > gpurun_out/$TAG/cg_stamps.txt 2>&1; grep -v amdgpu gpurun_out/$TAG/cg_stamps.txt; fi
if [ -f tiny_ntt_amd/lib/libtinyntt_fstamps.so ]; then TINYNTT_LIB=$R/tiny_ntt_amd/lib/libtinyntt_fstamps.so tos 200 python tools/gpu_fused_clock.py > gpurun_out/$TAG/fused_clock.txt 2>&1; grep -v amdgpu gpurun_out/$TAG/fused_clock.txt; fi
for v in cg8_padded cg4_padded cg8 cg; do echo "#### $v"; SPIN_VARIANT=$v sub tools/gpu_clock_probe.sh base; done > gpurun_out/$TAG/cg_clock.log 2>&1; grep -c sclk gpurun_out/$TAG/cg_clock.log
tos 200 python tools/gpu_latency.py > gpurun_out/$TAG/latency.txt 2>&1
for seed in ${FUZZ_SEEDS:-5 6}; do tos 330 python tests/dev/gpu_fuzz.py $seed 300 > gpurun_out/$TAG/fuzz_$seed.txt 2>&1; tail -1 gpurun_out/$TAG/fuzz_$seed.txt; done
