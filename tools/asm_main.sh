#!/bin/bash
# Developer tool: compile only the benchmark-shape product kernel (-DTN_ONLY_MAIN) with extra flags and print its
# register/scratch usage and the static instruction histogram.  usage: tools/asm_main.sh [extra hipcc flags...]
cd "$(dirname "$0")/../tiny_ntt_amd/csrc"
D=$(mktemp -d /tmp/tnasm.XXXX)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -DTN_ONLY_MAIN "$@" -save-temps=obj \
  -Rpass-analysis=kernel-resource-usage -c kernels.hip -o $D/k.o 2>&1 | grep -A5 "polymul_fused_kernelImLi12ELi3ELb1" | grep -E "SGPRs:|VGPRs:|Scratch|Occupancy"
python3 ../../tools/asm_hist.py $D/kernels-hip-amdgcn-amd-amdhsa-gfx950.s polymul_fused_kernelImLi12ELi3ELb1 | head -${HIST_LINES:-12}
echo "asm: $D/kernels-hip-amdgcn-amd-amdhsa-gfx950.s"
