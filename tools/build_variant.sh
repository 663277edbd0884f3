#!/bin/bash
# Build an A/B variant of the library: tools/build_variant.sh <name> <extra hipcc flags...>  -> tiny_ntt_amd/lib/libtinyntt_<name>.so
set -e
NAME=$1; shift
cd "$(dirname "$0")/../tiny_ntt_amd/csrc"
mkdir -p ../lib
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function"
/opt/rocm/bin/hipcc $FLAGS "$@" -c kernels.hip -o ../lib/kernels_$NAME.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../lib/libtinyntt_$NAME.so ../lib/kernels_$NAME.o ../lib/capi.o
echo built libtinyntt_$NAME.so
