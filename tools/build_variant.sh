#!/bin/bash
# Build an A/B variant of the library: tools/build_variant.sh <name> <extra hipcc flags...>  -> tiny_ntt_amd/lib/libtinyntt_<name>.so
set -e
NAME=$1; shift
cd "$(dirname "$0")/../tiny_ntt_amd/csrc"
mkdir -p ../lib
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function"
/opt/rocm/bin/hipcc $FLAGS "$@" -c kernels.hip -o ../lib/kernels_$NAME.o
[ -f ../lib/capi.o ] || make ../lib/capi.o
g++ -O2 -fPIC -DTN_BUILD_ID="\"variant-$NAME\"" -c build_id.cpp -o ../lib/build_id_$NAME.o      # never equal to a shipped build's id
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../lib/libtinyntt_$NAME.so ../lib/kernels_$NAME.o ../lib/capi.o ../lib/build_id_$NAME.o
echo built libtinyntt_$NAME.so
