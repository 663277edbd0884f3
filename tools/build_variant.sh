#!/bin/bash
# Build an A/B variant of the library: tools/build_variant.sh <name> <extra hipcc flags...>  -> tiny_ntt_amd/lib/libtinyntt_<name>.so
# EVERY translation unit that sees the shared headers (kernels.hip, capi.cpp, the cg_part slices) is compiled with the same
# flags: macros in fused_core.h / modarith.h / cg_core.h also drive the host side (schedule replays, record formats).
# CG_PARTS="4 5 6" (env) limits which constant-geometry slices are rebuilt with the flags; the others are the stock objects.
set -e
NAME=$1; shift
cd "$(dirname "$0")/../tiny_ntt_amd/csrc"
mkdir -p ../lib/var_$NAME
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function"
HIPCC=/opt/rocm/bin/hipcc
PARTS=${CG_PARTS:-0 1 2 3 4 5 6}
pids=()
$HIPCC $FLAGS "$@" -c kernels.hip -o ../lib/var_$NAME/kernels.o & pids+=($!)
$HIPCC $FLAGS "$@" -x hip -c capi.cpp -o ../lib/var_$NAME/capi.o & pids+=($!)
for k in $PARTS; do $HIPCC $FLAGS "$@" -DTN_CG_PART=$k -c cg_part.hip -o ../lib/var_$NAME/cg_part$k.o & pids+=($!); done
for p in "${pids[@]}"; do wait $p; done
[ -f ../lib/multi.o ] || make ../lib/multi.o
OBJS="../lib/var_$NAME/kernels.o ../lib/var_$NAME/capi.o ../lib/multi.o"
for k in 0 1 2 3 4 5 6; do
  if [ -f ../lib/var_$NAME/cg_part$k.o ]; then OBJS="$OBJS ../lib/var_$NAME/cg_part$k.o"; else [ -f ../lib/cg_part$k.o ] || make ../lib/cg_part$k.o; OBJS="$OBJS ../lib/cg_part$k.o"; fi
done
g++ -O2 -fPIC -DTN_BUILD_ID="\"variant-$NAME\"" -c build_id.cpp -o ../lib/var_$NAME/build_id.o      # never equal to a shipped build's id
$HIPCC -shared -fPIC --offload-arch=gfx950 -o ../lib/libtinyntt_$NAME.so $OBJS ../lib/var_$NAME/build_id.o
echo built libtinyntt_$NAME.so
