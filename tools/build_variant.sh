#!/bin/bash
# Builds build_ab/librtcore_<name>.so: one source (render.hip unless a third argument names another, e.g. build.hip)
# compiled with extra flags, the other objects as they are.
# Usage: tools/build_variant.sh <name> "<extra hipcc flags>" [source.hip]
# A variant that changes the walker's node-phase loop (TRV_WALK_NUM / TRV_WALK_DEN, anything it inlines) can come out of
# the compiler as two nested masked loops that drop the `popped` flag of waiting lanes; such a library faults the GPU.
# Read the loop in the listing first (add -S --cuda-device-only to the compile line): one `s_andn2_b64 exec, exec` per
# node-phase loop, both s_bcnt1_i32_b64 of the phase test right under the loop header.  Delete build_ab/ when done: the
# directory travels to the GPU box with everything else.
set -e
cd "$(dirname "$0")/../raytracing_amd/csrc"
mkdir -p ../../build_ab
SRC=${3:-render.hip}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function"
/opt/rocm/bin/hipcc $FLAGS $2 -c $SRC -o ../../build_ab/${SRC%.hip}_$1.o
OBJS=""
for o in api build render post brute; do
    if [ "$o.hip" == "$SRC" ]; then OBJS="$OBJS ../../build_ab/${o}_$1.o"; else OBJS="$OBJS $o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../../build_ab/librtcore_$1.so $OBJS
rm -f ../../build_ab/${SRC%.hip}_$1.o
