#!/bin/bash
# Builds build_ab/librtcore_<name>.so: render.hip compiled with extra flags, the other objects as they are.
# Usage: tools/build_variant.sh <name> "<extra hipcc flags>"
set -e
cd "$(dirname "$0")/../raytracing_amd/csrc"
mkdir -p ../../build_ab
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function"
/opt/rocm/bin/hipcc $FLAGS $2 -c render.hip -o ../../build_ab/render_$1.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../../build_ab/librtcore_$1.so api.o build.o post.o brute.o ../../build_ab/render_$1.o
rm -f ../../build_ab/render_$1.o
