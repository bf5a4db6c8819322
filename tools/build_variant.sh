#!/bin/bash
# usage: tools/build_variant.sh <name> [-D...]: variants/lib_<name>.so = libyacrd_hip.so with engine.hip compiled under the
# given defines (A/B builds for a GPU call: cp variants/lib_<name>.so yacrd_amd/lib/libyacrd_hip.so on the box)
set -e
name=$1; shift
cd "$(dirname "$0")/../yacrd_amd/csrc"
mkdir -p ../../variants/obj
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-fast-math "$@" -c -o ../../variants/obj/engine_$name.o engine.hip
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -shared -o ../../variants/lib_$name.so ../../variants/obj/engine_$name.o ../lib/obj/stream.o ../lib/obj/gpu_paf.o
echo built variants/lib_$name.so
