#!/bin/bash
# usage: scripts/build_variant.sh <name> [extra hipcc flags...] — an A/B build of the render kernels: kernels.hip compiled
# with the extra flags into ray_tracing_amd/libmirt_<name>.so (everything else from the objects of the last `make`).
set -e
NAME=$1; shift
cd "$(dirname "$0")/../ray_tracing_amd/csrc"
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-function"
hipcc --offload-arch=gfx950 $FLAGS "$@" -c kernels.hip -o build/kernels_$NAME.o
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmirt_$NAME.so build/kernels_$NAME.o build/wavefront.o build/mirt_api.o build/comm.o build/host_capi.o -lz -ldl -pthread
echo built ray_tracing_amd/libmirt_$NAME.so
