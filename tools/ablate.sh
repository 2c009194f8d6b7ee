#!/bin/bash
# Builds diagnostic / experimental variants of the library into dealii-asm_amd/abl/:
#   tools/ablate.sh name:"-DFLAG ..." ...      (e.g.  a1:-DASM_ABLATE=1  pt:-DASM_PREFETCH_TOP)
# run them with tools/ablate_run.sh name ... on the GPU box.
set -e
cd "$(dirname "$0")/../dealii-asm_amd"
mkdir -p abl
for spec in "$@"; do
  n="${spec%%:*}"; d="${spec#*:}"
  mkdir -p build_abl_$n
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -I/opt/rocm/include $d \
     -c csrc/kernels_tiled.hip -o build_abl_$n/kernels_tiled.o 2>/dev/null &
done
wait
for spec in "$@"; do
  n="${spec%%:*}"
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 build/context.o build/kernels_generic.o build/kernels_vector.o \
     build_abl_$n/kernels_tiled.o -o abl/lib_$n.so -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
done
