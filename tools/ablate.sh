#!/bin/bash
# Builds diagnostic / experimental variants of the library into dealii-asm_amd/abl/:
#   tools/ablate.sh name:"-DFLAG ..." ...      (e.g.  a1:-DASM_ABLATE=1  stamp:-DASM_STAMP)
# run them with tools/ablate_run.sh / tools/ab_run.sh name ... on the GPU box.  A variant with -DASM_STAMP gets a
# context.o built with the same flag (it exports the stamp read-out).
set -e
cd "$(dirname "$0")/../dealii-asm_amd"
mkdir -p abl
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -I/opt/rocm/include"
for spec in "$@"; do
  n="${spec%%:*}"; d="${spec#*:}"
  mkdir -p build_abl_$n
  /opt/rocm/bin/hipcc $FL $d -c csrc/kernels_tiled.hip -o build_abl_$n/kernels_tiled.o 2>/dev/null &
  case "$d" in *ASM_STAMP*) /opt/rocm/bin/hipcc $FL -DASM_STAMP -c csrc/context.hip -o build_abl_$n/context.o 2>/dev/null & ;; esac
done
wait
for spec in "$@"; do
  n="${spec%%:*}"
  ctx=build/context.o; [ -f build_abl_$n/context.o ] && ctx=build_abl_$n/context.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $ctx $(ls build/*.o | grep -v "kernels_tiled.o\|context.o") \
     build_abl_$n/kernels_tiled.o -o abl/lib_$n.so -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
done
