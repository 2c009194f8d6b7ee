#!/bin/bash
# Diagnostic build of the library with stage timers in the wave-pair kernels (-DASM_PAIR_STAMP) -> dealii-asm_amd/abl/lib_pairstamp.so
set -e
cd "$(dirname "$0")/../dealii-asm_amd"
mkdir -p abl/build_pairstamp
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -I/opt/rocm/include -DASM_PAIR_STAMP $PAIR_EXTRA"
for f in context kernels_generic kernels_vector kernels_tiled multigrid kernels_slice7 kernels_small kernels_pair; do
  /opt/rocm/bin/hipcc $F -c csrc/$f.hip -o abl/build_pairstamp/$f.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 abl/build_pairstamp/*.o -o abl/lib_pairstamp.so -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
