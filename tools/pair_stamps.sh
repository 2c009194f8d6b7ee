#!/bin/bash
# Diagnostic build of the library with stage timers in the wave-pair kernels (-DASM_PAIR_STAMP) -> dealii-asm_amd/abl/lib_pairstamp.so
# (PAIR_EXTRA: further -D flags, PAIR_OUT: another name for the library)
set -e
cd "$(dirname "$0")/../dealii-asm_amd"
OUT=${PAIR_OUT:-lib_pairstamp.so}
B=abl/build_${OUT%.so}
mkdir -p $B
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -I/opt/rocm/include -DASM_PAIR_STAMP $PAIR_EXTRA"
for f in context kernels_generic kernels_vector kernels_tiled multigrid kernels_slice7 kernels_small kernels_pair; do
  /opt/rocm/bin/hipcc $F -c csrc/$f.hip -o $B/$f.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $B/*.o -o abl/$OUT -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
