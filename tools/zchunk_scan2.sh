#!/bin/bash
# usage (GPU box): tools/zchunk_scan2.sh lib [lib ...] : kernel ms for several z chunkings (prologue vs per-layer cost)
export ASM_BENCH_LAMBDA=1.7 ASM_TILED_ROWS=0
P='import sys,json; d=json.loads(sys.stdin.read()); k=d["roofline"]["kernels"]; print(round(d["ms_per_step"],4), round(k["laplace"]["avg_ms"],4), round(k["fdm"]["avg_ms"],4))'
for n in "$@"; do
  for z in 4 8 16 27; do
    echo "$n zchunks=$z: $(ASM_TILED_ZCHUNKS=$z DEALII_ASM_AMD_LIB=dealii-asm_amd/abl/lib_$n.so python bench.py --no-cpu-baseline 2>/dev/null | python -c "$P")"
  done
done
