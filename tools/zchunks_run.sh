#!/bin/bash
# usage (GPU box): tools/zchunks_run.sh -> bench.py step and kernel times for fixed numbers of z chunks (ASM_TILED_ZCHUNKS)
export ASM_BENCH_LAMBDA=1.7
P='import sys,json; d=json.loads(sys.stdin.read()); k=d["roofline"]["kernels"]; print(round(d["ms_per_step"],4), round(k["laplace"]["avg_ms"],4), round(k["fdm"]["avg_ms"],4))'
for z in 5 8 10 16 3 5; do echo "zchunks=$z: $(ASM_TILED_ZCHUNKS=$z python bench.py --no-cpu-baseline | python -c "$P")"; done
