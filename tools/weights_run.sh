#!/bin/bash
# usage (GPU box): tools/weights_run.sh -> bench.py step and kernel times for the weightings post / none / symm
P='import sys,json; d=json.loads(sys.stdin.read()); k=d["roofline"]["kernels"]; print(d["config"]["workload"][60:], round(d["ms_per_step"],4), {n:round(v["avg_ms"],4) for n,v in k.items()})'
export ASM_BENCH_LAMBDA=1.7
for w in post none symm; do python bench.py --no-cpu-baseline --weighting $w | python -c "$P"; done
DEALII_ASM_AMD_LIB=dealii-asm_amd/abl/lib_a128.so python bench.py --no-cpu-baseline | python -c "$P"
