#!/bin/bash
# usage (on the GPU box): tools/ablate_run.sh name ...   -> ms/step, laplace ms, fdm ms per variant
export ASM_BENCH_LAMBDA=1.7
P='import sys,json; d=json.loads(sys.stdin.read()); k=d["roofline"]["kernels"]; print(round(d["ms_per_step"],4), round(k["laplace"]["avg_ms"],4), round(k["fdm"]["avg_ms"],4))'
echo "full: $(python bench.py --no-cpu-baseline $ASM_BENCH_ARGS | python -c "$P")"
for n in "$@"; do
  echo "$n: $(DEALII_ASM_AMD_LIB=dealii-asm_amd/abl/lib_$n.so python bench.py --no-cpu-baseline $ASM_BENCH_ARGS | python -c "$P")"
done
echo "full: $(python bench.py --no-cpu-baseline $ASM_BENCH_ARGS | python -c "$P")"
