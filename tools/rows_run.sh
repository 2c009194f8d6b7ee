#!/bin/bash
# A/B of the tiled Laplace sweep on one box: first generation (0) vs second generation with 9..11 cell rows
# per workgroup, and the context's own choice (auto)
export ASM_BENCH_LAMBDA=1.7
for v in 0 9 10 11 auto; do
  if [ $v = auto ]; then unset ASM_TILED_ROWS; else export ASM_TILED_ROWS=$v; fi
  echo "== rows $v $*"
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['roofline']['kernels']
print('ms/step %.4f  laplace %.4f ms  fdm %.4f ms' % (d['ms_per_step'], k['laplace']['avg_ms'], k['fdm']['avg_ms']))"
done
