#!/bin/bash
# 7^3-patch FDM sweep: accumulate form (default) against the tmp + sum-kernel form (ASM_S7_TMP=1), bench.py's per-sweep time
P='import sys,json; d=json.loads(sys.stdin.read()); print("%-100s fdm %.3f ms  step %.3f ms" % (d["config"]["workload"], d["roofline"]["kernels"]["fdm"]["avg_ms"], d["ms_per_step"]))'
for w in "--workload cart_dirichlet --n-overlap 2" "--workload cart_periodic --patch vertex --weighting symm" "--workload kershaw --n-overlap 2"; do
  for tmp in 0 1; do
    if [ $tmp = 1 ]; then export ASM_S7_TMP=1; else unset ASM_S7_TMP; fi
    echo -n "ASM_S7_TMP=$tmp  "
    python bench.py --no-north-star --no-cpu-baseline --degree 2 --steps 10 --warmup 3 $w 2>/dev/null | python -c "$P"
  done
done
