#!/bin/bash
# chunk-count model of the tiled sweeps: ASM_TILED_TAIL=0.5 (default) against neighbours, on the three Cartesian bench meshes
P='import sys,json; d=json.loads(sys.stdin.read()); k=d["roofline"]["kernels"]; print("step %.4f  A %.4f  P %.4f  rows %s" % (d["ms_per_step"], k["laplace"]["avg_ms"], k["fdm"]["avg_ms"], d["config"].get("laplace_rows")))'
for w in "--cells 80 80 80" "--cells 128 128 96" "--workload cart_periodic" "--cells 64 64 64" "--cells 96 96 96" "--cells 160 160 128"; do
  for t in 0 0.5 1.0; do
    echo "$w tail=$t $(ASM_TILED_TAIL=$t python bench.py --no-cpu-baseline --no-north-star $w --steps 10 --warmup 3 2>/dev/null | python -c "$P")"
  done
done
