#!/bin/bash
# z chunks of the tiled sweeps at the north-star size (128x128x96 cells): the cost model's pick against fixed values
P='import sys,json; d=json.loads(sys.stdin.read()); k=d["roofline"]["kernels"]; print("step %.4f  A %.4f  P %.4f  rows %s" % (d["ms_per_step"], k["laplace"]["avg_ms"], k["fdm"]["avg_ms"], d["config"].get("laplace_rows")))'
for z in default 3 4 5 6 7 8 10 12 16; do
  if [ $z = default ]; then unset ASM_TILED_ZCHUNKS; else export ASM_TILED_ZCHUNKS=$z; fi
  echo "zchunks=$z $(python bench.py --no-cpu-baseline --no-north-star --cells 128 128 96 --steps 10 --warmup 3 2>/dev/null | python -c "$P")"
done
