#!/bin/bash
# z chunks of the tiled FDM sweep on the 78^3 Kershaw workload (7 x 11 tiles): the cost model against fixed values
P='import sys,json; d=json.loads(sys.stdin.read()); k=d["roofline"]["kernels"]; print("step %.4f  A %.4f  P %.4f" % (d["ms_per_step"], k["laplace"]["avg_ms"], k["fdm"]["avg_ms"]))'
for z in default 5 6 7 8 10 13 16 20 26; do
  if [ $z = default ]; then unset ASM_TILED_ZCHUNKS; else export ASM_TILED_ZCHUNKS=$z; fi
  echo "kershaw78 zchunks=$z $(python bench.py --no-cpu-baseline --no-north-star --workload kershaw --steps 10 --warmup 3 2>/dev/null | python -c "$P")"
done
