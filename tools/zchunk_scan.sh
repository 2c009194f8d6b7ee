#!/bin/bash
# usage (GPU box): tools/zchunk_scan.sh -> ms/step of the periodic workload with the single rank's ghost exchange going
# through RCCL (ASM_SELF_INTERFACE), per number of z chunks (0 = the library's own choice)
ms() { tail -1 | python -c "import sys,json; print('%.4f' % json.loads(sys.stdin.read())['ms_per_step'])"; }
echo "wrap-around (no exchange): $(python bench.py --workload cart_periodic --no-cpu-baseline 2>/dev/null | ms) ms"
for n in 0 2 4 6 8 11 13 16; do
  if [ $n -gt 0 ]; then export ASM_TILED_ZCHUNKS=$n; fi
  echo "zchunks $n  self-interface $(ASM_SELF_INTERFACE=1 python bench.py --workload cart_periodic --no-cpu-baseline 2>/dev/null | ms) ms"
done
