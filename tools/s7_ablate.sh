for dbg in ${S7_DBGS:-0 8 16}; do
  echo "ASM_S7_DEBUG=$dbg $(ASM_S7_DEBUG=$dbg python bench.py --no-north-star --no-cpu-baseline --n-overlap 2 --degree 2 --steps 10 --warmup 3 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['avg_launch_ms'],3))")"
done
for lz in ${S7_LZS:-}; do
  echo "ASM_S7_LZ=$lz $(ASM_S7_LZ=$lz python bench.py --no-north-star --no-cpu-baseline --n-overlap 2 --degree 2 --steps 10 --warmup 3 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['avg_launch_ms'],3))")"
done
