#!/bin/bash
# usage (on the GPU box): tools/ab_run.sh rounds name ...  -> interleaved rounds of bench.py per variant (dealii-asm_amd/abl/lib_<name>.so),
# median ms/step, Laplace ms, FDM ms per variant (rule: A/B numbers come from interleaved rounds on ONE box)
export ASM_BENCH_LAMBDA=1.7
R=$1; shift
P='import sys,json; d=json.loads(sys.stdin.read()); k=d["roofline"]["kernels"]; print(round(d["ms_per_step"],4), round(k["laplace"]["avg_ms"],4), round(k["fdm"]["avg_ms"],4))'
for r in $(seq 1 $R); do
  for n in "$@"; do
    echo "$n $(DEALII_ASM_AMD_LIB=dealii-asm_amd/abl/lib_$n.so python bench.py --no-cpu-baseline $ASM_BENCH_ARGS 2>/dev/null | python -c "$P")"
  done
done | python3 -c '
import sys, statistics, collections
d = collections.OrderedDict()
for l in sys.stdin:
    f = l.split()
    if len(f) == 4:
        d.setdefault(f[0], []).append([float(x) for x in f[1:]])
for k, v in d.items():
    print(k, "step %.4f  laplace %.4f  fdm %.4f   (n=%d, min step %.4f)" % (statistics.median(x[0] for x in v), statistics.median(x[1] for x in v), statistics.median(x[2] for x in v), len(v), min(x[0] for x in v)))
'
