#!/bin/bash
# usage (on the GPU box, from the repo root): tools/pmc.sh "<counters pass 1>" "<counters pass 2>" ...
# Runs `python bench.py --steps 3 --warmup 1 --no-cpu-baseline` under rocprofv3 --pmc, one pass per
# argument (counter collection only, no tracing), and prints the mean counter value per dispatch
# of the two tiled kernels.
R=$PWD
export ASM_BENCH_LAMBDA=1.7
cd /tmp && export TMPDIR=/tmp
i=0
for c in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $c -d $R/gpurun_out/pmc_$i --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_$i.log 2>&1 || { tail -5 $R/gpurun_out/pmc_$i.log; exit 1; }
done
cd $R
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob('gpurun_out/pmc_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        name = 'laplace' if 'laplace_cart_tiled' in k else ('fdm' if 'fdm_tiled' in k else None)
        if name is None:
            continue
        a = acc[(name, r['Counter_Name'])]
        a[0] += float(r['Counter_Value']); a[1] += 1
for (k, c), (v, n) in sorted(acc.items()):
    print(f"{k:8s} {c:28s} {v / n:14.4g}  (n={n})")
PY
