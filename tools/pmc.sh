#!/bin/bash
# usage (on the GPU box, from the repo root): tools/pmc.sh "<counters pass 1>" "<counters pass 2>" ...
# Runs `python bench.py --steps 3 --warmup 1 --no-cpu-baseline` under rocprofv3 --pmc, one pass per
# argument (counter collection only, no tracing), and prints the mean counter value per dispatch
# of every sweep kernel variant that ran (ASM_TILED_ROWS=0 keeps the context from launching the other
# Laplace variants while it tunes).  PMC_BENCH_ARGS="--n-overlap 2 --degree 2" selects another workload.
R=$PWD
rm -rf $R/gpurun_out/pmc_[0-9]*   # (every call reports its own passes only)
export ASM_BENCH_LAMBDA=1.7
cd /tmp && export TMPDIR=/tmp
i=0
for c in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $c -d $R/gpurun_out/pmc_$i --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline $PMC_BENCH_ARGS > $R/gpurun_out/pmc_$i.log 2>&1 || { tail -5 $R/gpurun_out/pmc_$i.log; exit 1; }
done
cd $R
python3 - <<'PY'
import csv, glob, collections, re
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob('gpurun_out/pmc_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if not any(t in k for t in ('laplace_cart_tiled', 'fdm_tiled', 'laplace_generic', 'fdm_generic', 'fdm_gather', 'fdm_slice7', 's7_init', 'pair_sweep')):
            continue
        # kernel name with its template arguments, without the parameter list (the context's tuning launches of
        # other kernel variants stay separate rows)
        name = re.search(r'(\w+(<[^>]*>)?)\(asmamd::DevGrid', k).group(1)
        a = acc[(name, r['Counter_Name'])]
        a[0] += float(r['Counter_Value']); a[1] += 1
for (k, c), (v, n) in sorted(acc.items()):
    print(f"{k:44s} {c:28s} {v / n:14.4g}  (n={n})")
PY
