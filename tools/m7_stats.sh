#!/bin/bash
# per-kernel times (rocprofv3 --kernel-trace --stats) of the 7^3-patch FDM sweep: Cartesian n_overlap = 2 by default
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/kt7
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt7 --output-format csv -- python3 $R/bench.py --no-cpu-baseline --n-overlap 2 --degree 2 --steps 5 --warmup 2 "$@" > $R/gpurun_out/kt7.log 2>/dev/null
cd $R
tail -1 gpurun_out/kt7.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['roofline']['kernels']
print(d['config']['workload'], 'ms/step %.3f' % d['ms_per_step'], {n: round(v['avg_ms'], 3) for n, v in k.items()})"
python3 - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/kt7/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Name']
        if "fdm_" in n or "s7_init" in n or 'laplace_cart_tiled_kernel<1>' in n or 'laplace_generic' in n:
            print('  %-44s calls %4s avg %8.1f us' % (n.split('(anonymous namespace)::')[1][:44], r['Calls'], float(r['AverageNs']) / 1e3))
PY
