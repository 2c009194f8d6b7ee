#!/bin/bash
# usage (GPU box, repo root): tools/solve_profile.sh [solve_run.py arguments] -> per-kernel totals of one whole solve
R=$PWD
O=$R/gpurun_out/solve_prof
rm -rf $O
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/kt --output-format csv -- python3 $R/tools/solve_run.py "$@" > $O/run.log 2>&1
cd $R
find $O/kt -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O/kt -name "*kernel_trace.csv" -exec cp {} $O/kernel_trace.csv \;
python - <<PY
import csv
rows = list(csv.DictReader(open("$O/kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.2f ms" % (tot / 1e6))
for r in rows[:22]:
    print("%8.3f ms %6d calls %8.1f us avg  %5.1f %%  %s" % (float(r["TotalDurationNs"]) / 1e6, int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["Percentage"]), r["Name"][:110]))
PY
cat $O/run.log
rm -rf $O/kt
