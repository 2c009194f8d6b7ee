#!/bin/bash
# usage (GPU box, repo root): tools/rccl_self_profile.sh -> gpurun_out/rccl_self/: bench line of the periodic workload
# with wrap-around and with the single rank's ghost exchange going through RCCL (ASM_SELF_INTERFACE), and the kernel
# stats of the latter (the RCCL device kernels appear in it)
R=$PWD
O=$R/gpurun_out/rccl_self
rm -rf $O
mkdir -p $O
python bench.py --workload cart_periodic --no-cpu-baseline > $O/bench_wrap.json
ASM_SELF_INTERFACE=1 python bench.py --workload cart_periodic --no-cpu-baseline > $O/bench_self.json
cd /tmp && export TMPDIR=/tmp ASM_SELF_INTERFACE=1
rocprofv3 --kernel-trace --stats -d $O/kt --output-format csv -- python3 $R/bench.py --workload cart_periodic --no-cpu-baseline --steps 20 --warmup 5 > $O/kt.log 2>&1
cd $R
find $O/kt -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O/kt -name "*kernel_trace.csv" -exec cp {} $O/kernel_trace.csv \;
rm -rf $O/kt
python - <<PY
import json, csv
for n in ("wrap", "self"):
    d = json.loads(open("$O/bench_%s.json" % n).read().strip().splitlines()[-1])
    print(n, "ms_per_step %.4f" % d["ms_per_step"], "DoF/s %.3e" % d["value"], d["config"].get("rehearsal", ""))
for r in list(csv.DictReader(open("$O/kernel_stats.csv")))[:14]:
    print("%9.3f ms %6s calls %8.1f us avg  %s" % (float(r["TotalDurationNs"]) / 1e6, r["Calls"], float(r["AverageNs"]) / 1e3, r["Name"][:100]))
PY
