#!/bin/bash
# kernel times for Chebyshev degrees 1..4 (the three-term update variant of the FDM sweep appears from degree 3 on)
P='import sys,json; d=json.loads(sys.stdin.read()); k=d["roofline"]["kernels"]; print(d["config"]["workload"][32:70], round(d["ms_per_step"],4), {n:round(v["avg_ms"],4) for n,v in k.items()}, "%.3e"%d["value"])'
for k in 1 2 3 4; do python bench.py --no-cpu-baseline --degree $k | python -c "$P"; done
