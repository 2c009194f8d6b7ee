#!/bin/bash
# the BASELINE configurations that run on the generic kernels (patch size m = 7), one GPU, 3.3e7 DoFs
P='import sys,json; d=json.loads(sys.stdin.read()); k=d["roofline"]["kernels"]; print(d["config"]["workload"][:95], d["config"]["kernel_variant"], round(d["ms_per_step"],3), {n:round(v["avg_ms"],3) for n,v in k.items()}, "%.3e"%d["value"])'
python bench.py --no-cpu-baseline --n-overlap 2 --degree 2 --steps 5 --warmup 2 | python -c "$P"
python bench.py --no-cpu-baseline --workload kershaw --n-overlap 2 --degree 2 --steps 5 --warmup 2 | python -c "$P"
python bench.py --no-cpu-baseline --workload cart_periodic --patch vertex --weighting symm --degree 2 --steps 5 --warmup 2 | python -c "$P"
python bench.py --no-cpu-baseline --workload kershaw --n-overlap 2 --weighting ras --degree 3 --steps 5 --warmup 2 | python -c "$P"
python bench.py --no-cpu-baseline --kernel generic --steps 5 --warmup 2 | python -c "$P"
