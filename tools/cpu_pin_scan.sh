#!/bin/bash
# usage (GPU box): tools/cpu_pin_scan.sh -> cpu_baseline of bench.py for several placements of its 16 threads
# (ASM_ORACLE_CPUS; profiles/r03_cpu_baseline_threads.txt)
P='import sys,json; d=json.loads(sys.stdin.read()); c=d["cpu_baseline"]; print("%.3e" % c["value"])'
export ASM_BENCH_NO_PIN=1
for cpus in "" "0,4,8,12,16,20,24,28,32,36,40,44,48,52,56,60" "0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15" "64,68,72,76,80,84,88,92,96,100,104,108,112,116,120,124" "0,2,4,6,8,10,12,14,16,18,20,22,24,26,28,30"; do
  if [ -z "$cpus" ]; then unset ASM_ORACLE_CPUS; else export ASM_ORACLE_CPUS=$cpus; fi
  echo "cpus=[$cpus] $(python bench.py --no-north-star 2>/dev/null | python -c "$P")"
done
