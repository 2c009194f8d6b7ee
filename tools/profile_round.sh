#!/bin/bash
# usage (GPU box, repo root): tools/profile_round.sh  -> everything under gpurun_out/prof_round/
# bench lines of the default and the other workloads, rocprofv3 kernel trace + stats of the default bench command,
# HBM traffic (FETCH_SIZE / WRITE_SIZE in separate counter-only passes)
set -e
R=$PWD
O=$R/gpurun_out/prof_round
rm -rf $O
mkdir -p $O
python bench.py > $O/bench_default.json
for w in "--workload cart_periodic" "--workload kershaw" "--degree 3" "--cells 128 128 96" "--n-overlap 2 --degree 2" \
         "--workload cart_periodic --patch vertex --weighting symm --degree 2" "--workload kershaw --n-overlap 2 --degree 2"; do
  python bench.py --no-cpu-baseline --no-north-star $w >> $O/bench_others.jsonl
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/kt --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-north-star > $O/kt.log 2>&1
cd $R
PMC_BENCH_ARGS="--no-north-star" tools/pmc.sh "FETCH_SIZE" "WRITE_SIZE" > $O/pmc_traffic.txt
PMC_BENCH_ARGS="--no-north-star --cells 128 128 96" tools/pmc.sh "FETCH_SIZE" "WRITE_SIZE" > $O/pmc_traffic_north_star.txt
find $O/kt -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/kt gpurun_out/pmc_1 gpurun_out/pmc_2
