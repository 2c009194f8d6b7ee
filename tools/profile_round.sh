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
# (the kernels the default line's context picked, pinned: the context's own measurement launches at creation - a few
#  hundred launches of the pair kernel with other chunk counts - would otherwise sit in the same per-kernel averages)
eval $(python3 - $O/bench_default.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
p = d["config"]["pair_sweeps"]
v = {(False, False): 0, (True, True): 1, (False, True): 2, (True, False): 3}[(p["laplace"], p["fdm"])]
print("export ASM_PAIR=%d ASM_PAIR_ZCHUNKS=%d ASM_PAIR_WAVES=%d ASM_TILED_ROWS=%d" % (v, p["z_chunks"], p["waves"], d["config"]["laplace_rows"]))
PY
)
echo "pinned: ASM_PAIR=$ASM_PAIR ASM_PAIR_ZCHUNKS=$ASM_PAIR_ZCHUNKS ASM_PAIR_WAVES=$ASM_PAIR_WAVES ASM_TILED_ROWS=$ASM_TILED_ROWS" > $O/kt_pinned.txt
rocprofv3 --kernel-trace --stats -d $O/kt --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-north-star > $O/kt.log 2>&1
cd $R
PMC_BENCH_ARGS="--no-north-star" tools/pmc.sh "FETCH_SIZE" "WRITE_SIZE" > $O/pmc_traffic_pinned.txt
unset ASM_PAIR ASM_PAIR_ZCHUNKS ASM_PAIR_WAVES ASM_TILED_ROWS
cd /tmp
cd $R
PMC_BENCH_ARGS="--no-north-star" tools/pmc.sh "FETCH_SIZE" "WRITE_SIZE" > $O/pmc_traffic.txt
PMC_BENCH_ARGS="--no-north-star --cells 128 128 96" tools/pmc.sh "FETCH_SIZE" "WRITE_SIZE" > $O/pmc_traffic_north_star.txt
# round 4: the wave-pair kernels forced (a context otherwise takes the faster family per sweep), and their stage timers
ASM_PAIR=1 PMC_BENCH_ARGS="--no-north-star" tools/pmc.sh "FETCH_SIZE" "WRITE_SIZE" > $O/pmc_traffic_pair.txt
ASM_PAIR=1 PMC_BENCH_ARGS="--no-north-star --cells 128 128 96" tools/pmc.sh "FETCH_SIZE" "WRITE_SIZE" > $O/pmc_traffic_pair_north_star.txt
for p in 0 1; do for w in "" "--workload cart_periodic" "--degree 3" "--cells 128 128 96"; do
  echo "ASM_PAIR=$p $w" >> $O/pair_vs_tiled.txt
  ASM_PAIR=$p tools/bench_lib.sh dealii-asm_amd/libdealii_asm_amd.so --no-cpu-baseline --no-north-star $w >> $O/pair_vs_tiled.txt
done; done
# (stage timers: the two-wave form only - the timers' registers do not fit beside a wave of four per tile)
if [ -f dealii-asm_amd/abl/lib_pairstamp.so ]; then ASM_PAIR=1 ASM_PAIR_WAVES=2 python tools/pair_stamps.py > $O/pair_stamps.txt 2>&1 || true; ASM_PAIR=1 ASM_PAIR_WAVES=2 python tools/pair_stamps.py 128 128 96 >> $O/pair_stamps.txt 2>&1 || true; fi
# waves per tile: four against two, model's chunk count and measured
for w in 4 2; do for z in "" 3 6 7 10; do
  echo "ASM_PAIR=1 ASM_PAIR_WAVES=$w ASM_PAIR_ZCHUNKS=$z" >> $O/pair_waves.txt
  ASM_PAIR=1 ASM_PAIR_WAVES=$w ASM_PAIR_ZCHUNKS=$z tools/bench_lib.sh dealii-asm_amd/libdealii_asm_amd.so --no-cpu-baseline --no-north-star >> $O/pair_waves.txt
done; done
bash tools/pmc_pair.sh > $O/pmc_pair_sq.txt 2>&1 || true
ASM_PAIR_WAVES=2 bash tools/pmc_pair.sh > $O/pmc_pair_sq_two_waves.txt 2>&1 || true
(cd tools && hipcc -O3 -std=c++17 --offload-arch=gfx950 -o microbench_lds microbench_lds.hip 2>/dev/null && ./microbench_lds > $O/microbench_lds.txt)
(cd tools && hipcc -O3 --offload-arch=gfx950 -o microbench_issue microbench_issue.hip 2>/dev/null; ./microbench_issue > $O/microbench_issue.txt; hipcc -O3 --offload-arch=gfx950 -o proto_lane proto_lane.hip 2>/dev/null; for a in "80 80 80 7" "80 80 80 10" "128 128 96 4"; do ./proto_lane $a | grep "cells\|check\|stamps\|lane sweep" >> $O/proto_lane.txt; done)
find $O/kt -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/kt gpurun_out/pmc_1 gpurun_out/pmc_2
