"""Copies the results of tools/profile_round.sh (gpurun_out/prof_round/) into profiles/rNN_* and rebuilds
profiles/hbm_traffic.json.  usage: python tools/collect_profiles.py 04"""
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(ROOT, "gpurun_out", "prof_round")
rnd = sys.argv[1] if len(sys.argv) > 1 else "04"
P = lambda name: os.path.join(ROOT, "profiles", "r%s_%s" % (rnd, name))


def read(name):
    return open(os.path.join(O, name)).read()


def counters(name):
    """{kernel: {counter: value}} of a tools/pmc.sh output"""
    out = {}
    for line in read(name).splitlines():
        m = re.match(r"(\S.*?)\s+(FETCH_SIZE|WRITE_SIZE)\s+([0-9.e+]+)\s+\(n=(\d+)\)", line)
        if m:
            out.setdefault(m.group(1).strip(), {})[m.group(2)] = float(m.group(3))
    return out


def total(c):  # bytes: gfx950 correction read = 2 x FETCH_SIZE x 1024 (profiles/r01_rocprofv3_pmc_hbm_traffic.txt)
    return 2 * c["FETCH_SIZE"] * 1024, c["WRITE_SIZE"] * 1024


shutil.copy(os.path.join(O, "bench_default.json"), P("bench_default.json"))
shutil.copy(os.path.join(O, "bench_others.jsonl"), P("bench_other_workloads.jsonl"))
shutil.copy(os.path.join(O, "kernel_stats.csv"), P("rocprofv3_kernel_stats.csv"))
for f in ("microbench_lds.txt", "microbench_issue.txt", "proto_lane.txt"):
    if os.path.exists(os.path.join(O, f)):
        shutil.copy(os.path.join(O, f), P(f))
pinned = read("kt_pinned.txt").strip()
line = [l for l in read("kt.log").splitlines() if l.startswith('{"metric"')][-1]
open(P("rocprofv3_kernel_stats_bench_line.txt"), "w").write(
    pinned + "\nbench line of the profiled command (python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-north-star under "
    "rocprofv3 --kernel-trace --stats; its kernel times are HIP events around every launch and agree with the AverageNs of "
    "rocprofv3_kernel_stats.csv):\n" + line + "\n")
open(P("pair_vs_tiled_ab.txt"), "w").write(
    "Wave-pair kernels forced for both sweeps (ASM_PAIR=1: the schedule model's chunk count, four waves per tile) against the tiled kernels "
    "(ASM_PAIR=0), one box, one call: ms/step {kernel: ms per launch}\n" + read("pair_vs_tiled.txt"))
open(P("pair_waves.txt"), "w").write(
    "Waves per tile (ASM_PAIR_WAVES) x z chunks (ASM_PAIR_ZCHUNKS, empty = the schedule model's pick: 10) at 80^3 cells, pair kernels for both "
    "sweeps, one box, one call:\n" + read("pair_waves.txt"))
open(P("pmc_pair_sq.txt"), "w").write(
    "SQ counters of the pair kernels at 80^3 cells (tools/pmc_pair.sh: rocprofv3 --pmc, two passes; ASM_PAIR=1, model's 10 chunks; ACTIVE / WAIT / "
    "WAVE_CYCLES in quad-cycles summed over waves)\n--- four waves per tile\n" + read("pmc_pair_sq.txt") +
    "--- two waves per tile (ASM_PAIR_WAVES=2)\n" + read("pmc_pair_sq_two_waves.txt"))
open(P("pair_stamps.txt"), "w").write(
    "Stage timers of the TWO-wave form (ASM_PAIR_WAVES=2; the timers' 32 registers do not fit beside a wave of the four-wave form), interior tile, "
    "wave 0 (13 of the 25 z columns), s_memtime cycles per layer; 'barrier 3' is no barrier any more (stage label kept):\n" + read("pair_stamps.txt"))

# ---- traffic
alg = {"80x80x80": 24.0 * 33076161, "128x128x96": 24.0 * 101320065}
sec = [("80^3, pinned picks (%s)" % pinned, "pmc_traffic_pinned.txt"),
       ("80^3, ASM_PAIR=1 (pair kernels for both sweeps, four waves, the model's 10 chunks)", "pmc_traffic_pair.txt"),
       ("128x128x96, ASM_PAIR=1", "pmc_traffic_pair_north_star.txt"),
       ("80^3, unpinned (averages include the creation-time measurement launches)", "pmc_traffic.txt"),
       ("128x128x96, unpinned", "pmc_traffic_north_star.txt")]
txt = """Round %s: HBM traffic per launch (rocprofv3 --pmc, FETCH_SIZE and WRITE_SIZE in separate counter-only passes of
`python bench.py --steps 3 --warmup 1 --no-cpu-baseline [--no-north-star | --cells 128 128 96]`, tools/pmc.sh; unit KB;
gfx950: read bytes = 2 x FETCH_SIZE x 1024, write bytes = WRITE_SIZE x 1024 - calibration: profiles/r01_rocprofv3_pmc_hbm_traffic.txt).
Algorithmic: 24 B/DoF per sweep = 793.8 MB at 80^3 cells (33 076 161 DoFs), 2 431.7 MB at 128x128x96 (101 320 065 DoFs).

A context measures its kernel candidates when it is created (tiled Laplace variants, pair kernel with up to 16 chunk counts in
two forms): those launches sit in the per-kernel averages of an unpinned run.  "pinned" = the picks of the default bench line
fixed in the environment (no measurement launches; tools/profile_round.sh).

""" % rnd
for title, fn in sec:
    txt += "== " + title + "\n"
    for l in read(fn).splitlines():
        if any(k in l for k in ("pair_sweep", "kernel2<1, 11>", "fdm_tiled")):
            txt += l[:100].rstrip() + "\n"
    txt += "\n"
pin, pair80, pairns, un80, unns = (counters(f) for _, f in sec)
rows = [("80x80x80", "fdm:wave-pair", "P^-1 pair kernel (pinned picks)", pin["pair_sweep_kernel<4, 4>"]),
        ("80x80x80", "fdm:tiled", "P^-1 tiled kernel", un80["fdm_tiled_kernel<4, false, true, true>"]),
        ("80x80x80", "laplace:tiled", "A    tiled kernel, 11 rows (pinned)", pin["laplace_cart_tiled_kernel2<1, 11>"]),
        ("80x80x80", "laplace:wave-pair", "A    pair kernel (10 chunks)", pair80["pair_sweep_kernel<1, 4>"]),
        ("128x128x96", "fdm:wave-pair", "P^-1 pair kernel", pairns["pair_sweep_kernel<4, 4>"]),
        ("128x128x96", "fdm:tiled", "P^-1 tiled kernel", unns["fdm_tiled_kernel<4, false, true, true>"]),
        ("128x128x96", "laplace:tiled", "A    tiled kernel, 11 rows", pairns["laplace_cart_tiled_kernel2<1, 11>"]),
        ("128x128x96", "laplace:wave-pair", "A    pair kernel", pairns["pair_sweep_kernel<1, 4>"])]
txt += "Summary (MB per launch = read + written; x algorithmic):\n"
tr = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
for size, key, label, c in rows:
    r, w = total(c)
    txt += "  %-11s %-40s %5.0f + %4.0f = %5.0f   %.3fx\n" % (size, label, r / 1e6, w / 1e6, (r + w) / 1e6, (r + w) / alg[size])
    e = tr.setdefault("cart_dirichlet:%s:tiled:cheb_step:1" % size, {})
    e[key] = float(round(r + w))
    e["source"] = "profiles/r%s_rocprofv3_pmc_hbm_traffic.txt" % rnd
txt += """The pair kernel reads more than the tiled one: a tile row of 8 cells + 1 DoF = 5 sectors (320 B) for 224 B owned, 33 rows for 28,
one halo layer per chunk (DESIGN.md 3.2a); with streaming stores it also wrote 320 MB instead of 286 MB (half sectors at tile
borders written separately).
"""
open(P("rocprofv3_pmc_hbm_traffic.txt"), "w").write(txt)
json.dump(tr, open(os.path.join(ROOT, "profiles", "hbm_traffic.json"), "w"), indent=2)
print(txt.split("Summary")[1])
