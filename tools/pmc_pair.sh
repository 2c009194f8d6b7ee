# SQ counters of the wave-pair kernels (instruction mix, busy and wait cycles): two passes -> gpurun_out/r04_pmc_v{1,2}
cd /tmp && export TMPDIR=/tmp
export ASM_PAIR=1
rm -rf $GRAFT_REPO_ROOT/gpurun_out/r04_pmc_v1 $GRAFT_REPO_ROOT/gpurun_out/r04_pmc_v2
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY -d $GRAFT_REPO_ROOT/gpurun_out/r04_pmc_v1 --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-north-star --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS -d $GRAFT_REPO_ROOT/gpurun_out/r04_pmc_v2 --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-north-star --no-cpu-baseline > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, os, collections
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out"
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(root + "/r04_pmc_v*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "pair_sweep" not in k: continue
        k = k[k.index("pair_sweep"):][:28]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        print("   %-24s %14.0f per launch" % (c, acc[k][c] / cnt[k][c]))
PY
