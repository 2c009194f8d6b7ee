# L2 (TCC) hit rate, requests and LDS conflicts of the pinned default kernels: rocprofv3 --pmc, three passes
cd /tmp && export TMPDIR=/tmp
export ASM_PAIR=2 ASM_PAIR_ZCHUNKS=${PAIR_Z:-7} ASM_PAIR_WAVES=4 ASM_TILED_ROWS=11
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/r04_pmc_c*
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum -d $R/gpurun_out/r04_pmc_c1 --output-format csv -- python3 $R/bench.py --steps 4 --warmup 2 --no-north-star --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_LDS -d $R/gpurun_out/r04_pmc_c2 --output-format csv -- python3 $R/bench.py --steps 4 --warmup 2 --no-north-star --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCP_TCC_READ_REQ_sum -d $R/gpurun_out/r04_pmc_c3 --output-format csv -- python3 $R/bench.py --steps 4 --warmup 2 --no-north-star --no-cpu-baseline > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, os, collections
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out"
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(root + "/r04_pmc_c*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "pair_sweep" in k: k = k[k.index("pair_sweep"):][:24]
        elif "laplace_cart_tiled_kernel2<1" in k: k = "laplace_cart_tiled_kernel2<1, 11>"
        else: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        print("   %-40s %16.0f per launch" % (c, acc[k][c] / cnt[k][c]))
PY
