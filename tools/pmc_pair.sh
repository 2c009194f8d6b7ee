cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM -d $GRAFT_REPO_ROOT/gpurun_out/r04_pmc_v1 --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 > /dev/null 2>&1
