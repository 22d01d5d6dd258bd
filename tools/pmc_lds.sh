#!/bin/bash
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_lds; mkdir -p $OUT
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_LDS --output-format csv -d $OUT/pmc -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-resident --no-host-frames --no-latency > $OUT/log 2>&1 || { echo "rocprofv3 run failed:"; tail -5 $OUT/log; exit 1; }
