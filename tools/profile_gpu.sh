#!/bin/bash
# Profiling recipe for the GPU box (run through gpurun from the repo root).  Writes under gpurun_out/prof_<tag>/.
# Counter passes are separate runs with --pmc only (no trace domains), as the MI355X guide prescribes.
# Usage: tools/profile_gpu.sh <tag> [extra bench.py flags, e.g. --front-end host]
set -o pipefail
TAG=${1:-r01}
shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04/prof_$TAG
BENCH="python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-resident --no-host-frames --no-latency $*"
export TMPDIR=/tmp
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1 && echo trace ok
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1 && echo fetch ok
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/pmc_write.log 2>&1 && echo write ok
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/pmc_sq1 -- $BENCH > $OUT/pmc_sq1.log 2>&1 && echo sq1 ok
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM --output-format csv -d $OUT/pmc_sq2 -- $BENCH > $OUT/pmc_sq2.log 2>&1 && echo sq2 ok
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_tcc -- $BENCH > $OUT/pmc_tcc.log 2>&1 && echo tcc ok
find $OUT -name "*.csv" | head -40
