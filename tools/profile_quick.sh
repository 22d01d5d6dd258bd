#!/bin/bash
# Quick counter passes for the raster kernel (see profile_gpu.sh for the full recipe).
TAG=${1:-q}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
BENCH="python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-resident --no-host-frames --no-latency"
export TMPDIR=/tmp
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1 && echo trace ok
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH --output-format csv -d $OUT/pmc_sq1 -- $BENCH > $OUT/pmc_sq1.log 2>&1 && echo sq1 ok
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM --output-format csv -d $OUT/pmc_sq2 -- $BENCH > $OUT/pmc_sq2.log 2>&1 && echo sq2 ok
rocprofv3 --pmc SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS SQ_IFETCH SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM --output-format csv -d $OUT/pmc_sq3 -- $BENCH > $OUT/pmc_sq3.log 2>&1 && echo sq3 ok
rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32 --output-format csv -d $OUT/pmc_sq4 -- $BENCH > $OUT/pmc_sq4.log 2>&1 && echo sq4 ok
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LEVEL_WAVES SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM --output-format csv -d $OUT/pmc_sq5 -- $BENCH > $OUT/pmc_sq5.log 2>&1 && echo sq5 ok
