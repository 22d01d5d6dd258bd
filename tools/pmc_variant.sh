#!/bin/bash
# Counter pass + timing for kernel experiment variants built into build/variants/ (tuning aid).
export TMPDIR=/tmp
for v in "$@"; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_var_$v; mkdir -p $OUT
  DOOMGPU_LIB=$GRAFT_REPO_ROOT/build/variants/libdoomgpu_$v.so rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-resident --no-host-frames --no-latency > $OUT/log 2>&1 || { echo "rocprofv3 run failed:"; tail -5 $OUT/log; exit 1; }
  DOOMGPU_LIB=$GRAFT_REPO_ROOT/build/variants/libdoomgpu_$v.so python3 tests/manual/gpu_kbench.py 2>&1 | tail -2 | head -1
done
