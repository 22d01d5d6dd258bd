#!/bin/bash
# Kernel trace of the device column walk + raster on 250 resident frames (both synthetic maps).
TAG=${1:-fe}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
export TMPDIR=/tmp
mkdir -p $OUT
for seed in 1993 1994; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$seed -- python3 tests/manual/gpu_fe_replay.py $seed > $OUT/trace_$seed.log 2>&1 && echo trace $seed ok
  cat $OUT/trace_$seed/*/*_kernel_stats.csv
done
