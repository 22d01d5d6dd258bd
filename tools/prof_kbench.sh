#!/bin/bash
# Per-kernel timing (rocprofv3 --kernel-trace --stats) of tests/manual/gpu_kbench.py: tools/prof_kbench.sh <tag> [ENV=V ...]
TAG=${1:-k}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02/trace_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tests/manual/gpu_kbench.py > $OUT/log.txt 2>&1
grep "mismatches" $OUT/log.txt
find $OUT -name "*kernel_stats.csv" | head -1 | xargs cat | cut -c1-160
