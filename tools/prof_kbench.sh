#!/bin/bash
# Per-kernel timing (rocprofv3 --kernel-trace) of tests/manual/gpu_kbench.py: tools/prof_kbench.sh <tag> [ENV=V ...]
TAG=${1:-k}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04/trace_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
env "$@" rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tests/manual/gpu_kbench.py > $OUT/log.txt 2>&1
echo "== $TAG $@"
grep "mismatches" $OUT/log.txt
python3 $GRAFT_REPO_ROOT/tools/trace_summary.py $OUT
