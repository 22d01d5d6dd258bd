#!/bin/bash
# PMC passes over tests/manual/gpu_kbench.py (one rocprofv3 run per counter set): tools/pmc_kbench.sh <tag> "<counters set 1>" "<set 2>" ...
TAG=${1:-p}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/${ROUND:-r05}/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d $OUT/set$i -- python3 $GRAFT_REPO_ROOT/tests/manual/gpu_kbench.py > $OUT/set$i.log 2>&1 || echo "set $i failed"
done
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT
