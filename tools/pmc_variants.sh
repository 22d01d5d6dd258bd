#!/bin/bash
# One PMC pass (rocprofv3 --pmc, no tracing) + one timing run of tests/manual/gpu_kbench.py per kernel variant in build/variants/:
#   tools/pmc_variants.sh "<counters>" v1 v2 ...      ("base" = the product library)
# Output: gpurun_out/r04/pmcv_<variant>.txt (summary lines of dg_raster_tiles) and the kbench line.
SET=$1; shift
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/${ROUND:-r05}/pmcv_$v; mkdir -p $OUT
  if [ "$v" != base ]; then export DOOMGPU_LIB=$GRAFT_REPO_ROOT/build/variants/libdoomgpu_$v.so; else unset DOOMGPU_LIB; fi
  KBENCH_SIZES=${KBENCH_SIZES:-1280x800x250} rocprofv3 --pmc $SET --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tests/manual/gpu_kbench.py > $OUT/log.txt 2>&1 || echo "variant $v: rocprofv3 failed"
  echo "== $v"; python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT | grep raster_tiles
  KBENCH_SIZES=${KBENCH_SIZES:-1280x800x250} python3 $GRAFT_REPO_ROOT/tests/manual/gpu_kbench.py 2>&1 | grep "B="
done
