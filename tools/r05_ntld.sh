#!/bin/bash
# non-temporal span loads in the rasteriser's staging (variant ntld) against the product with a tile-rows override (variant rows), tile rows 7 and 13
cd $GRAFT_REPO_ROOT
export KBENCH_SIZES=${KBENCH_SIZES:-1280x800x1000,2560x1600x248}
for round in 1 2 3; do for v in rows ntld; do for r in 7 13; do
  echo -n "$v rows $r: "; DOOMGPU_LIB=$GRAFT_REPO_ROOT/build/variants/libdoomgpu_$v.so DOOMGPU_TILE_ROWS=$r python3 tests/manual/gpu_kbench.py 2>&1 | grep "B=" | sed 's/.*mismatches \([0-9]*\)\/.*setup \([0-9.]*\) ms raster \([0-9.]*\) ms.*/\1 bad \3/' | tr '\n' ' '; echo
done; done; done
