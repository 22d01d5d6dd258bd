#!/bin/bash
# tile rows per workgroup of dg_raster_tiles under the frame-per-XCD mapping (variant with a DOOMGPU_TILE_ROWS override: build/variants/libdoomgpu_rows.so)
cd $GRAFT_REPO_ROOT
export DOOMGPU_LIB=$GRAFT_REPO_ROOT/build/variants/libdoomgpu_rows.so KBENCH_SIZES=${KBENCH_SIZES:-1280x800x1000,1024x768x1000}
for round in 1 2; do for r in ${ROWS:-7 13 5 4 3}; do
  echo -n "rows $r: "; DOOMGPU_TILE_ROWS=$r python3 tests/manual/gpu_kbench.py 2>&1 | grep "B=" | sed 's/.*mismatches \([0-9]*\)\/.*setup \([0-9.]*\) ms raster \([0-9.]*\) ms.*/\1 bad \3/' | tr '\n' ' '; echo
done; done
