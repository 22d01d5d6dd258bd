#!/bin/bash
# kbench of both rasterisers of the working tree, interleaved (DOOMGPU_RASTER=tall | tiles)
cd $GRAFT_REPO_ROOT
export KBENCH_SIZES=${KBENCH_SIZES:-1280x800x250,320x200x1000,1280x800x1000}
for round in 1 2; do
  for r in tall tiles; do echo "== $r"; DOOMGPU_RASTER=$r timeout -k 10 300 python3 tests/manual/gpu_kbench.py 2>&1 | grep "B=\|rror\|fault" ; done
done
