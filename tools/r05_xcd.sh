#!/bin/bash
# XCD-aware workgroup -> (frame, strip) mapping of dg_raster_tiles (tools/experiments/r05_xcd_swizzle.patch built as build/variants/libdoomgpu_xcd.so):
# DOOMGPU_XCD_MODE = R: runs of R consecutive frames per XCD, dealt round robin (0 = dispatch order); three interleaved rounds
cd $GRAFT_REPO_ROOT
export DOOMGPU_LIB=$GRAFT_REPO_ROOT/build/variants/libdoomgpu_xcd.so KBENCH_SIZES=${KBENCH_SIZES:-1280x800x1000,320x200x1000,2560x1600x248}
for round in 1 2 3; do for m in ${MODES:-0 1 2 4 8}; do
  echo -n "mode $m: "; DOOMGPU_XCD_MODE=$m python3 tests/manual/gpu_kbench.py 2>&1 | grep "B=" | sed 's/.*mismatches \([0-9]*\)\/.*setup \([0-9.]*\) ms raster \([0-9.]*\) ms.*/\1 bad \3/' | tr '\n' ' '; echo
done; done
