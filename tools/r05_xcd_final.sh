#!/bin/bash
# the shipped frame-per-XCD mapping: its parity test, then DOOMGPU_FRAME_PER_XCD=0 / 1 interleaved at several sizes (product library)
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "workgroup_to_frame or sizes_with_partial or widths" 2>&1 | tail -2 || exit 1
export KBENCH_SIZES=${KBENCH_SIZES:-1280x800x1000,1024x768x1000,800x600x1000,640x400x1000,2560x1600x250}
for round in 1 2 3; do for m in 0 1; do
  echo -n "frame per XCD $m: "; DOOMGPU_FRAME_PER_XCD=$m python3 tests/manual/gpu_kbench.py 2>&1 | grep "B=" | sed 's/.*mismatches \([0-9]*\)\/.*setup \([0-9.]*\) ms raster \([0-9.]*\) ms.*/\1 bad \3/' | tr '\n' ' '; echo
done; done
