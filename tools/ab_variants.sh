#!/bin/bash
# A/B timing of kernel variants in build/variants/ inside ONE gpurun call (box-to-box variance is ~2-3 %): tools/ab_variants.sh v5 v6 ...
for round in 1 2 3; do
  for v in "$@"; do
    echo -n "$v: "; DOOMGPU_LIB=$GRAFT_REPO_ROOT/build/variants/libdoomgpu_$v.so python3 tests/manual/gpu_kbench.py 2>&1 | grep "1280x800" | sed 's/.*raster \([0-9.]*\) ms.*/\1/'
  done
done
