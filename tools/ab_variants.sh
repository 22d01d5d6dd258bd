#!/bin/bash
# A/B timing of kernel variants in build/variants/ inside ONE gpurun call (box-to-box variance is ~2-3 %): tools/ab_variants.sh head base ...
# ("base" = the product library).  Three interleaved rounds; prints "front-end kernels ms / raster ms" of each size in KBENCH_SIZES.
for round in 1 2 3; do
  for v in "$@"; do
    if [ "$v" != base ]; then export DOOMGPU_LIB=$GRAFT_REPO_ROOT/build/variants/libdoomgpu_$v.so; else unset DOOMGPU_LIB; fi
    echo -n "$v: "; KBENCH_SIZES=${KBENCH_SIZES:-1280x800x250} python3 tests/manual/gpu_kbench.py 2>&1 | grep "B=" | sed 's/.*setup \([0-9.]*\) ms raster \([0-9.]*\) ms.*/\1\/\2/' | tr '\n' ' '; echo
  done
done
