#!/bin/bash
cd $GRAFT_REPO_ROOT
export KBENCH_SIZES=${KBENCH_SIZES:-1280x800x1000}
for round in 1 2; do
  echo "== tall, 4 waves/SIMD"; DOOMGPU_RASTER=tall timeout -k 10 300 python3 tests/manual/gpu_kbench.py 2>&1 | grep "B=\|rror\|fault"
  echo "== tall, 6 waves/SIMD"; DOOMGPU_LIB=$GRAFT_REPO_ROOT/build/variants/libdoomgpu_tall6.so DOOMGPU_RASTER=tall timeout -k 10 300 python3 tests/manual/gpu_kbench.py 2>&1 | grep "B=\|rror\|fault"
  echo "== tiles"; DOOMGPU_RASTER=tiles timeout -k 10 300 python3 tests/manual/gpu_kbench.py 2>&1 | grep "B=\|rror\|fault"
done
