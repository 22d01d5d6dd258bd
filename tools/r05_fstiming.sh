#!/bin/bash
# phase clocks of dg_fs_frame from the instrumented variant (tools/experiments/r05_fs_timing.patch built as build/variants/libdoomgpu_fstiming.so)
cd $GRAFT_REPO_ROOT
export DOOMGPU_LIB=$GRAFT_REPO_ROOT/build/variants/libdoomgpu_fstiming.so DOOMGPU_RASTER_OVERLAP=0
B="python3 bench.py --steps 10 --warmup 1 --no-cpu-baseline --no-resident --no-host-frames --no-latency --no-side-legs --front-end segs"
echo "== doom2-scale map, 1280x800"; timeout -k 10 300 $B --config 3 --synth-map 2002:32x24:500 2>&1 >/dev/null | grep -A24 "fs timing"
echo "== config 3"; timeout -k 10 300 $B --config 3 2>&1 >/dev/null | grep -A24 "fs timing"
echo "== config 2"; timeout -k 10 300 $B --config 2 2>&1 >/dev/null | grep -A24 "fs timing"
