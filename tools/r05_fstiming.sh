#!/bin/bash
# phase clocks of dg_fs_frame / wave clocks of dg_fs_segs from the instrumented variant (tools/experiments/r05_fs_timing.patch built as build/variants/libdoomgpu_fstiming.so)
cd $GRAFT_REPO_ROOT
export DOOMGPU_LIB=$GRAFT_REPO_ROOT/build/variants/libdoomgpu_fstiming.so
B="python3 bench.py --config 3 --steps 10 --warmup 1 --no-cpu-baseline --no-resident --no-host-frames --no-latency --no-side-legs --front-end segs"
echo "== doom2-scale map"; timeout -k 10 300 $B --synth-map 2002:32x24:500 2>&1 >/dev/null | grep -A48 "fs timing"
echo "== bench map (seed 1993)"; timeout -k 10 300 $B 2>&1 >/dev/null | grep -A48 "fs timing"
