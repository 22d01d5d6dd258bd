#!/bin/bash
# In-pipeline A/B of library builds inside ONE gpurun call: tools/ab_bench.sh head base ...  ("base" = the product library).
# Three interleaved rounds of the default bench command (no CPU baseline / host-frames legs); prints frames/s, ms per step and the raster
# kernel's in-pipeline and isolated ms of each.
for round in 1 2 3; do
  for v in "$@"; do
    if [ "$v" != base ]; then export DOOMGPU_LIB=$GRAFT_REPO_ROOT/build/variants/libdoomgpu_$v.so; else unset DOOMGPU_LIB; fi
    echo -n "$v: "; python3 bench.py --no-cpu-baseline --no-host-frames --no-latency ${BENCH_ARGS} 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print(round(d['value']), 'fps', round(d['ms_per_step'],4), 'ms/step | raster', round(r.get('mean_launch_ms', 0),4), 'ms in pipeline, frac', round(r['frac'],4), 'isolated', round(r.get('isolated_frac',0),4), '| fe', round(r.get('front_end_kernels_mean_ms',0),4), '| resident', d.get('resident_replay'))"
  done
done
