#!/bin/bash
# after a change to the seg walk: its GPU parity tests, then frames/s + kernel trace on the doom2-scale map and on configs 2 / 3
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests/test_seg_walk_limits.py tests/test_gpu_parity.py tests/test_hand_wad.py -m gpu -x -q -k "seg or walk or front_end or doom2 or state or random_views or 2560 or auto or room or forms" 2>&1 | tail -4 || exit 1
FES="segs" STEPS=10 tools/r05_doom2.sh || exit 1
OUT=gpurun_out/r05/segs_check; mkdir -p $OUT
for c in 2 3; do
  [ $c = 2 ] && export DOOMGPU_RASTER_OVERLAP=0 || unset DOOMGPU_RASTER_OVERLAP      # config 2 on one stream: the kernels' own durations
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_c$c -- python3 bench.py --config $c --front-end segs --steps 20 --warmup 2 --no-cpu-baseline --no-resident --no-host-frames --no-latency --no-side-legs > $OUT/c$c.json 2> $OUT/c$c.err || exit 1
  echo "config $c"; cat $OUT/trace_c$c/*/*_kernel_stats.csv | cut -d, -f1-4 | head -9
  python3 -c "
import json,sys
d=json.loads(open('$OUT/c$c.json').read().strip().splitlines()[-1]); print('  frames/s under the tracer', round(d['value']))"
done
