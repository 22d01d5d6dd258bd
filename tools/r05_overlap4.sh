#!/bin/bash
# the raster launch on its own stream (overlapping the next batch's front-end kernels) at sizes above the 500 000-pixel rule, with the round's last build
cd $GRAFT_REPO_ROOT
for round in 1 2; do for wh in "1280 800" "1024 768"; do set -- $wh; for o in 0 1; do
  echo -n "$1x$2 overlap $o: "; DOOMGPU_RASTER_OVERLAP=$o python3 bench.py --width $1 --height $2 --steps 20 --warmup 2 --no-cpu-baseline --no-resident --no-host-frames --no-latency --no-side-legs 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print(round(d['value']), 'fps | ms/step', round(d['ms_per_step'],4), 'raster', round(r['mean_launch_ms'],4), 'fe', round(r['front_end_kernels_mean_ms'],4))"
done; done; done
