#!/bin/bash
cd $GRAFT_REPO_ROOT
for sz in "640 400" "800 600" "1024 768"; do set -- $sz
  for ov in 0 1; do echo -n "$1x$2 overlap=$ov: "; DOOMGPU_RASTER_OVERLAP=$ov python3 bench.py --width $1 --height $2 --steps 20 --no-cpu-baseline --no-host-frames --no-latency --no-side-legs --no-resident --front-end segs 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value']), 'fps', round(d['ms_per_step'],4), 'ms/step', d['config']['front_end'][:60])"; done; done
