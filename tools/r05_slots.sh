#!/bin/bash
# does the raster launch's time in the timed steps depend on how many slots (3 GB framebuffers) rotate?  bench.py --slots 1 2 4 8
cd $GRAFT_REPO_ROOT
for s in 4 2 1 8 4; do echo -n "slots $s: "; python3 bench.py --slots $s --steps 20 --warmup 2 --no-cpu-baseline --no-host-frames --no-latency --no-side-legs 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print(round(d['value']), 'fps | raster', round(r['mean_launch_ms'],4), 'isolated', round(r.get('isolated_launch_ms') or 0,4), 'fe', round(r['front_end_kernels_mean_ms'],4), '| resident', round(d['resident_replay']['value']))"; done
