#!/bin/bash
# Every single-GPU BASELINE configuration through bench.py's timed path (one line each): tools/bench_configs.sh [extra bench.py flags]
for c in 1 2 3 4 5; do echo -n "config $c: "; python3 bench.py --config $c --no-cpu-baseline --no-host-frames --no-latency "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print(round(d['value']), 'fps', round(d['ms_per_step'],4), 'ms/step | raster', round(r['mean_launch_ms'],4), 'ms frac', round(r['frac'],4), 'isolated', round(r.get('isolated_frac',0),4), '| walk', round(r['front_end_kernels_mean_ms'],4), 'ms | host', round(d['host']['ms_per_batch'],3), 'ms/batch on', d['host']['threads'], 'threads | resident', round(d['resident_replay']['value']))"; done
