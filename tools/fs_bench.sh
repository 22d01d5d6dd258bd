#!/bin/bash
# End-to-end frames/s of BASELINE configs through the front ends: tools/fs_bench.sh "<front ends>" "<configs>" [extra bench.py flags]
# (one gpurun call; prints frames/s, ms per step, raster / front-end kernel ms, host ms per batch, resident replay, front ends used)
for fe in ${1:-auto device segs}; do for c in ${2:-2 3}; do echo -n "config $c --front-end $fe ${@:3}: "; python3 bench.py --config $c --front-end $fe --steps 20 --no-cpu-baseline --no-host-frames --no-latency "${@:3}" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print(round(d['value']), 'fps', round(d['ms_per_step'],4), 'ms/step (median', round(d['ms_per_step_median'],4), 'max', round(d['ms_per_step_max'],3), 'first', round(d['ms_per_step_first'],3), ') | raster', round(r['mean_launch_ms'],4), 'frac', round(r['frac'],4), 'fe kernels', round(r['front_end_kernels_mean_ms'],4), '| host ms', round(d['host']['ms_per_batch'],3), 'threads', d['host']['threads'], '| resident', round(d['resident_replay']['value']), d['fallbacks'], '|', d['config']['front_end'])"; done; done
