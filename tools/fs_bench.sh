for fe in segs; do for c in 2 3; do echo "== config $c front end $fe"; python3 bench.py --config $c --front-end $fe --steps 20 --no-cpu-baseline --no-host-frames --no-latency 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print(round(d['value']), 'fps', round(d['ms_per_step'],4), 'ms/step | raster', round(r['mean_launch_ms'],4), 'fe kernels', round(r['front_end_kernels_mean_ms'],4), 'host ms', round(d['host']['ms_per_batch'],3), 'resident', round(d['resident_replay']['value']), d['fallbacks'])"; done; done
