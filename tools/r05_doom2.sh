#!/bin/bash
# frames/s on the generated map of doom2's scale (18 131 segs), whole timed path at 1280x800, through each front end; then the kernel trace of the seg walk on it
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r05/doom2; mkdir -p $OUT
B="python3 bench.py --config 3 --synth-map 2002:32x24:500 --steps ${STEPS:-10} --warmup 1 --no-cpu-baseline --no-resident --no-host-frames --no-latency --no-side-legs"
for fe in ${FES:-segs device host auto}; do
  timeout -k 10 300 $B --front-end $fe > $OUT/$fe.json 2> $OUT/$fe.err || { echo "$fe FAILED"; tail -3 $OUT/$fe.err; exit 1; }
  python3 - $OUT/$fe.json $fe <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
g=d.get('gpu_ms_per_batch',{})
print(sys.argv[2], 'frames/s', round(d['value']), 'ms/step', round(d['ms_per_step'],3), 'gpu ms/batch median', round(g.get('median',0),3), 'front_end', d['config']['front_end'], 'fallbacks', d.get('fallbacks'))
PY
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B --front-end segs > $OUT/trace.log 2>&1 && cat $OUT/trace/*/*_kernel_stats.csv | cut -d, -f1-4,6,7 | head -12
