#!/bin/bash
cd $GRAFT_REPO_ROOT
for sz in "320 200" "640 400" "800 600" "1024 768"; do set -- $sz
  for rep in 1 2; do echo -n "$1x$2 auto (default overlap policy) run $rep: "; python3 bench.py --width $1 --height $2 --steps 20 --no-cpu-baseline --no-host-frames --no-latency --no-side-legs --no-resident 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value']), 'fps', round(d['ms_per_step'],4), 'ms/step host', round(d['host']['ms_per_batch'],3), d['config']['front_end'])"; done; done
