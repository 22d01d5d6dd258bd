#!/bin/bash
# the default benchmark command (what the driver runs) + the front-end table
cd $GRAFT_REPO_ROOT
python3 bench.py --steps 20 --warmup 2 > gpurun_out/r05/bench_default.json 2> gpurun_out/r05/bench_default.err
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r05/bench_default.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("value", round(d["value"]), "ms/step", round(d["ms_per_step"], 4), "raster", round(r["mean_launch_ms"], 4), "frac", round(r["frac"], 4), "fe", round(r["front_end_kernels_mean_ms"], 4), d["config"]["front_end"])
print("issue", r.get("issue"))
for k, v in (d.get("side_legs") or {}).items():
    print(k, {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v.items() if kk != "workload"})
PY
tools/fs_bench.sh "segs device" "2 3" --no-side-legs
