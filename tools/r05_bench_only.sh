#!/bin/bash
cd $GRAFT_REPO_ROOT
S=$(date +%s.%N)
python3 bench.py --steps 20 --warmup 2 > gpurun_out/r05/bench_default.json 2> gpurun_out/r05/bench_default.err
E=$(date +%s.%N); echo "bench.py wall time: $(echo "$E - $S" | bc) s"
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r05/bench_default.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("value", round(d["value"]), "ms/step", round(d["ms_per_step"], 4), "raster", round(r["mean_launch_ms"], 4), "frac", round(r["frac"], 4), "fe", round(r["front_end_kernels_mean_ms"], 4), d["config"]["front_end"])
print("issue", r.get("issue"))
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["gpu_frames_bit_exact"], "resident", round(d["resident_replay"]["value"]), "latency 1024", d["latency"]["sizes"]["1024x768"]["pinned"]["median_ms"])
for k, v in (d.get("side_legs") or {}).items():
    print(k, {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v.items() if kk not in ("workload", "unit", "warmup", "fallbacks")})
PY
