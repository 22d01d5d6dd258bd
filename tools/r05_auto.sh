#!/bin/bash
# DG_FE_AUTO after a change to its choice: its GPU tests, the default bench line with its side legs, config 2 and 640x400 in AUTO
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "auto or alternating or overlapped" 2>&1 | tail -2 || exit 1
S=$(date +%s)
python3 bench.py --steps 20 --warmup 2 > gpurun_out/r05/bench_default.json 2> gpurun_out/r05/bench_default.err || { tail -5 gpurun_out/r05/bench_default.err; exit 1; }
echo "bench.py wall time: $(( $(date +%s) - S )) s"
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r05/bench_default.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("value", round(d["value"]), "ms/step", round(d["ms_per_step"], 4), "raster", round(r["mean_launch_ms"], 4), "frac", round(r["frac"], 4), "fe", round(r["front_end_kernels_mean_ms"], 4), d["config"]["front_end"])
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["gpu_frames_bit_exact"], "resident", round(d["resident_replay"]["value"]))
for k, v in (d.get("side_legs") or {}).items():
    print(k, {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v.items() if kk not in ("workload", "unit", "warmup", "fallbacks")})
PY
tools/fs_bench.sh "auto" "2 3" --no-side-legs
