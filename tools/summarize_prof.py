#!/usr/bin/env python3
"""Condense a gpurun_out/prof_<tag>/ directory (tools/profile_gpu.sh) into a markdown summary for profiles/."""
import collections, csv, glob, sys
base, title = sys.argv[1], sys.argv[2]
out = [f"# rocprofv3 summary — {title}\n\n"]
import os as _os
ks = sorted(glob.glob(base + "/trace/*/*_kernel_stats.csv"), key=_os.path.getmtime)
if ks:
    out.append("## --kernel-trace --stats (kernel_stats.csv)\n```\n" + open(ks[-1]).read() + "```\n")
# dg_raster_tiles launch by launch (kernel_trace.csv): bench.py's untimed clock warm-up and warm-up steps come first, its timed steps last —
# the stats above average over all of them, the GPU's clock ramp included
kt = sorted(glob.glob(base + "/trace/*/*_kernel_trace.csv"), key=_os.path.getmtime)
if kt:
    rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(kt[-1])) if "dg_raster_tiles" in r["Kernel_Name"])
    d = [x[1] / 1e6 for x in rows]
    if d:
        last = d[-20:]
        out.append("## dg_raster_tiles launch by launch (kernel_trace.csv, ms, in launch order)\n```\n" + " ".join(f"{x:.3f}" for x in d) + "\n```\n"
                   f"all {len(d)} launches: mean {sum(d)/len(d):.4f} ms; the last {len(last)} (bench.py's timed steps): **mean {sum(last)/len(last):.4f} ms**, min {min(last):.4f}, max {max(last):.4f}\n\n")
out.append("## PMC passes (separate runs, mean per dispatch)\n```\n")
import os
latest = {}
for f in glob.glob(base + "/pmc_*/*/*_counter_collection.csv"):
    d = f.split("/pmc_")[1].split("/")[0]
    if d not in latest or os.path.getmtime(f) > os.path.getmtime(latest[d]):
        latest[d] = f
for f in [latest[d] for d in sorted(latest)]:
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if "rocclr" in k:
            continue
        for c, x in v.items():
            out.append(f"{k:28s} {c:22s} {sum(x)/len(x):18.1f}  (n={len(x)})\n")
out.append("```\n")
sys.stdout.write("".join(out))
