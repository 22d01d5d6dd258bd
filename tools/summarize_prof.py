#!/usr/bin/env python3
"""Condense a gpurun_out/prof_<tag>/ directory (tools/profile_gpu.sh) into a markdown summary for profiles/."""
import collections, csv, glob, sys
base, title = sys.argv[1], sys.argv[2]
out = [f"# rocprofv3 summary — {title}\n\n"]
ks = glob.glob(base + "/trace/*/*_kernel_stats.csv")
if ks:
    out.append("## --kernel-trace --stats (kernel_stats.csv)\n```\n" + open(ks[0]).read() + "```\n")
out.append("## PMC passes (separate runs, mean per dispatch)\n```\n")
for f in sorted(glob.glob(base + "/pmc_*/*/*_counter_collection.csv")):
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
