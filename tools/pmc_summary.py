#!/usr/bin/env python3
"""Mean counter value per dispatch, grouped by (kernel, grid): tools/pmc_summary.py <dir with rocprofv3 --pmc csv output>."""
import collections, csv, glob, sys
acc = collections.defaultdict(list)
for fn in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        acc[(r["Kernel_Name"].split("(")[0][-28:], r["Grid_Size"], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    if len(v) >= 5:
        print(f"{k[0]:30s} grid {k[1]:>10s} {k[2]:34s} {sum(v)/len(v):16.1f}  (n={len(v)})")
