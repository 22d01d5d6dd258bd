#!/usr/bin/env python3
"""Median duration per (kernel, grid) from a rocprofv3 --kernel-trace csv directory: tools/trace_summary.py <dir> [min_calls]"""
import collections, csv, glob, sys
d = collections.defaultdict(list)
for fn in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        d[(r["Kernel_Name"].split("(")[0][-24:], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
mc = int(sys.argv[2]) if len(sys.argv) > 2 else 5
for k, v in sorted(d.items()):
    if len(v) >= mc and "dg" in k[0]:
        v = sorted(v)
        print(f"{k[0]:26s} grid {k[1]:>7s} x {k[2]:>4s} x {k[3]:>4s}  n={len(v):3d}  median {v[len(v)//2]/1e3:8.1f} us  min {v[0]/1e3:8.1f} us")
