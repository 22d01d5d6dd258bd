#!/usr/bin/env python3
"""Durations of, and idle gaps between, the last kernels of a rocprofv3 --kernel-trace csv directory (one in-order stream): tools/trace_gaps.py <dir>"""
import csv, glob, sys
f=sorted(glob.glob(sys.argv[1]+'/**/*kernel_trace.csv', recursive=True))[-1]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
def nm(r):
    n=r['Kernel_Name']
    for k in ('raster_tiles','fe_columns','fe_gaps','fe_scan','fe_scatter','fillBuffer','copyBuffer','row_table'):
        if k in n: return k
    return n[:20]
prev=None
for r in rows[-13:]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    print(nm(r), round((e-s)/1e3,1), 'gap before', round((s-prev)/1e3,1) if prev else 0)
    prev=e
