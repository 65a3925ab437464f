#!/usr/bin/env python3
"""Timeline of the last kernels of a rocprofv3 --kernel-trace csv.  Usage: trace_tail.py trace.csv [n]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 32
t0 = None
for r in rows[-n:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    t0 = s if t0 is None else t0
    name = r["Kernel_Name"].replace("void ", "").split("(")[0]
    print(f"{(s - t0) / 1e6:9.3f} -> {(e - t0) / 1e6:9.3f} {(e - s) / 1e3:10.1f}us grid={r['Grid_Size_X']:>9s} q={r.get('Queue_Id', '')} {name[:50]}")
