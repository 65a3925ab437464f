#!/usr/bin/env python3
"""Print the launches of the last rt_render_frames of a rocprofv3 kernel trace (csv) as a timeline.
Usage: timeline.py <kernel_trace.csv>"""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
first = [i for i, r in enumerate(rows) if "k_generate" in r["Kernel_Name"]][-1]
t0 = int(rows[first]["Start_Timestamp"])
for r in rows[first:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e6:8.3f} {(e - s) / 1e6:8.3f}  q{r.get('Queue_Id')}  {r['Kernel_Name'][:64]}")
