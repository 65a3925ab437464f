#!/usr/bin/env python3
"""Print the launches of a rocprofv3 kernel trace (csv) as a timeline, from the k_generate launch that is the `back`-th
from the end (default 1: the last rt_render_frames / frame).  Usage: timeline.py <kernel_trace.csv> [back]"""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
first = [i for i, r in enumerate(rows) if "k_generate" in r["Kernel_Name"]][-back]
t0 = int(rows[first]["Start_Timestamp"])
for r in rows[first:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.3f %8.3f  q%s  %s" % ((s - t0) / 1e6, (e - s) / 1e6, r.get("Queue_Id"), r["Kernel_Name"][:64]))
