#!/usr/bin/env python3
"""Per-kernel totals of the LAST rt_bvh_build in a rocprofv3 kernel trace made of tools/build_trace.py.
Usage: build_breakdown.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "k_calculate" in r["Kernel_Name"]]
marker = rows[starts[0]]["Kernel_Name"] if starts else None
firsts = [i for i in starts if rows[i]["Kernel_Name"] == marker]
rows = rows[firsts[-1]:]
t0, t1 = int(rows[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in rows)
tot, cnt = defaultdict(float), defaultdict(int)
busy = 0.0
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    name = r["Kernel_Name"].split("(")[0][:48]
    tot[name] += d
    cnt[name] += 1
    busy += d
print(f"span {(t1 - t0) / 1e6:.3f} ms, kernels {busy:.3f} ms in {len(rows)} launches")
for name in sorted(tot, key=tot.get, reverse=True):
    print(f"{tot[name]:9.3f} ms {cnt[name]:5d}  {name}")
