#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc CSV output (tools/prof_pmc.sh) per kernel."""
import csv
import glob
import json
import sys
from collections import defaultdict

root = sys.argv[1]
agg = defaultdict(lambda: defaultdict(float))
calls = defaultdict(lambda: defaultdict(int))
for f in glob.glob(f"{root}/*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        c = row["Counter_Name"]
        agg[k][c] += float(row["Counter_Value"])
        calls[k][c] += 1
out = {}
for k in sorted(agg, key=lambda k: -agg[k].get("SQ_WAVE_CYCLES", 0)):
    out[k] = {c: agg[k][c] for c in sorted(agg[k])}
    out[k]["_dispatches"] = max(calls[k].values())
print(json.dumps(out, indent=1))
