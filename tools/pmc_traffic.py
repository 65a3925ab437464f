#!/usr/bin/env python3
"""HBM bytes per launch of the traversal kernels from the FETCH_SIZE / WRITE_SIZE
passes of tools/prof_round.sh.  FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE tallies each 128-byte memory-side read request as 64 bytes, so it is
doubled; WRITE_SIZE is taken as is.
Usage: pmc_traffic.py <pmc_summary.json> <bench_line.json>"""
import json
import sys

summary = json.load(open(sys.argv[1]))
bench = json.load(open(sys.argv[2]))
out = {}
for kname, v in summary.items():
    if "k_traverse<" not in kname or "FETCH_SIZE" not in v:
        continue
    src = "ShadowSrc" if "ShadowSrc" in kname else "QueueSrc" if "QueueSrc" in kname else "ArraySrc"
    n = v["_dispatches"]
    fetch = v["FETCH_SIZE"] * 1024.0 * 2.0
    write = v.get("WRITE_SIZE", 0.0) * 1024.0
    out[f"k_traverse<{src}>"] = {
        "hbm_bytes_per_launch": int((fetch + write) / n), "fetch_bytes_per_launch": int(fetch / n),
        "write_bytes_per_launch": int(write / n), "launches": n,
        "workload": bench["config"]["workload"], "n_gpus": bench["n_gpus"],
        "l2_hit_rate": round(v["TCC_HIT_sum"] / max(v["TCC_HIT_sum"] + v.get("TCC_MISS_sum", 0.0), 1.0), 4) if "TCC_HIT_sum" in v else None,
    }
print(json.dumps(out, indent=1))
