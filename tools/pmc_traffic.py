#!/usr/bin/env python3
"""HBM bytes per launch of the traversal kernels from the FETCH_SIZE / WRITE_SIZE
passes of tools/prof_round.sh, plus the accounts the other passes allow: L2 hit
rate, lane utilisation (divergence), resident waves per SIMD (occupancy) and the
share of SIMD cycles with a vector instruction executing.  FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE tallies each 128-byte memory-side read request as 64 bytes, so it is
doubled; WRITE_SIZE is taken as is.
Usage: pmc_traffic.py <pmc_summary.json> <bench_line.json>"""
import json
import sys

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import source_id  # noqa: E402

summary = json.load(open(sys.argv[1]))
bench = json.load(open(sys.argv[2]))
out = {}
for kname, v in summary.items():
    # the plain (non-counting) instantiations only; k_traverse<false, ShadowSrcOverlapped> launches share the
    # chip with other kernels in normal runs and are not the kernel by itself
    if not kname.startswith("k_traverse<false, ") or "FETCH_SIZE" not in v:
        continue
    src = kname[len("k_traverse<false, "):].rstrip(">").strip()
    if src not in ("ShadowSrc", "QueueSrc", "ArraySrc"):
        continue
    n = v["_dispatches"]
    fetch = v["FETCH_SIZE"] * 1024.0 * 2.0
    write = v.get("WRITE_SIZE", 0.0) * 1024.0
    # SQ_WAVE_CYCLES, SQ_ACTIVE_INST_* count in units of 4 clocks; GRBM_GUI_ACTIVE is summed over the 8 XCDs.
    simd_clocks = v.get("GRBM_GUI_ACTIVE", 0.0) / 8.0 * 1024.0           # 256 CUs x 4 SIMDs, per-XCD busy clocks
    acct = {}
    if simd_clocks and "SQ_ACTIVE_INST_VALU" in v:
        acct = {
            "lane_utilisation": round(v["SQ_THREAD_CYCLES_VALU"] / (64.0 * v["SQ_ACTIVE_INST_VALU"]), 3),
            "waves_per_simd": round(v["SQ_WAVE_CYCLES"] * 4.0 / simd_clocks, 2),
            "valu_busy": round(v["SQ_ACTIVE_INST_VALU"] * 4.0 / simd_clocks, 3),
            "kernel_clocks_per_launch": int(v["GRBM_GUI_ACTIVE"] / 8.0 / n) if (n := v["_dispatches"]) else 0,
            "valu_instructions_per_launch": int(v["SQ_INSTS_VALU"] / v["_dispatches"]),
            "salu_instructions_per_launch": int(v["SQ_INSTS_SALU"] / v["_dispatches"]),
            "vmem_read_instructions_per_launch": int(v["SQ_INSTS_VMEM_RD"] / v["_dispatches"]),
            "l1_to_l2_read_requests_per_launch": int(v.get("TCP_TCC_READ_REQ_sum", 0.0) / v["_dispatches"]),
            # the vector memory path: lane accesses the L1s served, and how busy the 256 address (TA) and data-return (TD)
            # units were over the kernel's clocks
            "l1_lane_accesses_per_launch": int(v.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0.0) / v["_dispatches"]),
            "ta_busy": round(v.get("TA_TA_BUSY_sum", 0.0) / (256.0 * v["GRBM_GUI_ACTIVE"] / 8.0), 3),
            "td_busy": round(v.get("TD_TD_BUSY_sum", 0.0) / (256.0 * v["GRBM_GUI_ACTIVE"] / 8.0), 3),
        }
    out[f"k_traverse<{src}>"] = {
        **acct,
        "hbm_bytes_per_launch": int((fetch + write) / n), "fetch_bytes_per_launch": int(fetch / n),
        "write_bytes_per_launch": int(write / n), "launches": n,
        "workload": bench["config"]["workload"], "n_gpus": bench["n_gpus"], "source_id": source_id(),
        "l2_hit_rate": round(v["TCC_HIT_sum"] / max(v["TCC_HIT_sum"] + v.get("TCC_MISS_sum", 0.0), 1.0), 4) if "TCC_HIT_sum" in v else None,
    }
print(json.dumps(out, indent=1))
