#!/usr/bin/env python3
"""Per-kernel summary (calls, total ms, average us, share) from a rocprofv3
--kernel-trace results database."""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by sum(duration) desc").fetchall()
tot = sum(r[2] for r in rows)
print(f"{'kernel':64s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>10s} {'%':>6s}")
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    name = r[0].replace("void ", "").split("(")[0]
    print(f"{name[:64]:64s} {r[1]:7d} {r[2] / 1e6:10.3f} {r[3] / 1e3:10.2f} {r[4] / 1e3:9.2f} {r[5] / 1e3:10.2f} {100.0 * r[2] / tot:6.2f}")
