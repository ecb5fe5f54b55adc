#!/usr/bin/env python3
"""Per-step view of a rocprofv3 rocpd database: steps are delimited by adamw_table_kernel; prints, for the last N steps,
wall time, busy time, idle gaps and the kernel count (to see launch gaps / serialization inside graph replays)."""
import sqlite3, sys
from prof_summary import short
c = sqlite3.connect(sys.argv[1])
rows = sorted(c.execute("select name, start, end from kernels").fetchall(), key=lambda r: r[1])
steps, cur = [], []
for n, s, e in rows:
    cur.append((short(n), s, e))
    if "adamw_table" in n:
        steps.append(cur); cur = []
for st in steps[-int(sys.argv[2]) if len(sys.argv) > 2 else -6:]:
    t0, t1 = st[0][1], st[-1][2]
    busy = sum(e - s for _, s, e in st)
    # union of intervals (kernels may overlap with two streams)
    cover, end = 0, t0
    for _, s, e in sorted(st, key=lambda r: r[1]):
        if e > end:
            cover += e - max(s, end); end = e
    gaps = sorted(((b[1] - a[2]) for a, b in zip(st[:-1], st[1:]) if b[1] > a[2]), reverse=True)
    print(f"step: {len(st):4d} kernels, wall {(t1-t0)/1e6:7.3f} ms, sum of kernel times {busy/1e6:7.3f} ms, covered {cover/1e6:7.3f} ms, "
          f"idle {(t1-t0-cover)/1e6:6.3f} ms; 5 largest gaps (us): {[round(g/1e3,1) for g in gaps[:5]]}")
