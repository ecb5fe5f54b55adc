#!/usr/bin/env python3
"""Timeline of the kernels around the optimizer update of one step (rocprofv3 rocpd database): start / end in us relative to the
step's first adamw_table_kernel launch.  usage: prof_timeline.py results.db [step_index_from_end=2] [window_us=2500] [before_us=200]"""
import sqlite3, sys
from prof_summary import short
c = sqlite3.connect(sys.argv[1])
rows = sorted(c.execute("select name, start, end from kernels").fetchall(), key=lambda r: r[1])
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
win = float(sys.argv[3]) if len(sys.argv) > 3 else 2500.0
before = float(sys.argv[4]) if len(sys.argv) > 4 else 200.0
ad = [i for i, r in enumerate(rows) if "adamw_table" in r[0]]
# group adamw launches into steps: a gap of more than 1.5 ms between consecutive launches starts a new step
groups, cur = [], [ad[0]]
for i in ad[1:]:
    if rows[i][1] - rows[cur[-1]][1] > 1.5e6:
        groups.append(cur); cur = [i]
    else:
        cur.append(i)
groups.append(cur)
g = groups[-back]
t0 = rows[g[0]][1]
print(f"# step with {len(g)} adamw launches; update spans {(rows[g[-1]][2] - t0) / 1e3:.1f} us")
for n, s, e in rows:
    if t0 - before * 1e3 <= s <= t0 + win * 1e3:
        print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  {short(n)[:90]}")
