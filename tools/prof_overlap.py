#!/usr/bin/env python3
"""Which kernels run concurrently with a given kernel (rocprofv3 rocpd database)?  usage: prof_overlap.py results.db name_substring [max_instances]"""
import collections, sqlite3, sys
from prof_summary import short
c = sqlite3.connect(sys.argv[1])
rows = sorted(c.execute("select name, start, end from kernels").fetchall(), key=lambda r: r[1])
pat = sys.argv[2]
lim = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
tot = collections.Counter()
n = 0
for i, (nm, s, e) in enumerate(rows):
    if pat not in nm:
        continue
    n += 1
    if n > lim:
        break
    j = i - 1
    while j >= 0 and rows[j][1] > s - 5_000_000:
        if rows[j][2] > s:
            tot[short(rows[j][0])[:70]] += min(rows[j][2], e) - s
        j -= 1
    j = i + 1
    while j < len(rows) and rows[j][1] < e:
        tot[short(rows[j][0])[:70]] += min(rows[j][2], e) - rows[j][1]
        j += 1
print(f"{n} instances of *{pat}*; kernels overlapping them (total overlap us):")
for k, v in tot.most_common(20):
    print(f"  {v / 1e3:10.1f}  {k}")
