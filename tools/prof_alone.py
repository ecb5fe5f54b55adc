#!/usr/bin/env python3
"""How much of a kernel's in-step duration is concurrency?  From a rocprofv3 rocpd database: for every kernel name over the last N
steps (delimited by adamw_table_kernel) the mean duration of the dispatches that ran ALONE (another queue's kernels cover < 10 % of
their span) and of those that ran NEXT TO another queue's kernels (>= 50 % covered), with counts, and the time the second group
would take at the first group's mean.
usage: prof_alone.py results.db [N=24] [top=40]"""
import sqlite3, sys, bisect
from collections import defaultdict
from prof_summary import short

db = sys.argv[1]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 24
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
c = sqlite3.connect(db)
rows = sorted(c.execute("select name, start, end, queue_id, grid_x, workgroup_x from kernels").fetchall(), key=lambda r: r[1])
steps, cur = [], []
for r in rows:
    cur.append(r)
    if "adamw_table" in r[0]:
        steps.append(cur); cur = []
steps = steps[-N:]
stat = defaultdict(lambda: [0, 0.0, 0, 0.0, 0, 0.0])      # alone n, t; mixed n, t; next-to n, t
for st in steps:
    byq = defaultdict(list)
    for r in st:
        byq[r[3]].append((r[1], r[2]))
    for r in st:
        s, e = r[1], r[2]
        cov = 0
        for q, iv in byq.items():
            if q == r[3]:
                continue
            i = bisect.bisect_left(iv, (s, 0)) - 1
            i = max(i, 0)
            while i < len(iv) and iv[i][0] < e:
                cov += max(0, min(e, iv[i][1]) - max(s, iv[i][0]))
                i += 1
        f = cov / max(e - s, 1)
        k = (short(r[0]), r[4] // max(r[5], 1))
        v = stat[k]
        j = 0 if f < 0.1 else (4 if f >= 0.5 else 2)
        v[j] += 1; v[j + 1] += (e - s) / 1e3
print(f"# last {len(steps)} steps; alone = < 10 % of the span covered by another queue's kernels, next-to = >= 50 %")
print(f"{'kernel':70s} {'wgs':>6s} {'alone n':>8s} {'us':>7s} {'mixed n':>8s} {'us':>7s} {'next n':>7s} {'us':>7s} {'excess ms/step':>14s}")
tab = []
for (n, wg), v in stat.items():
    a = v[1] / v[0] if v[0] else None
    ex = ((v[3] - v[2] * a) + (v[5] - v[4] * a)) / 1e3 / len(steps) if a else 0.0
    tab.append((-(v[1] + v[3] + v[5]), n, wg, v, a, ex))
tab.sort()
tot = 0.0
for _, n, wg, v, a, ex in tab[:top]:
    f = lambda cnt, t: f"{cnt:8d} {t / cnt:7.1f}" if cnt else f"{0:8d} {'-':>7s}"
    print(f"{n[:70]:70s} {wg:6d} {f(v[0], v[1])} {f(v[2], v[3])} {f(v[4], v[5])[1:]} {ex:14.3f}")
    tot += ex
print(f"# excess over the alone mean, listed kernels: {tot:.3f} ms per step")
