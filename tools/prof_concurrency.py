#!/usr/bin/env python3
"""How much of a step runs with 1 / 2 / 3+ kernels in flight (rocprofv3 rocpd database; steps delimited by adamw_table_kernel), and
which kernels own the time that runs ALONE.  usage: prof_concurrency.py results.db [n_steps]"""
import collections, sqlite3, sys
from prof_summary import short
c = sqlite3.connect(sys.argv[1])
rows = sorted(c.execute("select name, start, end from kernels").fetchall(), key=lambda r: r[1])
steps, cur = [], []
for n, s, e in rows:
    cur.append((short(n), s, e))
    if "adamw_table" in n:
        steps.append(cur); cur = []
for st in steps[-(int(sys.argv[2]) if len(sys.argv) > 2 else 3):]:
    ev = []
    for k, s, e in st:
        ev.append((s, 1, k)); ev.append((e, -1, k))
    ev.sort()
    active, last, hist, alone = {}, st[0][1], collections.Counter(), collections.Counter()
    for t, d, k in ev:
        n = sum(active.values())
        hist[min(n, 3)] += t - last
        if n == 1:
            alone[next(a for a, v in active.items() if v > 0)] += t - last
        last = t
        active[k] = active.get(k, 0) + d
    tot = sum(hist.values())
    print(f"step wall {tot/1e6:.3f} ms: idle {hist[0]/1e6:.3f}, one kernel {hist[1]/1e6:.3f}, two {hist[2]/1e6:.3f}, three+ {hist[3]/1e6:.3f} ms")
    print("   alone: " + ", ".join(f"{k[:44]} {v/1e6:.2f}" for k, v in alone.most_common(10)))
