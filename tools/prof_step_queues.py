#!/usr/bin/env python3
"""Per-step, per-queue view of a rocprofv3 rocpd database (two compute streams: which of them is the critical path, and when).
Steps are delimited by adamw_table_kernel.  For the last N steps prints the wall time, each hardware queue's busy time and kernel
count, how long exactly one / both / no queue was busy, and (with --dump K) the K-th last step's kernels with their queue.
usage: prof_step_queues.py results.db [N=6] [--dump K] [--kinds]"""
import sqlite3, sys
from collections import defaultdict
from prof_summary import short

db = sys.argv[1]
args = sys.argv[2:]
N = int(args[0]) if args and not args[0].startswith("--") else 6
dump = int(args[args.index("--dump") + 1]) if "--dump" in args else 0
kinds = "--kinds" in args
c = sqlite3.connect(db)
rows = sorted(c.execute("select name, start, end, queue_id, grid_x, workgroup_x from kernels").fetchall(), key=lambda r: r[1])
steps, cur = [], []
for r in rows:
    cur.append(r)
    if "adamw_table" in r[0]:
        steps.append(cur); cur = []


def family(n):
    n = short(n)
    for k, v in (("wgrad_grouped", "wgrad"), ("adamw", "adamw"), ("gemm_", "gemm"), ("ln_fwd", "ln"), ("ln_bwd", "ln"), ("attn", "attn")):
        if k in n:
            return v
    return "other"


for si, st in enumerate(steps[-N:]):
    t0, t1 = min(r[1] for r in st), max(r[2] for r in st)
    qs = defaultdict(list)
    for r in st:
        qs[r[3]].append(r)
    ev = []
    for r in st:
        ev.append((r[1], 1, r[3])); ev.append((r[2], -1, r[3]))
    ev.sort()
    active = defaultdict(int)
    last, span = t0, defaultdict(float)
    for t, d, q in ev:
        nb = sum(1 for v in active.values() if v > 0)
        span[min(nb, 2)] += t - last
        last = t
        active[q] += d
    print(f"step -{len(steps[-N:]) - si}: {len(st)} kernels, wall {(t1 - t0) / 1e6:.3f} ms; none busy {span[0] / 1e6:.3f}, one queue {span[1] / 1e6:.3f}, "
          f"two or more {span[2] / 1e6:.3f} ms; " + "; ".join(f"queue {q}: {len(v)} kernels {sum(r[2] - r[1] for r in v) / 1e6:.3f} ms" for q, v in sorted(qs.items())))
    if kinds:
        fam = defaultdict(float)
        for r in st:
            fam[family(r[0])] += (r[2] - r[1]) / 1e6
        print("    kernel time by family (ms): " + ", ".join(f"{k} {v:.3f}" for k, v in sorted(fam.items(), key=lambda kv: -kv[1])))
if dump:
    st = steps[-dump]
    t0 = min(r[1] for r in st)
    for n, s, e, q, gx, wx in st:
        print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f}  q{q}  {gx // max(wx, 1):5d} wg  {short(n)[:80]}")
