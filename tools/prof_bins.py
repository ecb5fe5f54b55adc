#!/usr/bin/env python3
"""Bin a prof_step_queues.py --dump file: per 200-us bin, each queue's busy share and its two dominant kernels.
usage: prof_bins.py step_dump.txt [bin_us=200]"""
import collections, sys
rows = []
for ln in open(sys.argv[1]):
    p = ln.split(None, 6)
    if len(p) < 7 or not p[3].startswith("q"):
        continue
    try:
        s, e = float(p[0]), float(p[1])
    except ValueError:
        continue
    rows.append((s, e, p[3], p[6].strip()))
binw = float(sys.argv[2]) if len(sys.argv) > 2 else 200.0
qs = sorted({r[2] for r in rows})
T = max(r[1] for r in rows)
for b in range(int(T // binw) + 1):
    lo, hi = b * binw, (b + 1) * binw
    out = []
    for q in qs:
        busy, names = 0.0, collections.Counter()
        for s, e, qq, n in rows:
            if qq == q and e > lo and s < hi:
                d = min(e, hi) - max(s, lo)
                busy += d
                names[n.split("<")[0][:20]] += d
        out.append(f"{q}:{busy / binw * 100:4.0f}% {','.join(k for k, _ in names.most_common(2)):42s}")
    print(f"{lo:7.0f} " + " | ".join(out))
