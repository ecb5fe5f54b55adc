#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (--kernel-trace --stats) into a per-kernel table (text, for profiles/)."""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"(gemm_bf16_kernel<[^>]*>|gemm_f32_kernel<[^>]*>|[\w:]+(<[^(]{0,60})?)", name)
    return (m.group(1) if m else name)[:90]


def main(path, top=45):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    rows = c.execute("select name, start, end from kernels").fetchall()
    agg = {}
    t0, t1 = min(r[1] for r in rows), max(r[2] for r in rows)
    for n, s, e in rows:
        k = short(n)
        a = agg.setdefault(k, [0, 0])
        a[0] += 1
        a[1] += e - s
    tot = sum(a[1] for a in agg.values())
    print(f"# kernels: {len(rows)} dispatches, {len(agg)} distinct; GPU busy {tot/1e6:.2f} ms of {(t1-t0)/1e6:.2f} ms wall ({100*tot/(t1-t0):.1f} %)")
    print(f"{'kernel':90s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'pct':>6s}")
    for k, (n, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"{k:90s} {n:7d} {d/1e6:10.3f} {d/n/1e3:9.2f} {100*d/tot:6.2f}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 45)
