#!/usr/bin/env python3
"""Decompose the GEMM time at one output shape into fixed (prologue + epilogue) and per-k-tile cost: time vs K for several epilogues."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["GRAPH"] = "1"
from tools.gemm_bench import bench
M, N = (int(x) for x in (sys.argv[1:3] if len(sys.argv) > 2 else (5120, 3072)))
for layout, epi, cdt in [("nt", "none", "bf16"), ("nt", "none", "f32"), ("nt", "bias", "bf16"), ("nt", "gelugrad", "bf16"), ("nn", "mulaux", "bf16"), ("nn", "acc", "f32")]:
    row = []
    for K in (256, 768, 1536, 3072):
        us, tf = bench(layout, M, N, K, epi, cdt)
        row.append((K, us, tf))
    (k1, u1, _), (k2, u2, _) = row[1], row[3]
    per = (u2 - u1) / ((k2 - k1) / 64)
    print(f"{layout} {M}x{N} {epi:>8s} {cdt}: " + "  ".join(f"K={k}: {u:6.1f} us ({t:5.0f} TF)" for k, u, t in row) + f"  | per k-tile {per*1e3:6.0f} ns, fixed {u1 - per * k1 / 64:5.1f} us", flush=True)
