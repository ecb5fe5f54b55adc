#!/usr/bin/env python3
"""Per-shape sweep of hamt_gemm over the GEMMs of one HAMT step (B = 64 by default), hipGraph-replayed chains so that the
numbers are kernel time, next to torch.matmul (hipBLASLt; measuring stick only, never on the product path).
usage: gemm_sweep.py [--batch 64] [--blas]      (HAMT_FAST_BM=64|128|256 forces the tile height, read once per process)"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GRAPH", "1")
import torch

from tools.gemm_bench import bench


def shapes(B):
    T, P, V = B * 80, B * 5 * 36, B * 43
    out = []
    for M, tag in ((T, "text"), (P, "pano"), (V, "visn"), (T + V, "cross")):
        out += [("nt", M, 2304, 768, "bias", "bf16", tag + " qkv"), ("nt", M, 768, 768, "bias", "f32", tag + " out"),
                ("nt", M, 3072, 768, "gelugrad", "bf16", tag + " ffn1"), ("nt", M, 768, 3072, "bias", "f32", tag + " ffn2"),
                ("nn", M, 768, 768, "none", "bf16", tag + " d_out"), ("nn", M, 768, 2304, "acc", "f32", tag + " d_qkv"),
                ("nn", M, 3072, 768, "mulaux", "bf16", tag + " d_ffn2"), ("nn", M, 768, 3072, "acc", "f32", tag + " d_ffn1")]
    return out


def blas(M, N, K, iters=30):
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    b = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
    o = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        torch.matmul(a, b.t(), out=o)
        with torch.cuda.graph(g, stream=st):
            for _ in range(iters):
                torch.matmul(a, b.t(), out=o)
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--blas", action="store_true")
    a = ap.parse_args()
    print(f"# B={a.batch} HAMT_FAST_BM={os.environ.get('HAMT_FAST_BM', 'auto')}")
    tot = 0.0
    for layout, M, N, K, epi, cdt, tag in shapes(a.batch):
        us, tf = bench(layout, M, N, K, epi, cdt)
        line = f"{tag:12s} {layout} {M:6d} {N:5d} {K:5d} {epi:>8s} {cdt:>5s} {us:8.1f} us {tf:7.1f} TF/s"
        if a.blas:
            ub = blas(M, N, K)
            line += f" | blas {ub:8.1f} us {2.0 * M * N * K / ub / 1e6:7.1f} TF/s"
        print(line, flush=True)
        tot += us
    print(f"# sum {tot:.1f} us")
