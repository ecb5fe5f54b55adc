#!/usr/bin/env python3
"""Micro-benchmark of hamt_gemm (bf16 fast path) on the shapes of the HAMT step; HIP-event timing on the launch stream."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vln_hamt_amd import ops


def bench(M, N, K, out_dtype=torch.float32, bias=True, iters=50):
    a = (torch.randn(M, K, device="cuda")).to(torch.bfloat16)
    b = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    bs = torch.randn(N, device="cuda") if bias else None
    out = torch.empty(M, N, device="cuda", dtype=out_dtype)
    for _ in range(5):
        ops.gemm(a, b, out, bias=bs, prec="bf16")
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        ops.gemm(a, b, out, bias=bs, prec="bf16")
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) / iters * 1e3
    return us, 2.0 * M * N * K / us / 1e6


if __name__ == "__main__":
    shapes = [(5120, 768, 64), (5120, 768, 256), (5120, 768, 768), (5120, 768, 3072), (5120, 3072, 768), (5120, 2304, 768),
              (11520, 768, 768), (11520, 3072, 768), (11520, 768, 3072), (2752, 768, 768), (768, 768, 5120), (3072, 768, 5120),
              (768, 3072, 5120), (768, 768, 11520), (768, 30522, 768), (4096, 4096, 4096), (8192, 8192, 1024)]
    if len(sys.argv) > 1:
        shapes = [tuple(int(x) for x in s.split("x")) for s in sys.argv[1:]]
    print(f"{'M':>6} {'N':>6} {'K':>6} {'us':>9} {'TFLOP/s':>9}   (fp32 C)      us   TF (bf16 C)")
    for (M, N, K) in shapes:
        us, tf = bench(M, N, K)
        us2, tf2 = bench(M, N, K, torch.bfloat16)
        print(f"{M:6d} {N:6d} {K:6d} {us:9.1f} {tf:9.1f}            {us2:9.1f} {tf2:7.1f}")
