#!/usr/bin/env python3
"""Micro-benchmark of hamt_ln_fwd / hamt_ln_bwd (fused dropout + residual + LayerNorm) on the HAMT row counts."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vln_hamt_amd import ops

def t(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        fn()
        with torch.cuda.graph(g, stream=st):
            for _ in range(iters): fn()
    torch.cuda.synchronize(); g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

X16 = os.environ.get("BF16") == "1"      # the bf16 dense -> LayerNorm interface of the step (bf16 x, bf16 saved sum)
for M in (5120, 4608, 11520, 2368, 384):
    for p in (0.1, 0.0):
        H = 768
        x = torch.randn(M, H, device="cuda"); x = x.to(torch.bfloat16) if X16 else x; r = torch.randn(M, H, device="cuda")
        g = torch.ones(H, device="cuda"); b = torch.zeros(H, device="cuda")
        y, y16, z, mean, rstd, cid = ops._ln_fwd(x, r, g, b, 1e-12, p, 0.0, True)
        dy = torch.randn(M, H, device="cuda")
        f = t(lambda: ops._ln_fwd(x, r, g, b, 1e-12, p, 0.0, True))
        bw = t(lambda: ops._ln_bwd(dy, z, mean, rstd, g, 1e-12, p, 0.0, cid, False, True, True))
        fb = M * H * (4 + 4 + 4 + 4 + 2) / 1e6; bb = M * H * (4 + 4 + 4 + 2) / 1e6
        print(f"M={M:6d} p={p}: fwd {f:6.1f} us ({fb / f * 1e-3 * 1e3:5.2f} TB/s)  bwd {bw:6.1f} us ({bb / bw:5.2f} TB/s)")
