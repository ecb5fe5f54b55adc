#!/usr/bin/env python3
"""Is the GEMM main loop clock (power) limited?  Same launch on random, sign-constant and zero operands (cdna_hip_programming.md 5.4 rule 25)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vln_hamt_amd import ops
def run(M, N, K, fill, iters=30):
    a = torch.randn(M, K, device="cuda"); b = torch.randn(N, K, device="cuda") * 0.05
    if fill == "zero": a.zero_(); b.zero_()
    elif fill == "abs": a.abs_(); b.abs_()
    elif fill == "small": a = torch.randint(-2, 3, (M, K), device="cuda").float(); b = torch.randint(-2, 3, (N, K), device="cuda").float()
    a, b = a.bfloat16(), b.bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        ops.gemm(a, b, out)
        with torch.cuda.graph(g, stream=st):
            for _ in range(iters): ops.gemm(a, b, out)
    torch.cuda.synchronize(); g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / iters * 1e3
    return us, 2.0 * M * N * K / us / 1e6
for shp in [(5120, 3072, 3072), (5120, 3072, 768), (4096, 4096, 4096)]:
    print(shp, "  ".join(f"{f}: {run(*shp, f)[0]:7.1f} us {run(*shp, f)[1]:6.0f} TF" for f in ("rand", "abs", "small", "zero")), flush=True)
