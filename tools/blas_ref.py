#!/usr/bin/env python3
"""Measuring stick only (never on the product path): torch.matmul (hipBLASLt) bf16 on the step's GEMM shapes."""
import torch
def t(M, N, K, iters=30):
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16); b = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
    for _ in range(5): torch.matmul(a, b.t())
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): torch.matmul(a, b.t())
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / iters * 1e3
    print(f"nt {M:6d} {N:6d} {K:6d}: {us:8.1f} us {2.0*M*N*K/us/1e6:8.1f} TFLOP/s")
for shp in [(5120, 2304, 768), (5120, 768, 768), (5120, 768, 3072), (5120, 3072, 768), (11520, 2304, 768), (2752, 768, 768), (4096, 4096, 4096), (768, 768, 5120), (3072, 768, 5120)]:
    t(*shp)
