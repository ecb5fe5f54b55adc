#!/usr/bin/env python3
"""Cycle accounting of the four-phase 256-square GEMM kernel (private -DHAMT_PROF build, HAMT_P8=1): per wave row and phase,
the shader cycles per k-tile spent in the load segment, at the first barrier, in the multiply segment, at the second barrier."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vln_hamt_amd import _lib as L, ops
lib = L.load()
buf = (C.c_ulonglong * 40)()
for (M, N, K) in [(5120, 3072, 768), (5120, 3072, 3072), (4096, 4096, 4096)]:
    a = torch.randn(M, K, device="cuda").bfloat16(); b = torch.randn(N, K, device="cuda").bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ops.gemm(a, b, out); lib.hamt_p8_prof_fetch(buf, 1)
    for _ in range(5): ops.gemm(a, b, out)
    lib.hamt_p8_prof_fetch(buf, 1)
    v = list(buf)
    print(f"nt {M}x{N}x{K}")
    for g in (0, 1):
        nkt = max(v[34 + g], 1)
        tot = 0
        for p in range(4):
            l, b1, m, b2 = (v[g * 16 + p * 4 + q] / nkt for q in range(4))
            tot += l + b1 + m + b2
            print(f"  wave row {g} phase {p}: load {l:6.0f}  barrier {b1:6.0f}  multiply {m:6.0f}  barrier {b2:6.0f}")
        print(f"  wave row {g}: {tot:7.0f} cycles per k-tile")
