#!/usr/bin/env python3
"""gemm_q4_kernel against the kernels it replaces: correctness vs an fp64 product and time (warm operands / cold weights and activations,
tools/cold_probe.py).  Run twice: HAMT_Q4=0 (the round-4 dispatch) and HAMT_Q4=1 (q4 whenever it can run the problem)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vln_hamt_amd import ops
from vln_hamt_amd import _lib as L
from tools.cold_probe import run

dev = "cuda"
SHAPES = ["nt:5120x768x768:bias:bf16", "nn:5120x768x768:none:bf16", "nt:5120x768x3072:bias:bf16", "nn:5120x768x3072:acc:f32", "nn:5120x768x2304:acc:f32",
          "nn:5120x768x768:acc:f32", "nt:5120x1536x768:bias:bf16", "nt:2752x768x768:bias:bf16", "nt:2752x768x3072:bias:bf16", "nn:2752x768x3072:acc:f32",
          "nn:2752x768x2304:acc:f32", "nt:2752x1536x768:bias:bf16", "nt:11520x768x768:bias:bf16", "nt:11520x768x3072:bias:bf16", "nn:11520x768x3072:acc:f32",
          "nt:384x768x3072:bias:bf16", "nt:1280x768x3072:bias:bf16", "nt:5003x760x832:bias:f32", "nn:5003x760x832:acc:f32", "nt:12800x768x3072:bias:bf16"]
for sp in (sys.argv[1:] or SHAPES):
    layout, dims, epi, cdt = sp.split(":")
    M, N, K = (int(x) for x in dims.split("x"))
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    A = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev, generator=g) * 0.05).to(torch.bfloat16)
    ref = A.double() @ W.double().t()
    b = W if layout == "nt" else W.t().contiguous()
    out = torch.full((M, N), float("nan"), device=dev, dtype=torch.float32 if cdt == "f32" else torch.bfloat16)
    kw = dict(b_kmajor=layout == "nn", prec="bf16")
    if epi == "bias":
        kw["bias"] = torch.randn(N, device=dev, generator=g); ref = ref + kw["bias"].double()
    elif epi == "acc":
        kw["epilogue"] = L.EPI_ACCUM
        base = torch.randn(M, N, device=dev, generator=g)
        out = base.clone() if cdt == "f32" else base.to(torch.bfloat16)
        ref = ref + out.double()
    ops.gemm(A, b, out, **kw)
    torch.cuda.synchronize()
    kern = L.last_kernel()
    err = float((out.double() - ref).abs().max() / ref.abs().max())
    nW = max(2, min(64, (400 << 20) // (N * K * 2) + 1)); nA = max(2, min(32, (400 << 20) // (M * K * 2) + 1))
    w = run(layout, M, N, K, epi, cdt, 1, 1)
    c = run(layout, M, N, K, epi, cdt, nW, nA)
    print(f"{sp:32s} {kern:40s} rel err {err:.2e}  warm {w:7.1f} us  cold {c:7.1f} us  ({2.0 * M * N * K / c / 1e6:6.1f} TF/s cold)", flush=True)
