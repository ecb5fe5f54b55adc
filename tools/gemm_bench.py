#!/usr/bin/env python3
"""Micro-benchmark of hamt_gemm (bf16 fast path) on the shapes of the HAMT step; HIP-event timing on the launch stream.
usage: gemm_bench.py [layout:MxNxK[:epi[:cdtype]] ...]   layout in nt|nn|tn, epi in none|bias|gelu|dgelu|acc"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vln_hamt_amd import ops
from vln_hamt_amd import _lib as L


def bench(layout, M, N, K, epi="bias", cdt="f32", iters=30):
    dev = "cuda"
    A = torch.randn(M, K, device=dev)
    B = torch.randn(K, N, device=dev) * 0.05
    a = (A.t().contiguous() if layout == "tn" else A).to(torch.bfloat16)
    b = (B.t().contiguous() if layout == "nt" else B).to(torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.float32 if cdt == "f32" else torch.bfloat16)
    kw = dict(a_kmajor=layout == "tn", b_kmajor=layout in ("nn", "tn"), prec="bf16")
    if os.environ.get("NOSTORE"):
        kw["alpha"] = -12345.0
    if epi == "bias":
        kw["bias"] = torch.randn(N, device=dev)
    elif epi == "gelu":
        kw.update(bias=torch.randn(N, device=dev), epilogue=L.EPI_GELU | L.EPI_SAVE_PRE, aux=torch.empty(M, N, device=dev))
    elif epi == "gelu16":
        kw.update(bias=torch.randn(N, device=dev), epilogue=L.EPI_GELU | L.EPI_SAVE_PRE, aux=torch.empty(M, N, device=dev, dtype=torch.bfloat16))
    elif epi == "dgelu":
        kw.update(epilogue=L.EPI_MUL_DGELU, aux=torch.randn(M, N, device=dev))
    elif epi == "dgelu16":
        kw.update(epilogue=L.EPI_MUL_DGELU, aux=torch.randn(M, N, device=dev).to(torch.bfloat16))
    elif epi == "gelugrad":
        kw.update(bias=torch.randn(N, device=dev), epilogue=L.EPI_GELU_GRAD, aux=torch.empty(M, N, device=dev, dtype=torch.bfloat16))
    elif epi == "gelugrad8":
        kw.update(bias=torch.randn(N, device=dev), epilogue=L.EPI_GELU_GRAD, aux=torch.empty(M, N, device=dev, dtype=torch.uint8))
    elif epi == "mulaux8":
        kw.update(epilogue=L.EPI_MUL_AUX, aux=torch.randint(0, 256, (M, N), device=dev, dtype=torch.uint8))
    elif epi == "mulaux":
        kw.update(epilogue=L.EPI_MUL_AUX, aux=torch.randn(M, N, device=dev).to(torch.bfloat16))
    elif epi == "acc":
        kw.update(epilogue=L.EPI_ACCUM)
        out.zero_()
    for _ in range(3):
        ops.gemm(a, b, out, **kw)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if os.environ.get("GRAPH"):      # replay a captured chain of launches: no host launch cost in the measurement
        g = torch.cuda.CUDAGraph()
        st = torch.cuda.Stream()
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            ops.gemm(a, b, out, **kw)
            with torch.cuda.graph(g, stream=st):
                for _ in range(iters):
                    ops.gemm(a, b, out, **kw)
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        s.record()
        g.replay()
        e.record()
    else:
        s.record()
        for _ in range(iters):
            ops.gemm(a, b, out, **kw)
        e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) / iters * 1e3
    return us, 2.0 * M * N * K / us / 1e6


if __name__ == "__main__":
    specs = sys.argv[1:] or [
        "nt:5120x768x768:bias", "nt:5120x2304x768:bias:bf16", "nt:5120x3072x768:gelu:bf16", "nt:5120x3072x768:gelu16:bf16",
        "nt:5120x768x3072:bias", "nt:11520x3072x768:gelu:bf16", "nt:11520x768x3072:bias",
        "nn:5120x768x768:none:bf16", "nn:5120x768x2304:acc", "nn:5120x3072x768:dgelu:bf16", "nn:5120x3072x768:dgelu16:bf16",
        "nn:5120x768x3072:acc", "nn:11520x3072x768:dgelu:bf16",
        "tn:768x768x5120:none", "tn:2304x768x5120:none", "tn:3072x768x5120:none", "tn:768x3072x5120:none",
        "tn:768x768x11520:none", "tn:3072x768x11520:none", "tn:768x768x2752:none", "tn:30522x768x768:none",
        "nt:4096x4096x4096:none", "nn:4096x4096x4096:none", "tn:4096x4096x4096:none"]
    print(f"{'layout':6} {'M':>6} {'N':>6} {'K':>6} {'epi':>8} {'C':>5} {'us':>9} {'TFLOP/s':>9}")
    for sp in specs:
        parts = sp.split(":")
        layout = parts[0]
        M, N, K = (int(x) for x in parts[1].split("x"))
        epi = parts[2] if len(parts) > 2 else "none"
        cdt = parts[3] if len(parts) > 3 else "f32"
        us, tf = bench(layout, M, N, K, epi, cdt)
        print(f"{layout:6} {M:6d} {N:6d} {K:6d} {epi:>8} {cdt:>5} {us:9.1f} {tf:9.1f}")
