#!/usr/bin/env python3
"""How much of a GEMM's in-step time is operand temperature?  The per-shape sweeps re-issue ONE problem (operands resident in the 256 MB
Infinity Cache after the first pass); in the step every layer's weights arrive cold from HBM (the bf16 shadow arena is 350 MB and is rewritten
by the update) and the activations were written by the previous kernel.  Here: a captured chain of N launches of one shape, with
  warm:  the same A and W every launch            cold-W: a different W per launch (N copies, > 256 MB in total)
  cold-AW: different W and different A per launch
usage: cold_probe.py [layout:MxNxK:epi:cdt ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vln_hamt_amd import ops
from vln_hamt_amd import _lib as L

dev = "cuda"


def run(layout, M, N, K, epi, cdt, nW, nA, iters=64):
    Ws = [(torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16) if layout == "nt" else (torch.randn(K, N, device=dev) * 0.05).to(torch.bfloat16) for _ in range(nW)]
    As = [torch.randn(M, K, device=dev).to(torch.bfloat16) for _ in range(nA)]
    out = torch.zeros(M, N, device=dev, dtype=torch.float32 if cdt == "f32" else torch.bfloat16)
    kw = dict(b_kmajor=layout == "nn", prec="bf16")
    if epi == "bias":
        kw["bias"] = torch.randn(N, device=dev)
    elif epi == "acc":
        kw["epilogue"] = L.EPI_ACCUM
    elif epi == "gelugrad":
        kw.update(bias=torch.randn(N, device=dev), epilogue=L.EPI_GELU_GRAD, aux=torch.empty(M, N, device=dev, dtype=torch.bfloat16))
    elif epi == "mulaux":
        kw.update(epilogue=L.EPI_MUL_AUX, aux=torch.randn(M, N, device=dev).to(torch.bfloat16))
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        ops.gemm(As[0], Ws[0], out, **kw)
        with torch.cuda.graph(g, stream=st):
            for i in range(iters):
                ops.gemm(As[i % nA], Ws[i % nW], out, **kw)
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        s.record(); g.replay(); e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / iters * 1e3)
    return best


if __name__ == "__main__":
    specs = sys.argv[1:] or ["nt:5120x768x3072:bias:bf16", "nn:5120x768x3072:acc:f32", "nn:5120x768x2304:acc:f32", "nt:5120x768x768:bias:bf16", "nn:5120x768x768:none:bf16",
                             "nt:5120x2304x768:bias:bf16", "nt:5120x3072x768:gelugrad:bf16", "nn:5120x3072x768:mulaux:bf16"]
    print(f"{'shape':34s} {'warm':>8s} {'cold-W':>8s} {'cold-AW':>8s}   (us per launch; kernel: {''})")
    for sp in specs:
        layout, dims, epi, cdt = sp.split(":")
        M, N, K = (int(x) for x in dims.split("x"))
        wbytes = N * K * 2
        nW = max(2, min(64, (400 << 20) // wbytes + 1))
        abytes = M * K * 2
        nA = max(2, min(64, (400 << 20) // abytes + 1))
        w = run(layout, M, N, K, epi, cdt, 1, 1)
        cw = run(layout, M, N, K, epi, cdt, nW, 1)
        caw = run(layout, M, N, K, epi, cdt, nW, nA)
        print(f"{sp:34s} {w:8.1f} {cw:8.1f} {caw:8.1f}   nW={nW} nA={nA}  {L.last_kernel()}", flush=True)
