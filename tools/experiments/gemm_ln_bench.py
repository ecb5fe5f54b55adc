#!/usr/bin/env python3
"""hamt_gemm_ln_fwd (dense + bias + dropout + residual + LayerNorm in one launch) against the two-launch path it replaces
(hamt_gemm with a bf16 output, then hamt_ln_fwd) at the row counts / reductions of the HAMT step.  Captured hipGraph loops, HIP
events on the launch stream, random data.   usage: gemm_ln_bench.py [MxK ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vln_hamt_amd import ops


def timed(fn, iters=20):
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(iters):
                fn()
        g.replay()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(st)
        for _ in range(3):
            g.replay()
        e.record(st)
        torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / (3 * iters)


def case(M, K, H=768, p=0.1):
    dev = "cuda"
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(H, K, device=dev) * 0.03).to(torch.bfloat16)
    b, ga, be = torch.randn(H, device=dev) * 0.1, 1 + 0.1 * torch.randn(H, device=dev), 0.1 * torch.randn(H, device=dev)
    r = torch.randn(M, H, device=dev)
    o = torch.empty(M, H, device=dev, dtype=torch.bfloat16)

    def two():
        ops.gemm(a, w, o, bias=b)
        ops._ln_fwd(o, r, ga, be, 1e-12, p, 0.0, True)

    def gemm_only():
        ops.gemm(a, w, o, bias=b)

    t2, tg = timed(two), timed(gemm_only)
    out = [f"{M:6d} {K:5d}  gemm {tg:6.1f} + ln {t2 - tg:5.1f} = {t2:6.1f} us"]
    for bm in (32, 64):
        t1 = timed(lambda: ops.gemm_ln_fwd(a, w, b, r, ga, be, 1e-12, p, tile_rows=bm))
        out.append(f"fused<{bm}> {t1:6.1f} us ({t2 / t1:4.2f}x)")
    print("   ".join(out), flush=True)


if __name__ == "__main__":
    specs = sys.argv[1:] or ["5120x768", "2752x768", "7872x768", "11520x768", "1280x768", "688x768", "2560x768", "20480x768",
                             "5120x3072", "2752x3072", "11520x3072", "1280x3072", "20480x3072"]
    for sp in specs:
        M, K = (int(x) for x in sp.split("x"))
        case(M, K)
