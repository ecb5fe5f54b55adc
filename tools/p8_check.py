#!/usr/bin/env python3
"""Correctness of the GEMM fast path against fp32 torch.matmul on the same bf16 operands, over the step's shapes and a few
ragged ones, for the epilogues the blocks use; then timings.  Run with HAMT_P8=1 to exercise the four-phase 256-square kernel."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vln_hamt_amd import ops
from vln_hamt_amd import _lib as L


def check(layout, M, N, K, epi, cdt):
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(M * 7 + N * 3 + K)
    A = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
    B = (torch.randn(N, K, device=dev, generator=g) * 0.05).to(torch.bfloat16)       # [N][K]
    ref = A.float() @ B.float().t()
    b = B if layout == "nt" else B.t().contiguous()
    out = torch.full((M, N), 7.0, device=dev, dtype=torch.float32 if cdt == "f32" else torch.bfloat16)
    kw = dict(b_kmajor=layout == "nn", prec="bf16")
    aux = None
    if epi == "bias":
        bias = torch.randn(N, device=dev, generator=g)
        kw["bias"] = bias
        ref = ref + bias
    elif epi == "acc":
        kw["epilogue"] = L.EPI_ACCUM
        base = torch.randn(M, N, device=dev, generator=g)
        out = base.clone() if cdt == "f32" else base.to(torch.bfloat16)
        ref = ref + out.float()
    elif epi == "mulaux":
        aux = torch.randn(M, N, device=dev, generator=g).to(torch.bfloat16)
        kw.update(epilogue=L.EPI_MUL_AUX, aux=aux)
        ref = ref * aux.float()
    elif epi == "gelugrad":
        bias = torch.randn(N, device=dev, generator=g)
        aux = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        kw.update(bias=bias, epilogue=L.EPI_GELU_GRAD, aux=aux)
        pre = (ref + bias).double()
        ref = torch.nn.functional.gelu(pre).float()
        phi = 0.5 * (1 + torch.erf(pre / 2 ** 0.5))
        dref = (phi + pre * torch.exp(-0.5 * pre * pre) / (2 * torch.pi) ** 0.5).float()
    ops.gemm(A, b, out, **kw)
    torch.cuda.synchronize()
    err = float((out.float() - ref).abs().max()) / max(1.0, float(ref.abs().max()))
    tol = 2e-2 if cdt == "bf16" else 2e-3
    msg = f"{layout} {M}x{N}x{K} {epi} {cdt}: err {err:.2e}"
    if epi == "gelugrad":
        derr = float((aux.float() - dref).abs().max())
        msg += f" gelu' err {derr:.2e}"
        err = max(err, derr / 2)
    print(msg + ("" if err < tol else "   <<<<<< FAIL"), flush=True)
    return err < tol


if __name__ == "__main__":
    ok = True
    for spec in [("nt", 512, 512, 128, "none", "f32"), ("nt", 512, 256, 192, "bias", "f32"), ("nt", 5120, 2304, 768, "bias", "bf16"),
                 ("nt", 5120, 3072, 768, "gelugrad", "bf16"), ("nt", 5120, 768, 3072, "bias", "f32"), ("nt", 2752, 768, 768, "bias", "f32"),
                 ("nt", 300, 520, 256, "bias", "bf16"), ("nt", 7872, 2304, 768, "bias", "bf16"),
                 ("nn", 512, 512, 256, "none", "f32"), ("nn", 5120, 768, 2304, "acc", "f32"), ("nn", 5120, 3072, 768, "mulaux", "bf16"),
                 ("nn", 2752, 768, 3072, "acc", "f32"), ("nn", 5120, 768, 768, "none", "bf16"), ("nn", 333, 768, 1024, "none", "bf16")]:
        ok &= check(*spec)
    print("ALL OK" if ok else "FAILURES")
    if ok and "--time" in sys.argv:
        os.environ["GRAPH"] = "1"
        from tools.gemm_sweep import shapes
        from tools.gemm_bench import bench
        for layout, M, N, K, epi, cdt, tag in shapes(64):
            us, tf = bench(layout, M, N, K, epi, cdt)
            print(f"{tag:12s} {layout} {M:6d} {N:5d} {K:5d} {epi:>8s} {cdt:>5s} {us:8.1f} us {tf:7.1f} TF/s", flush=True)
