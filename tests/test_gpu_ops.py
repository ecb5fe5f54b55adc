"""-m gpu: every HIP kernel behind the C-ABI against a plain PyTorch fp32/fp64 CPU reference of the same op.

Tolerances: HAMT_PREC_F32 paths (exact fp32 MFMA) <= 2e-5 relative to the output scale; HAMT_PREC_BF16 is
checked twice -- tightly (2e-5) against a reference fed the SAME bf16-rounded operands (pins indexing/layout
bit-for-bit up to fp32 summation order) and loosely (2e-2 of scale) against the un-rounded fp32 reference.
"""
import math
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _ops():
    from vln_hamt_amd import ops
    return ops


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def bf16_round(x):
    return x.to(torch.bfloat16).to(torch.float32)


def close(a, b, tol, what=""):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = max(1.0, float(b.abs().max()))
    err = float((a - b).abs().max())
    assert err <= tol * scale, f"{what}: max|d|={err:.3e} scale={scale:.3e} tol={tol}"


# ------------------------------------------------------------------------------------------ GEMM
LAYOUTS = {"nt": (False, False), "nn": (False, True), "tn": (True, True)}
SHAPES = [(200, 136, 96), (130, 70, 36), (64, 3, 768), (33, 1, 40), (50, 64, 4), (77, 100, 1001), (2048, 1536, 64),
          (300, 200, 768), (129, 1000, 128), (1, 130, 64)]


def _mk(layout, M, N, K, seed):
    a_km, b_km = LAYOUTS[layout]
    A = rnd(M, K, seed=seed)          # logical [M,K]
    B = rnd(K, N, seed=seed + 1)      # logical [K,N]
    a_st = A.t().contiguous() if a_km else A.contiguous()        # stored
    b_st = B.contiguous() if b_km else B.t().contiguous()
    return A, B, a_st, b_st, a_km, b_km


@pytest.mark.parametrize("layout", list(LAYOUTS))
@pytest.mark.parametrize("shape", SHAPES)
def test_gemm_fp32_exact(layout, shape):
    ops = _ops()
    M, N, K = shape
    A, B, a_st, b_st, a_km, b_km = _mk(layout, M, N, K, 1)
    out = torch.empty(M, N, device=DEV)
    ops.gemm(a_st.to(DEV), b_st.to(DEV), out, a_kmajor=a_km, b_kmajor=b_km, prec="fp32")
    close(out, A.double() @ B.double(), 2e-5 * math.sqrt(K) / 4, f"gemm fp32 {layout} {shape}")


@pytest.mark.parametrize("layout", list(LAYOUTS))
@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("dts", ["ff", "fb", "bf", "bb"])
def test_gemm_bf16(layout, shape, dts):
    ops = _ops()
    M, N, K = shape
    A, B, a_st, b_st, a_km, b_km = _mk(layout, M, N, K, 2)
    a_dev = a_st.to(DEV).to(torch.bfloat16) if dts[0] == "b" else a_st.to(DEV)
    b_dev = b_st.to(DEV).to(torch.bfloat16) if dts[1] == "b" else b_st.to(DEV)
    out = torch.empty(M, N, device=DEV)
    ops.gemm(a_dev, b_dev, out, a_kmajor=a_km, b_kmajor=b_km, prec="bf16")
    close(out, bf16_round(A).double() @ bf16_round(B).double(), 2e-5 * math.sqrt(K) / 4, f"gemm bf16(rounded ref) {layout} {shape} {dts}")
    close(out, A.double() @ B.double(), 2e-2, f"gemm bf16(fp32 ref) {layout} {shape}")


@pytest.mark.parametrize("layout", ["nn", "tn"])
@pytest.mark.parametrize("shape", [(300, 200, 768), (129, 1000, 128), (768, 768, 5120), (70, 3072, 640), (200, 30522 // 8, 256)])
def test_gemm_fast_kstrided_layouts(layout, shape):
    """bf16 x bf16 NN / TN through the glds + ds_read_b64_tr_b16 fast kernel (incl. ragged reduction rows)."""
    ops = _ops()
    M, N, K = shape
    A, B, a_st, b_st, a_km, b_km = _mk(layout, M, N, K, 11)
    ref = bf16_round(A).double() @ bf16_round(B).double()
    out = torch.empty(M, N, device=DEV)
    ops.gemm(a_st.to(DEV).to(torch.bfloat16), b_st.to(DEV).to(torch.bfloat16), out, a_kmajor=a_km, b_kmajor=b_km, prec="bf16")
    close(out, ref, 2e-5 * math.sqrt(K) / 4, f"fast {layout} {shape}")
    # ragged reduction: K-strided B stores only K-5 rows while the (zero padded) A defines K
    if layout == "nn":
        Az = A.clone()
        Az[:, K - 5:] = 0
        ops.gemm(Az.to(DEV).to(torch.bfloat16), b_st[:K - 5].contiguous().to(DEV).to(torch.bfloat16), out, b_kmajor=True, prec="bf16", k_red=K)
        close(out, bf16_round(Az).double() @ bf16_round(B).double(), 2e-5 * math.sqrt(K) / 4, "fast nn ragged K")


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_gemm_epilogues(prec):
    ops = _ops()
    from vln_hamt_amd import _lib as L
    M, N, K = 150, 200, 64
    A, W = rnd(M, K, seed=3), rnd(N, K, seed=4)
    bias = rnd(N, seed=5)
    Ar, Wr = (bf16_round(A), bf16_round(W)) if prec == "bf16" else (A, W)
    lin = Ar.double() @ Wr.double().t() + bias.double()
    a, w, b = A.to(DEV), W.to(DEV), bias.to(DEV)
    out, pre = torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV)
    ops.gemm(a, w, out, bias=b, epilogue=L.EPI_GELU | L.EPI_SAVE_PRE, aux=pre, prec=prec)
    close(pre, lin, 3e-5, "save_pre")
    close(out, F.gelu(lin), 3e-5, "gelu")
    ops.gemm(a, w, out, bias=b, epilogue=L.EPI_RELU, prec=prec)
    close(out, torch.relu(lin), 3e-5, "relu")
    base = rnd(M, N, seed=6)
    out = base.to(DEV).clone()
    ops.gemm(a, w, out, epilogue=L.EPI_ACCUM, prec=prec, alpha=0.5)
    close(out, base.double() + 0.5 * (Ar.double() @ Wr.double().t()), 3e-5, "accum+alpha")
    h = rnd(M, N, seed=7)
    hd = h.to(DEV)
    ops.gemm(a, w, out, epilogue=L.EPI_MUL_DGELU, aux=hd, prec=prec)
    hh = h.double().requires_grad_(True)
    F.gelu(hh).sum().backward()
    close(out, (Ar.double() @ Wr.double().t()) * hh.grad, 3e-5, "mul_dgelu")
    # strided output / operand views (packed qkv layout) and bf16 output
    packed = torch.zeros(M, 3 * N, device=DEV)
    ops.gemm(a, w, packed[:, N:2 * N], bias=b, prec=prec)
    close(packed[:, N:2 * N], lin, 3e-5, "strided C")
    assert float(packed[:, :N].abs().max()) == 0.0 and float(packed[:, 2 * N:].abs().max()) == 0.0
    o16 = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    ops.gemm(a, w, o16, bias=b, prec=prec)
    close(o16.float(), lin, 1e-2, "bf16 C")


@pytest.mark.parametrize("layout,shape,variant", [
    ("nt", (1280, 768, 3072), "gemm_kg_kernel<32, 2, 4"), ("nn", (1280, 768, 2304), "gemm_kg_kernel<32, 2, 4"),
    ("nt", (1968, 768, 3072), "gemm_kg_kernel<64, 3, 2"), ("nn", (1968, 768, 2304), "gemm_kg_kernel<64, 3, 2"),
    ("nt", (688, 768, 768), "gemm_kg_kernel<32, 2, 4"), ("nt", (100, 130, 1536), "gemm_kg_kernel<32, 2, 4"),
    ("nn", (1301, 700, 1600), "gemm_kg_kernel<32, 2, 4"), ("nn", (2500, 520, 1664), "gemm_kg_kernel<64, 3, 2"),
    ("nt", (5120, 768, 3072), "gemm_kg_kernel<128, 2, 2"), ("nn", (2752, 768, 2304), "gemm_kg_kernel<128, 2, 2"),
    ("nn", (4001, 1000, 2368), "gemm_kg_kernel<128, 2, 2")])
def test_gemm_k_groups(layout, shape, variant):
    """Small grids with a long reduction run as K groups inside one workgroup (gemm_kg_kernel): every k-tile residue class,
    ragged edges, the run-time epilogue (bias, accumulate, bf16 output with an aux multiply, dropout + residual)."""
    ops = _ops()
    from vln_hamt_amd import _lib as L
    M, N, K = shape
    A, B, a_st, b_st, a_km, b_km = _mk(layout, M, N, K, 21)
    a16, b16 = a_st.to(DEV).to(torch.bfloat16), b_st.to(DEV).to(torch.bfloat16)
    ref = bf16_round(A).double() @ bf16_round(B).double()
    tol = 2e-5 * math.sqrt(K) / 4
    out = torch.empty(M, N, device=DEV)
    ops.gemm(a16, b16, out, b_kmajor=b_km, prec="bf16")
    # (about one round of 128-square tiles with a plain / bias / accumulate epilogue: the four-deep 128-square tile, gemm_q4.hip;
    # the K-group tile of that size keeps the other epilogues)
    assert L.last_kernel().startswith("gemm_q4_kernel" if variant.startswith("gemm_kg_kernel<128") else variant), L.last_kernel()
    close(out, ref, tol, f"kg plain {layout} {shape}")
    bias = rnd(N, seed=5)
    base = rnd(M, N, seed=6)
    out = base.to(DEV).clone()
    ops.gemm(a16, b16, out, b_kmajor=b_km, bias=bias.to(DEV), epilogue=L.EPI_ACCUM, prec="bf16", alpha=0.5)
    assert L.last_kernel().startswith(variant), L.last_kernel()
    close(out, base.double() + 0.5 * ref + bias.double(), tol, "kg bias + accum + alpha")
    h = bf16_round(rnd(M, N, seed=7))
    o16 = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    ops.gemm(a16, b16, o16, b_kmajor=b_km, epilogue=L.EPI_MUL_AUX, aux=h.to(DEV).to(torch.bfloat16), prec="bf16")
    assert L.last_kernel().startswith(variant), L.last_kernel()
    close(o16.float(), ref * h.double(), 1e-2, "kg mul_aux bf16")
    # dropout + residual: same mask stream as the plain-tile kernels (HAMT_KG=1 would run those) -> compare with the mask replayed
    res = rnd(M, N, seed=8).to(DEV)
    ops.gemm(a16, b16, out, b_kmajor=b_km, bias=bias.to(DEV), epilogue=L.EPI_ADD_AUX, aux=res, prec="bf16", drop=(0.25, 77))
    y = (out - res).cpu().double()
    lin = ref + bias.double()
    kept = y != 0
    frac = float(kept.double().mean())
    assert abs(frac - 0.75) < 0.02, frac
    close(torch.where(kept, y, torch.zeros_like(y)), torch.where(kept, lin / 0.75, torch.zeros_like(lin)), 3 * tol, "kg dropout + residual")
    if layout == "nn":      # ragged reduction: B stores K - 5 rows, A is zero there
        Az = A.clone()
        Az[:, K - 5:] = 0
        ops.gemm(Az.to(DEV).to(torch.bfloat16), b_st[:K - 5].contiguous().to(DEV).to(torch.bfloat16), out, b_kmajor=True, prec="bf16", k_red=K)
        assert L.last_kernel().startswith("gemm_q4_kernel" if variant.startswith("gemm_kg_kernel<128") else variant), L.last_kernel()
        close(out, bf16_round(Az).double() @ bf16_round(B).double(), tol, "kg nn ragged K")


# The step's own GEMM shapes at the benchmarked per-GPU batch (B = 64: text 5120 rows, panorama 11520, text + vision 7872 ...),
# each on the kernel the launcher picks for it there, against fp64 products of the same bf16-rounded operands.  The smaller
# SHAPES above never reach the 256-square two-phase tile (gemm_p8_kernel needs >= ~200 tiles and K >= 128) nor the 128-row plain
# tile; round 2 shipped a NaN in exactly that family which only a training soak saw.
BENCH_GEMMS = [   # (layout, M, N, K, epilogue, C dtype, kernel the launcher must pick)
    ("nt", 5120, 2304, 768, "bias", "bf16", "gemm_p8_kernel<1, false, false, 3>"),       # text QKV: 240 tiles of 256 x 192 = one full round
    ("nt", 7872, 2304, 768, "bias", "bf16", "gemm_fast_kernel<128, 1, false, false>"),   # packed QKV over text + vision rows: two rounds of either 256-row tile -> plain 128-row tiles
    ("nt", 11520, 2304, 768, "bias", "bf16", "gemm_p8_kernel<1, false, false, 4>"),         # panorama QKV
    ("nt", 5120, 3072, 768, "gelugrad", "bf16", "gemm_p8_kernel<129, false, false, 4>"),    # text FFN-1 (+ saved gelu')
    ("nt", 11520, 3072, 768, "gelugrad", "bf16", "gemm_p8_kernel<129, false, false, 4>"),   # panorama FFN-1
    ("nt", 5003, 2304, 768, "bias", "bf16", "gemm_p8_kernel<1, false, false, 3>"),       # ragged last tile row
    ("nt", 5120, 2304, 768, "gelugrad", "bf16", "gemm_p8_kernel<129, false, false, 3>"), # (FFN-1's epilogue on the narrow tile)
    ("nt", 5120, 2250, 768, "bias", "f32", "gemm_p8_kernel<1, false, false, 3>"),        # ragged last 192-column tile (138 columns), fp32 C
    ("nt", 5120, 2318, 768, "bias", "f32", "gemm_p8_kernel<1, false, false, 4>"),           # ragged last tile column, fp32 C
    ("nt", 11520, 768, 3072, "bias", "bf16", "gemm_p8_kernel<1, false, false, 3>"),      # panorama FFN-2
    ("nt", 11520, 768, 3072, "drop_res", "f32", "gemm_p8_kernel<1537, false, false, 4>"),   # bias + dropout + residual (pre-LN ViT form)
    ("nn", 5120, 3072, 768, "mulaux", "bf16", "gemm_p8_kernel<256, false, true, 4>"),       # dgrad of FFN-2 x gelu'
    ("nn", 11520, 3072, 768, "mulaux", "bf16", "gemm_p8_kernel<256, false, true, 3>"),
    ("nn", 11520, 768, 3072, "acc", "f32", "gemm_p8_kernel<8, false, true, 3>"),         # dgrad of FFN-1 into the residual gradient
    ("nn", 11520, 768, 2304, "acc", "f32", "gemm_p8_kernel<8, false, true, 3>"),         # dgrad of QKV into the residual gradient
    ("nn", 5120, 2304, 768, "none", "bf16", "gemm_p8_kernel<0, false, true, 3>"),
    ("nn", 5013, 2248, 768, "mulaux", "bf16", "gemm_p8_kernel<256, false, true, 3>"),    # ragged rows and a ragged 192-column tile, K-strided B
    ("nn", 5009, 3072, 768, "mulaux", "bf16", "gemm_p8_kernel<256, false, true, 4>"),       # ragged rows
    ("nt", 5120, 768, 768, "bias", "bf16", "gemm_q4_kernel<1, false>"),                  # attention output projection: 240 tiles of 128 x 128, four-deep ring
    ("nn", 5120, 768, 768, "none", "bf16", "gemm_q4_kernel<0, true>"),                   # its dgrad
    ("nt", 2752, 768, 768, "bias", "f32", "gemm_q4_kernel<1, false>"),                   # vision stream (132 tiles)
    ("nt", 1280, 768, 768, "bias", "bf16", "gemm_kg_kernel<32, 2, 4, false>"),           # (B = 16: too few 128-square tiles -- K groups on 32-row tiles)
    ("nn", 1280, 768, 768, "none", "bf16", "gemm_kg_kernel<32, 2, 4, true>"),
    ("nt", 5120, 1536, 768, "bias", "bf16", "gemm_fast_kernel<128, 1, false, false>"),   # key / value projection of the cross attention (N = 1536: not the q4 tile's)
    ("nt", 5003, 760, 832, "bias", "f32", "gemm_q4_kernel<1, false>"),                   # ragged rows and columns
    ("nn", 5003, 760, 832, "acc", "f32", "gemm_q4_kernel<8, true>"),
    ("nn", 5120, 768, 2304, "acc", "f32", "gemm_q4_kernel<8, true>"),                    # dgrad of QKV into the residual gradient
    ("nt", 3200, 768, 192, "bias", "bf16", "gemm_q4_kernel<1, false>"),                  # the shortest reduction the ring takes (3 k-tiles)
    ("nt", 3200, 768, 256, "none", "bf16", "gemm_q4_kernel<0, false>"),
    ("nt", 4096, 4096, 64, "bias", "bf16", "gemm_fast_kernel<128, 1, false, false>"),    # 128-row plain tile
    ("nn", 4096, 4096, 64, "acc", "f32", "gemm_fast_kernel<128, 8, false, true>"),
    ("nt", 5120, 3072, 768, "gelugrad8", "bf16", "gemm_p8_kernel<129, false, false, 4>"),   # text FFN-1, gelu' saved as one byte (HAMT_U8G)
    ("nt", 2752, 3072, 768, "gelugrad8", "bf16", None),                                      # vision stream
    ("nt", 389, 3068, 768, "gelugrad8", "bf16", None),                                       # ragged rows / columns: byte-wise tail of the code image
    ("nn", 5120, 3072, 768, "mulaux8", "bf16", "gemm_p8_kernel<256, false, true, 4>"),       # dgrad of FFN-2 x the one-byte gelu'
    ("nn", 11520, 3072, 768, "mulaux8", "bf16", "gemm_p8_kernel<256, false, true, 3>"),
    ("nn", 2752, 3072, 768, "mulaux8", "bf16", None),
    ("nn", 389, 3068, 768, "mulaux8", "bf16", None),
    ("nt", 5120, 768, 3072, "bias", "bf16", "gemm_q4_kernel<1, false>"),                  # text FFN-2
    ("nn", 5120, 768, 3072, "acc", "f32", "gemm_q4_kernel<8, true>"),
]


@pytest.mark.parametrize("layout,M,N,K,epi,cdt,kernel", BENCH_GEMMS, ids=[f"{s[0]}-{s[1]}x{s[2]}x{s[3]}-{s[4]}-{s[5]}" for s in BENCH_GEMMS])
def test_gemm_bench_shapes_vs_fp64(layout, M, N, K, epi, cdt, kernel):
    ops = _ops()
    from vln_hamt_amd import _lib as L
    g = torch.Generator(device=DEV).manual_seed(M * 7 + N * 3 + K)
    A = torch.randn(M, K, device=DEV, generator=g).to(torch.bfloat16)
    W = (torch.randn(N, K, device=DEV, generator=g) * 0.05).to(torch.bfloat16)         # logical [N][K]
    ref = A.double() @ W.double().t()
    b = W if layout == "nt" else W.t().contiguous()
    cd = torch.float32 if cdt == "f32" else torch.bfloat16
    out = torch.full((M, N), float("nan"), device=DEV, dtype=cd)                          # an unwritten element shows
    kw = dict(b_kmajor=layout == "nn", prec="bf16")
    aux = dref = keep_frac = None
    if epi in ("bias", "gelugrad", "gelugrad8", "drop_res"):
        bias = torch.randn(N, device=DEV, generator=g)
        kw["bias"] = bias
        ref = ref + bias.double()
    if epi == "acc":
        kw["epilogue"] = L.EPI_ACCUM
        base = torch.randn(M, N, device=DEV, generator=g)
        out = base.clone() if cdt == "f32" else base.to(torch.bfloat16)
        ref = ref + out.double()
    elif epi == "mulaux":
        aux = torch.randn(M, N, device=DEV, generator=g).to(torch.bfloat16)
        kw.update(epilogue=L.EPI_MUL_AUX, aux=aux)
        ref = ref * aux.double()
    elif epi == "mulaux8":      # aux = codes q of the one-byte gelu' image: value 0.005 q - 0.13 (hamt.h HAMT_U8G)
        aux = torch.randint(0, 256, (M, (N + 15) // 16 * 16), device=DEV, generator=g, dtype=torch.uint8)[:, :N]
        kw.update(epilogue=L.EPI_MUL_AUX, aux=aux)
        ref = ref * (aux.double() * 0.005 - 0.13)
    elif epi in ("gelugrad", "gelugrad8"):
        if epi == "gelugrad8":
            aux = torch.full((M, (N + 15) // 16 * 16), 255, device=DEV, dtype=torch.uint8)[:, :N]
        else:
            aux = torch.full((M, N), float("nan"), device=DEV, dtype=torch.bfloat16)
        kw.update(epilogue=L.EPI_GELU_GRAD, aux=aux)
        pre = ref
        phi = 0.5 * (1 + torch.erf(pre / 2 ** 0.5))
        ref = pre * phi
        dref = phi + pre * torch.exp(-0.5 * pre * pre) / (2 * math.pi) ** 0.5
    elif epi == "drop_res":
        res = torch.randn(M, N, device=DEV, generator=g)
        kw.update(epilogue=L.EPI_ADD_AUX, aux=res, drop=(0.1, 4242))
    ops.gemm(A, b, out, **kw)
    assert kernel is None or L.last_kernel() == kernel, L.last_kernel()
    torch.cuda.synchronize()
    assert bool(torch.isfinite(out).all()), "non-finite / unwritten output elements"
    # fp32 accumulation of K bf16 products: 2e-5 sqrt(K)/4 of the scale (as the small-shape tests); + bf16 rounding of a bf16 C
    tol = 2e-5 * math.sqrt(K) / 4 + (2 ** -8 if cdt == "bf16" else 0.0)
    if epi == "drop_res":
        y = out.double() - res.double()
        kept = y != 0
        keep_frac = float(kept.double().mean())
        assert abs(keep_frac - 0.9) < 2e-3, keep_frac
        close(torch.where(kept, y, torch.zeros_like(y)), torch.where(kept, ref / 0.9, torch.zeros_like(ref)), 3 * tol, f"{kernel} dropout + residual")
    else:
        close(out, ref, tol, f"{kernel} {layout} {M}x{N}x{K} {epi}")
    if dref is not None and epi == "gelugrad8":
        # code of the value the kernel's fp32 epilogue saw: nearest of 0.005 q - 0.13 (half a step = 0.0025, + the fp32 pre-activation's error)
        err = float((aux.double() * 0.005 - 0.13 - dref).abs().max())
        assert err <= 0.0025 + 2 * tol, f"{kernel} one-byte gelu': {err}"
    elif dref is not None:
        assert bool(torch.isfinite(aux).all())
        close(aux, dref, 2 ** -7, f"{kernel} saved gelu'")


def _ref_ln(x, g, b, eps):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * g + b


def _ref_attn(q, k, v, add_mask, heads):
    B, Sq, H = q.shape
    Sk, dh = k.shape[1], H // heads
    qh, kh, vh = (t.view(B, -1, heads, dh).transpose(1, 2) for t in (q, k, v))
    s = qh @ kh.transpose(-1, -2) / math.sqrt(dh)
    if add_mask is not None:
        s = s + add_mask.view(B, 1, 1, Sk)
    return (torch.softmax(s, -1) @ vh).transpose(1, 2).reshape(B, Sq, H)


def _grad_report(named, ref_grads, what, cos_min, probe_tol):
    """bf16 block gradients against fp64 autograd: global cosine over all parameters + per-tensor relative error."""
    num = den_a = den_b = 0.0
    # (a key bias has NO gradient mathematically -- softmax is invariant to a per-query shift of the scores -- so errors are taken
    # relative to max(||ref||, 5 % of the largest gradient norm among tensors of the same rank))
    gmax = {d: max(float(rg.norm()) for (_, p), rg in zip(named, ref_grads) if p.dim() == d) for d in {p.dim() for _, p in named}}
    for (name, p), rg in zip(named, ref_grads):
        assert p.grad is not None, f"{what}: {name} has no gradient"
        g = p.grad.double()
        assert bool(torch.isfinite(g).all()), f"{what}: {name} gradient not finite"
        num += float((g * rg).sum()); den_a += float((g * g).sum()); den_b += float((rg * rg).sum())
        rel = float((g - rg).norm()) / max(float(rg.norm()), 0.05 * gmax[p.dim()])
        assert rel <= probe_tol, f"{what}: {name} ||g - ref|| / ||ref|| = {rel:.3e}"
    cos = num / math.sqrt(den_a * den_b)
    assert cos >= cos_min, f"{what}: gradient cosine {cos:.6f}"


@pytest.mark.parametrize("B,S", [(64, 80), (320, 36)])       # the step's text stream (5120 rows) and its panorama encoder (64 x 5 panoramas x 36 views = 11520 rows)
def test_bert_layer_block_at_bench_rows_vs_fp64(B, S):
    """SelfAttnBlockFn + FfnBlockFn (BertLayer, vilmodel.py:188-201) forward + backward at the bench's row counts -- the shapes
    whose GEMMs run on gemm_p8_kernel / gemm_q4_kernel / gemm_fast_kernel<64, ...> and whose weight gradients run on
    the grouped 256-square tile -- against an fp64 torch restatement with the same (bf16-rounded) weights."""
    from vln_hamt_amd.model import vilmodel
    from vln_hamt_amd.modeling import HamtConfig
    from vln_hamt_amd import _lib as L
    H = 768
    torch.manual_seed(7)
    layer = vilmodel.BertLayer(HamtConfig(hamt_precision="bf16")).to(DEV).eval()       # eval: dropout off, same kernels
    for n, p in layer.named_parameters():
        with torch.no_grad():
            p.copy_(torch.randn_like(p) * (0.03 if p.dim() == 2 else 0.1) + (1.0 if "LayerNorm.weight" in n else 0.0))
            if p.dim() == 2:
                p.copy_(p.to(torch.bfloat16).float())        # weights exactly representable: the comparison isolates the kernels
    x = torch.randn(B, S, H, device=DEV).requires_grad_()
    lens = torch.randint(S // 2, S + 1, (B,), device=DEV)
    add_mask = ((torch.arange(S, device=DEV)[None] >= lens[:, None]).float() * -10000.0).view(B, 1, 1, S)
    dy = torch.randn(B, S, H, device=DEV)
    (y,) = layer(x, add_mask)
    (y * dy).sum().backward()
    torch.cuda.synchronize()

    P = {n: p.detach().double().requires_grad_() for n, p in layer.named_parameters()}
    xd = x.detach().double().requires_grad_()
    lin = lambda t, n: t @ P[n + ".weight"].t() + P[n + ".bias"]
    q, k, v = (lin(xd, "attention.self." + n) for n in ("query", "key", "value"))
    a = _ref_attn(q, k, v, add_mask.double(), 12)
    h1 = _ref_ln(lin(a, "attention.output.dense") + xd, P["attention.output.LayerNorm.weight"], P["attention.output.LayerNorm.bias"], 1e-12)
    f = F.gelu(lin(h1, "intermediate.dense"))
    yr = _ref_ln(lin(f, "output.dense") + h1, P["output.LayerNorm.weight"], P["output.LayerNorm.bias"], 1e-12)
    (yr * dy.double()).sum().backward()
    close(y, yr, 1e-2, f"BertLayer forward B={B} S={S}")            # north_star: <= 1e-2 for bf16 activations
    assert bool(torch.isfinite(x.grad).all())
    rel = float((x.grad.double() - xd.grad).norm() / xd.grad.norm())
    assert rel <= 2e-2, f"dx relative error {rel:.3e}"
    named = list(layer.named_parameters())
    _grad_report(named, [P[n].grad for n, _ in named], f"BertLayer B={B} S={S}", 0.9995, 3e-2)


def test_cross_attention_block_at_bench_rows_vs_fp64():
    """CrossAttnBlockFn + KvProjFn (BertXAttention, vilmodel.py:351-360) in both directions with the SHARED weights, at the
    x-layers' own shapes (80 text tokens x 43 history + observation tokens, B = 64), forward + backward against fp64."""
    from vln_hamt_amd.model import vilmodel
    from vln_hamt_amd.modeling import HamtConfig
    B, Sl, Sv, H = 64, 80, 43, 768
    torch.manual_seed(11)
    xa = vilmodel.BertXAttention(HamtConfig(hamt_precision="bf16")).to(DEV).eval()
    for n, p in xa.named_parameters():
        with torch.no_grad():
            p.copy_(torch.randn_like(p) * (0.03 if p.dim() == 2 else 0.1) + (1.0 if "LayerNorm.weight" in n else 0.0))
            if p.dim() == 2:
                p.copy_(p.to(torch.bfloat16).float())
    xl = torch.randn(B, Sl, H, device=DEV).requires_grad_()
    xv = torch.randn(B, Sv, H, device=DEV).requires_grad_()
    ll = torch.randint(Sl // 2, Sl + 1, (B,), device=DEV)
    lv = torch.randint(7, Sv + 1, (B,), device=DEV)
    ml = ((torch.arange(Sl, device=DEV)[None] >= ll[:, None]).float() * -10000.0).view(B, 1, 1, Sl)
    mv = ((torch.arange(Sv, device=DEV)[None] >= lv[:, None]).float() * -10000.0).view(B, 1, 1, Sv)
    dl, dv = torch.randn(B, Sl, H, device=DEV), torch.randn(B, Sv, H, device=DEV)
    yl = xa(xl, xv, ctx_att_mask=mv)
    yv = xa(xv, xl, ctx_att_mask=ml)
    ((yl * dl).sum() + (yv * dv).sum()).backward()
    torch.cuda.synchronize()

    P = {n: p.detach().double().requires_grad_() for n, p in xa.named_parameters()}
    xld, xvd = xl.detach().double().requires_grad_(), xv.detach().double().requires_grad_()
    lin = lambda t, n: t @ P[n + ".weight"].t() + P[n + ".bias"]

    def ref(x, c, m):
        a = _ref_attn(lin(x, "att.query"), lin(c, "att.key"), lin(c, "att.value"), m.double(), 12)
        return _ref_ln(lin(a, "output.dense") + x, P["output.LayerNorm.weight"], P["output.LayerNorm.bias"], 1e-12)

    rl, rv = ref(xld, xvd, mv), ref(xvd, xld, ml)
    ((rl * dl.double()).sum() + (rv * dv.double()).sum()).backward()
    close(yl, rl, 1e-2, "cross attention lang <- vis")
    close(yv, rv, 1e-2, "cross attention vis <- lang")
    for got, want, nm in ((xl.grad, xld.grad, "dlang"), (xv.grad, xvd.grad, "dvis")):
        rel = float((got.double() - want).norm() / want.norm())
        assert rel <= 2e-2, f"{nm} relative error {rel:.3e}"
    named = list(xa.named_parameters())
    _grad_report(named, [P[n].grad for n, _ in named], "BertXAttention both directions", 0.9995, 3e-2)


def test_gemm_tr_read_matches_scalar_fallback():
    """ds_read_b64_tr_b16 fragment path == scalar LDS gather path (HAMT_NO_TR=1), bit for bit."""
    code = r"""
import sys, torch
sys.path.insert(0, %r)
from vln_hamt_amd import ops
g = torch.Generator().manual_seed(0)
a = torch.randn(300, 200, generator=g).cuda(); b = torch.randn(300, 170, generator=g).cuda()
o1 = torch.empty(200, 170, device='cuda'); ops.gemm(a, b, o1, a_kmajor=True, b_kmajor=True, prec='bf16')
x = torch.randn(120, 300, generator=g).cuda(); o2 = torch.empty(120, 170, device='cuda'); ops.gemm(x, b, o2, b_kmajor=True, prec='bf16')
torch.save((o1.cpu(), o2.cpu()), sys.argv[1])
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))),)
    outs = []
    for env_extra, name in (({}, "/tmp/hamt_tr.pt"), ({"HAMT_NO_TR": "1"}, "/tmp/hamt_notr.pt")):
        env = dict(os.environ, **env_extra)
        r = subprocess.run([sys.executable, "-c", code, name], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(torch.load(name))
    for x, y in zip(*outs):
        assert torch.equal(x, y)


@pytest.mark.parametrize("tile", ["auto", "256", "128", "64"])
def test_wgrad_grouped_matches_per_problem_reference(tile, monkeypatch):
    """hamt_wgrad_grouped: heterogeneous problems in one call (ragged M/N, column-slice operands, short and long K,
    store and accumulate, fused bias sums) against fp64 matmuls of the same bf16 operands; every tile variant
    (256x256 / 8 waves, 128x128 and 64x128 / 4 waves) and the launcher's own choice."""
    import ctypes as C
    if tile != "auto":
        monkeypatch.setenv("HAMT_WGRAD_TILE", tile)
    from vln_hamt_amd import _lib as L
    ops = _ops()
    lib = L.load()
    specs = [  # (K rows, M out, N in, accum_dw, with_db, accum_db)
        (5120, 768, 768, 0, True, 0), (5120, 768, 3072, 0, False, 0), (5120, 3072, 768, 1, True, 1), (384, 768, 768, 0, True, 0),
        (2368 + 64 - 2368 % 64, 104, 136, 0, True, 0), (64, 8, 128, 1, False, 0), (11520, 768, 768, 0, True, 1), (640, 2304, 768, 0, True, 0),
        (384, 256 + 58, 768, 0, True, 0), (384, 4 * 256 + 186, 256, 1, True, 0),      # ragged last 256-row tile WITH a bias gradient
    ]
    specs = specs * 6      # > one kernarg table (60 entries) => several table-write launches
    keep, refs = [], []
    descs = (L.WgradDesc * len(specs))()
    for i, (K, M, N, aw, wdb, ab) in enumerate(specs):
        valid = K - (i % 3) * 7                    # rows >= valid are zero padding
        dyf = rnd(K, (max(M + 8, 256) + 8 + 7) // 8 * 8, seed=3 * i, scale=0.5)
        dyf[valid:] = 0
        # what lies behind an operand's last column (row padding of the buffer it is a slice of, the next tensor) must not
        # matter: NaN there.  (The 256-square tile's folded bias sums once let a NaN of an out-of-range row into the sums of
        # valid rows: 0 x NaN; the MLM decoder bias, 30 522 = 119 x 256 + 58 rows, caught it in training.)
        dyf[:, 8 * (i % 2) + M:] = float("nan")
        xf = rnd(K, (max(N, 256) + 8 + 7) // 8 * 8, seed=3 * i + 1)
        xf[:, N:] = float("nan")
        kv = 0
        if i % 2 == 1 and valid < K:          # K_valid: the padding rows of BOTH operands may hold anything -- NaN here
            kv = valid
            dyf[valid:] = float("nan")
            xf[valid:] = float("nan")
        dy = dyf.to(torch.bfloat16).to(DEV)[:, 8 * (i % 2):8 * (i % 2) + M]   # column slice of a wider buffer
        x = xf.to(torch.bfloat16).to(DEV)[:, :N]
        dw0 = rnd(M, N, seed=3 * i + 2)
        db0 = rnd(M, seed=3 * i + 5)
        dw, db = dw0.clone().to(DEV), db0.clone().to(DEV)
        ss = torch.zeros(((M + 63) // 64) * ((N + 127) // 128), device=DEV) if i % 4 != 3 else None     # per-tile sums of squares (optional)
        keep += [dy, x, dw, db, ss]
        d = descs[i]
        d.dy, d.x, d.dw, d.db = dy.data_ptr(), x.data_ptr(), dw.data_ptr(), (db.data_ptr() if wdb else None)
        d.M, d.N, d.K, d.ldy, d.ldx, d.ldw, d.accum_dw, d.accum_db = M, N, K, dy.stride(0), x.stride(0), N, aw, ab
        d.ss = ss.data_ptr() if ss is not None else None
        d.K_valid = kv
        kr = kv if kv else K
        # (the fp64 reference products on the GPU -- torch / rocBLAS dgemm as the CHECKER: 60 of them on the host cores took 20 s of this test)
        rw = (dy[:kr].double().t() @ x[:kr].double()).cpu() + (dw0.double() if aw else 0)
        rb = (dy[:kr].double().cpu().sum(0) + (db0.double() if ab else 0)) if wdb else db0.double()
        if not aw and i % 5 == 4:      # wire output (hamt_wgrad_desc.wire_scale): bf16(0.5 dW) into a bf16 array, no fp32 store, no ss
            dw = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
            keep.append(dw)
            d.dw, d.wire_scale, ss = dw.data_ptr(), 0.5, None
            rw = 0.5 * rw
        refs.append((dw, db, rw, rb, ss))
    tab = torch.empty(sum((sp[1] + 63) // 64 for sp in specs) * L.WGRAD_TABLE_ENTRY, dtype=torch.uint8, device=DEV)
    L.check(lib.hamt_debug_fill_lds(0x7FC07FC0, ops._stream()), "hamt_debug_fill_lds")     # every CU's LDS = bf16 NaNs: a tile read before its DMA landed shows
    L.check(lib.hamt_wgrad_grouped(len(specs), descs, tab.data_ptr(), tab.numel(), ops._stream()), "hamt_wgrad_grouped")
    torch.cuda.synchronize()
    n_wire = 0
    for i, (dw, db, rw, rb, ss) in enumerate(refs):
        if dw.dtype == torch.bfloat16:
            n_wire += 1
            assert bool(torch.isfinite(dw).all()), (i, "unwritten wire elements")
            close(dw.float(), rw, 2 ** -8, f"wire dW[{i}] {specs[i]}")
            close(db, rb, 3e-5, f"db[{i}] {specs[i]}")
            continue
        close(dw, rw, 3e-5, f"dW[{i}] {specs[i]}")
        close(db, rb, 3e-5, f"db[{i}] {specs[i]}")
        if ss is not None:       # sum over the tiles' slots = ||dW||^2 of the FINAL values (after accumulation), whatever the tile size
            want = float((rw ** 2).sum())
            got = float(ss.double().sum())
            assert abs(got - want) <= 2e-5 * want and bool(torch.isfinite(ss).all()), (i, specs[i], got, want)
            assert int((ss != 0).sum()) <= ss.numel()
    assert n_wire >= 4
    # argument validation is loud
    descs[0].K = 100
    with pytest.raises(L.HamtError):
        L.check(lib.hamt_wgrad_grouped(1, descs, tab.data_ptr(), tab.numel(), ops._stream()), "hamt_wgrad_grouped")


@pytest.mark.parametrize("tile", ["auto", "256", "128", "64"])
def test_wgrad_grouped_second_operand_pair(tile, monkeypatch):
    """hamt_wgrad_desc.dy2 / x2: dW (+)= dy^T x + dy2^T x2 and db (+)= colsum dy + colsum dy2 in ONE problem (a parameter used twice in a
    pass: the cross-attention weights LXRTXLayer shares between its two directions), next to ordinary problems, on every tile class:
    reductions of different lengths and row strides, a ragged (K2_valid, NaN padding) tail of the second pair, store and accumulate,
    the tile sums of squares of the FINAL values, NaN behind every operand's last column."""
    if tile != "auto":
        monkeypatch.setenv("HAMT_WGRAD_TILE", tile)
    from vln_hamt_amd import _lib as L
    ops = _ops()
    lib = L.load()
    specs = [  # (K, K2 (0: none), K2_valid (0: all), M out, N in, accum_dw, with_db)
        (5120, 2752, 0, 768, 768, 0, True), (2752, 5120, 0, 1536, 768, 0, True), (5120, 0, 0, 768, 3072, 0, True), (384, 5120, 5100, 768, 768, 1, True),
        (5120, 384, 0, 768, 768, 0, False), (640, 640, 601, 256 + 58, 768, 0, True), (1280, 0, 0, 768, 768, 1, True), (25600, 13760, 0, 768, 768, 0, True),
    ] * 5
    keep, refs = [], []
    descs = (L.WgradDesc * len(specs))()

    def operand(K, C, seed, valid=0):
        f = rnd(K, (max(C, 256) + 16 + 7) // 8 * 8, seed=seed, scale=0.5)
        f[:, C:] = float("nan")
        if valid:
            f[valid:] = float("nan")
        t = f.to(torch.bfloat16).to(DEV)
        keep.append(t)
        return t[:, :C]

    for i, (K, K2, kv2, M, N, aw, wdb) in enumerate(specs):
        dy, x = operand(K, M, 7 * i), operand(K, N, 7 * i + 1)
        dw0, db0 = rnd(M, N, seed=7 * i + 2), rnd(M, seed=7 * i + 3)
        dw, db = dw0.clone().to(DEV), db0.clone().to(DEV)
        ss = torch.zeros(((M + 63) // 64) * ((N + 127) // 128), device=DEV)
        keep += [dw, db, ss]
        d = descs[i]
        d.dy, d.x, d.dw, d.db = dy.data_ptr(), x.data_ptr(), dw.data_ptr(), (db.data_ptr() if wdb else None)
        d.M, d.N, d.K, d.ldy, d.ldx, d.ldw, d.accum_dw, d.accum_db = M, N, K, dy.stride(0), x.stride(0), N, aw, aw
        d.ss = ss.data_ptr()
        rw = (dy.double().t() @ x.double()).cpu() + (dw0.double() if aw else 0)
        rb = dy.double().cpu().sum(0) + (db0.double() if aw else 0)
        if K2:
            dy2, x2 = operand(K2, M, 7 * i + 4, kv2), operand(K2, N, 7 * i + 5, kv2)
            d.dy2, d.x2, d.K2, d.ldy2, d.ldx2, d.K2_valid = dy2.data_ptr(), x2.data_ptr(), K2, dy2.stride(0), x2.stride(0), kv2
            kr = kv2 if kv2 else K2
            rw = rw + (dy2[:kr].double().t() @ x2[:kr].double()).cpu()
            rb = rb + dy2[:kr].double().cpu().sum(0)
        refs.append((dw, db, rw, rb if wdb else db0.double(), ss))
    tab = torch.empty(sum((sp[3] + 63) // 64 for sp in specs) * L.WGRAD_TABLE_ENTRY, dtype=torch.uint8, device=DEV)
    L.check(lib.hamt_debug_fill_lds(0x7FC07FC0, ops._stream()), "hamt_debug_fill_lds")
    L.check(lib.hamt_wgrad_grouped(len(specs), descs, tab.data_ptr(), tab.numel(), ops._stream()), "hamt_wgrad_grouped")
    torch.cuda.synchronize()
    for i, (dw, db, rw, rb, ss) in enumerate(refs):
        close(dw, rw, 3e-5, f"dW[{i}] {specs[i]}")
        close(db, rb, 3e-5, f"db[{i}] {specs[i]}")
        want, got = float((rw ** 2).sum()), float(ss.double().sum())
        assert abs(got - want) <= 2e-5 * want and bool(torch.isfinite(ss).all()), (i, specs[i], got, want)
    descs[0].K_valid = 5000            # a ragged tail between the two reductions is refused, loudly
    with pytest.raises(L.HamtError):
        L.check(lib.hamt_wgrad_grouped(1, descs, tab.data_ptr(), tab.numel(), ops._stream()), "hamt_wgrad_grouped")


def test_parameter_used_twice_is_one_weight_gradient_problem(monkeypatch):
    """A linear layer applied to two inputs in one pass (what LXRTXLayer does with its shared cross-attention, vilmodel.py:401-412): the
    queued weight gradient is ONE problem with two operand pairs -- one launch -- and equals the two-launch result and fp64."""
    from vln_hamt_amd import wgrad
    from vln_hamt_amd.optim import AdamW
    ops = _ops()
    if not wgrad.ENABLED:
        pytest.skip("HAMT_NO_DEFER_WGRAD")
    torch.manual_seed(0)
    lin = torch.nn.Linear(768, 768).to(DEV)
    xa, xb = rnd(5120, 768, seed=1).to(DEV), rnd(2752, 768, seed=2).to(DEV)
    ga, gb = rnd(5120, 768, seed=3).to(DEV), rnd(2752, 768, seed=4).to(DEV)
    res = {}
    for merged in (True, False):
        monkeypatch.setattr(wgrad, "MERGE_PAIRS", merged)
        o = AdamW([{"params": list(lin.parameters()), "weight_decay": 0.0}], lr=1e-3)
        o.materialize()
        o.zero_grad()
        n0 = wgrad.stats["problems"]
        with _CountLaunches("hamt_wgrad_grouped_ex") as cnt:
            ya, yb = ops.linear(xa, lin.weight, lin.bias, ops.ACT_NONE, "bf16"), ops.linear(xb, lin.weight, lin.bias, ops.ACT_NONE, "bf16")
            torch.autograd.backward([ya, yb], [ga, gb])
        torch.cuda.synchronize()
        res[merged] = (lin.weight.grad.detach().clone(), lin.bias.grad.detach().clone(), cnt.n["hamt_wgrad_grouped_ex"])
        o.zero_grad()
    assert res[True][2] == 1 and res[False][2] == 2, (res[True][2], res[False][2])
    rw = bf16_round(ga.cpu()).double().t() @ bf16_round(xa.cpu()).double() + bf16_round(gb.cpu()).double().t() @ bf16_round(xb.cpu()).double()
    rb = bf16_round(ga.cpu()).double().sum(0) + bf16_round(gb.cpu()).double().sum(0)
    for merged in (True, False):
        close(res[merged][0], rw, 3e-5, f"dW merged={merged}")
        close(res[merged][1], rb, 3e-5, f"db merged={merged}")


class _CountLaunches:
    """counts calls of C entry points (patches the ctypes handles for the duration)"""

    def __init__(self, *names):
        self.names, self.n = names, {k: 0 for k in names}

    def __enter__(self):
        from vln_hamt_amd import _lib as L
        self.lib, self.orig = L.load(), {}
        for k in self.names:
            f = getattr(self.lib, k)
            self.orig[k] = f

            def wrap(*a, _f=f, _k=k):
                self.n[_k] += 1
                return _f(*a)
            setattr(self.lib, k, wrap)
        return self

    def __exit__(self, *a):
        for k, f in self.orig.items():
            setattr(self.lib, k, f)


def test_wgrad_kernel_timing_aid():
    """hamt_debug_wgrad_timing / _times (bench.py's `roofline`): one bracket per grouped KERNEL launch, positive durations, and
    the library's own work count = sum 2 M N K over the k-tiles it multiplies (whole 64-row tiles up to K_valid)."""
    import ctypes as C
    from vln_hamt_amd import _lib as L
    ops = _ops()
    lib = L.load()
    specs = [(1024, 768, 768, 0), (1024, 768, 3072, 0), (512, 3072, 768, 200), (256, 768, 768, 0)]      # (K, M, N, K_valid)
    descs = (L.WgradDesc * len(specs))()
    keep, want = [], 0.0
    for i, (K, M, N, kv) in enumerate(specs):
        dy = rnd(K, M, seed=i).to(torch.bfloat16).to(DEV)
        x = rnd(K, N, seed=10 + i).to(torch.bfloat16).to(DEV)
        dw = torch.empty(M, N, device=DEV)
        keep += [dy, x, dw]
        d = descs[i]
        d.dy, d.x, d.dw, d.db = dy.data_ptr(), x.data_ptr(), dw.data_ptr(), None
        d.M, d.N, d.K, d.ldy, d.ldx, d.ldw, d.accum_dw, d.accum_db, d.K_valid = M, N, K, M, N, N, 0, 0, kv
        want += 2.0 * M * N * (((kv + 63) // 64 * 64) if kv else K)
    tab = torch.empty(sum((sp[1] + 63) // 64 for sp in specs) * L.WGRAD_TABLE_ENTRY, dtype=torch.uint8, device=DEV)
    assert lib.hamt_debug_wgrad_times(None, None, None, 0) == 0
    L.check(lib.hamt_debug_wgrad_timing(1), "hamt_debug_wgrad_timing")
    for _ in range(2):
        L.check(lib.hamt_wgrad_grouped(len(specs), descs, tab.data_ptr(), tab.numel(), ops._stream()), "hamt_wgrad_grouped")
    us, rows, fl = (C.c_float * 16)(), (C.c_int * 16)(), (C.c_double * 16)()
    n = lib.hamt_debug_wgrad_times(us, rows, fl, 16)
    assert n >= 2 and n % 2 == 0, n
    assert all(0.0 < us[i] < 1e5 and rows[i] in (64, 128, 256) for i in range(n)), [(us[i], rows[i]) for i in range(n)]
    assert abs(sum(fl[i] for i in range(n)) - 2 * want) <= 1e-9 * want, (sum(fl[i] for i in range(n)), 2 * want)
    L.check(lib.hamt_debug_wgrad_timing(0), "hamt_debug_wgrad_timing")
    L.check(lib.hamt_wgrad_grouped(len(specs), descs, tab.data_ptr(), tab.numel(), ops._stream()), "hamt_wgrad_grouped")
    assert lib.hamt_debug_wgrad_times(us, rows, fl, 16) == 0          # off: nothing recorded
    torch.cuda.synchronize()


def test_deferred_wgrad_queue_semantics():
    """wgrad.py: queued gradients are published as .grad at the end of backward; a parameter used twice in one pass and
    gradient accumulation over two passes sum like autograd's AccumulateGrad; results equal the immediate path."""
    from vln_hamt_amd import wgrad
    ops = _ops()
    torch.manual_seed(0)
    lin = torch.nn.Linear(256, 192).to(DEV)
    x1 = rnd(130, 256, seed=1).to(DEV).requires_grad_()
    x2 = rnd(70, 256, seed=2).to(DEV)

    def run(enabled):
        wgrad.ENABLED = enabled
        try:
            lin.weight.grad = lin.bias.grad = None
            x1.grad = None
            y = ops.linear(x1, lin.weight, lin.bias, prec="bf16").square().sum() + ops.linear(x2, lin.weight, lin.bias, prec="bf16").sum()
            y.backward()
            assert wgrad.pending() == 0
            g1 = (lin.weight.grad.clone(), lin.bias.grad.clone(), x1.grad.clone())
            (ops.linear(x2, lin.weight, lin.bias, prec="bf16").sum() * 2).backward()      # accumulate over a second pass
            return g1, (lin.weight.grad.clone(), lin.bias.grad.clone())
        finally:
            wgrad.ENABLED = True

    n0 = wgrad.stats["problems"]
    (w_d, b_d, dx_d), (w_d2, b_d2) = run(True)
    assert wgrad.stats["problems"] - n0 == 3
    (w_i, b_i, dx_i), (w_i2, b_i2) = run(False)
    close(w_d, w_i, 2e-5, "dW deferred vs immediate")
    close(b_d, b_i, 2e-3, "db deferred vs immediate")   # queued db sums the bf16 image of dY, the immediate path fp32 dY
    close(dx_d, dx_i, 0.0, "dx")
    close(w_d2, w_i2, 2e-5, "dW after a second pass")
    close(b_d2, b_i2, 2e-3, "db after a second pass")


def test_cast_pad_and_transpose_exact():
    ops = _ops()
    x = rnd(77, 100, seed=1)
    y = ops.cast_pad16(x.to(DEV))
    assert y.shape == (128, 128) and torch.equal(y[:77, :100].float().cpu(), bf16_round(x))
    assert float(y[:, 100:].float().abs().max()) == 0 and float(y[77:].float().abs().max()) == 0
    buf = torch.zeros(77, 104, device=DEV)
    buf[:, 1:101] = x.to(DEV)                                   # odd column offset: unaligned rows take the scalar path
    assert torch.equal(ops.cast_pad16(buf[:, 1:101]).cpu(), y.cpu())
    t = ops.cast_t16(x.to(DEV))
    assert t.shape == (100, 128) and torch.equal(t[:, :77].float().cpu(), bf16_round(x).t()) and float(t[:, 77:].float().abs().max()) == 0
    t2 = ops.cast_t16(x.to(DEV).to(torch.bfloat16))
    assert torch.equal(t2.cpu(), t.cpu())


def test_colsum():
    ops = _ops()
    for (M, N) in [(1000, 130), (5000, 768), (3, 40)]:
        x = rnd(M, N, seed=M)
        close(ops.colsum(x.to(DEV)), x.double().sum(0), 1e-5, f"colsum {M}x{N}")


# ------------------------------------------------------------------------------------------ attention
def _attn_ref(q, k, v, mask, heads):
    B, Sq, H = q.shape
    Sk = k.shape[1]
    d = H // heads
    qh = q.view(B, Sq, heads, d).transpose(1, 2)
    kh = k.view(B, Sk, heads, d).transpose(1, 2)
    vh = v.view(B, Sk, heads, d).transpose(1, 2)
    s = qh @ kh.transpose(-1, -2) / math.sqrt(d)
    if mask is not None:
        s = s + mask
    return (torch.softmax(s, -1) @ vh).transpose(1, 2).reshape(B, Sq, H)


@pytest.mark.parametrize("B,heads,Sq,Sk,packed,use_mask", [
    (2, 2, 80, 80, True, True), (3, 2, 36, 36, True, False), (2, 3, 43, 80, False, True),
    (2, 2, 80, 43, False, True), (1, 2, 130, 150, False, True), (2, 1, 6, 20, False, True), (2, 2, 250, 250, True, True),
    (2, 2, 128, 128, True, True), (2, 2, 80, 6, False, True), (3, 1, 1, 37, False, False), (2, 2, 100, 17, False, True), (2, 1, 128, 130, False, True),
    (2, 2, 197, 197, True, False), (1, 2, 300, 256, False, True), (2, 1, 200, 100, False, True), (1, 1, 140, 257, False, True)])
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_attention_fwd_bwd(B, heads, Sq, Sk, packed, use_mask, prec):
    """fp32: exact-MFMA kernels (attn.hip) <= 5e-5; bf16: bf16-MFMA kernels (attn16.hip), operands and P rounded to bf16
    (fp32 softmax statistics) => <= 2e-2 of the output scale."""
    ops = _ops()
    t_out, t_grad = (2e-5, 5e-5) if prec == "fp32" else (1.5e-2, 2.5e-2)
    H = heads * 64
    q, k, v = (rnd(B, Sq, H, seed=1).double(), rnd(B, Sk, H, seed=2).double(), rnd(B, Sk, H, seed=3).double())
    mask = None
    if use_mask:
        keep = torch.rand(B, Sk, generator=torch.Generator().manual_seed(4)) > 0.3
        keep[:, 0] = True
        mask = ((1.0 - keep.double()) * -10000.0)[:, None, None, :]
    q.requires_grad_(True), k.requires_grad_(True), v.requires_grad_(True)
    ref = _attn_ref(q, k, v, mask, heads)
    go = rnd(B, Sq, H, seed=5).double()
    ref.backward(go)
    if packed:
        src = torch.cat([q, k, v], -1).detach().float().reshape(B * Sq, 3 * H).to(DEV).requires_grad_(True)
        out = ops.attention(src, None, mask.float().to(DEV) if mask is not None else None, B, heads, 0.0, prec)
        out.backward(go.float().reshape(B * Sq, H).to(DEV))
        g = src.grad.view(B, Sq, 3 * H)
        gq, gk, gv = g[..., :H], g[..., H:2 * H], g[..., 2 * H:]
    else:
        qs = q.detach().float().reshape(B * Sq, H).to(DEV).requires_grad_(True)
        kvs = torch.cat([k, v], -1).detach().float().reshape(B * Sk, 2 * H).to(DEV).requires_grad_(True)
        out = ops.attention(qs, kvs, mask.float().to(DEV) if mask is not None else None, B, heads, 0.0, prec)
        out.backward(go.float().reshape(B * Sq, H).to(DEV))
        gq = qs.grad.view(B, Sq, H)
        gkv = kvs.grad.view(B, Sk, 2 * H)
        gk, gv = gkv[..., :H], gkv[..., H:]
    close(out.view(B, Sq, H), ref, t_out, "attn out")
    close(gq, q.grad, t_grad, "attn dq")
    close(gk, k.grad, t_grad, "attn dk")
    close(gv, v.grad, t_grad, "attn dv")


@pytest.mark.parametrize("S", [80, 197, 300])
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_attention_dropout_properties(prec, S):
    """Counter-based dropout: same mask in fwd and bwd (adjoint identity in V), keep-rate, 1/(1-p) scaling.  S = 197 pairs
    the single-pass forward with the tiled backward (bf16 path), S = 300 is tiled both ways: one mask stream for all."""
    ops = _ops()
    B, heads, H, p = 4, 2, 128, 0.3
    qkv = rnd(B * S, 3 * H, seed=9).to(DEV)
    ops.manual_seed(1234)
    torch.manual_seed(0)

    def run(x):
        from vln_hamt_amd import ops as o
        o._call_counter[0] = 77                      # same call id => same mask
        return o.attention(x, None, None, B, heads, p, prec)
    x1 = qkv.clone().requires_grad_(True)
    o1 = run(x1)
    o1b = run(qkv.clone())
    assert torch.equal(o1, o1b)                      # deterministic for a fixed (seed, epoch, call id)
    x0 = qkv.clone()
    from vln_hamt_amd import ops as o
    o._call_counter[0] = 77
    o0 = o.attention(x0, None, None, B, heads, 0.0, prec)
    # E[dropout(P)] = P: averaged over many elements the outputs agree
    assert abs(float((o1 - o0).mean())) < 5e-3
    assert float((o1 - o0).abs().max()) > 1e-3       # but masks were applied
    # adjoint identity in V: <dO, O(V)> == <dV, V>   (O is linear in V for a fixed mask)
    go = rnd(B * S, H, seed=10).to(DEV)
    o1.backward(go)
    dv = x1.grad[:, 2 * H:]
    lhs = float((go.double() * o1.double()).sum())
    rhs = float((dv.double() * qkv[:, 2 * H:].double()).sum())
    scale = float((go.double().abs() * o1.double().abs()).sum())        # the inner product itself nearly cancels
    assert abs(lhs - rhs) <= (1e-6 if prec == "fp32" else 2e-3) * scale, (lhs, rhs, scale)


# ------------------------------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize("M,H,res", [(37, 768, True), (130, 128, False), (5, 1024, True), (1000, 768, True), (5120, 768, True), (4101, 768, False), (4608, 256, True)])
def test_ln_fwd_bwd(M, H, res):
    ops = _ops()
    x, r = rnd(M, H, seed=1), rnd(M, H, seed=2) if res else None
    ln = torch.nn.LayerNorm(H, eps=1e-12)
    with torch.no_grad():
        ln.weight.copy_(1 + 0.1 * rnd(H, seed=3)), ln.bias.copy_(0.1 * rnd(H, seed=4))
    xr = x.double().requires_grad_(True)
    rr = r.double().requires_grad_(True) if res else None
    lnd = torch.nn.LayerNorm(H, eps=1e-12).double()
    lnd.load_state_dict(ln.state_dict())
    ref = lnd(xr + rr if res else xr)
    go = rnd(M, H, seed=5)
    ref.backward(go.double())
    lng = torch.nn.LayerNorm(H, eps=1e-12).to(DEV)
    lng.load_state_dict(ln.state_dict())
    xg = x.to(DEV).requires_grad_(True)
    rg = r.to(DEV).requires_grad_(True) if res else None
    y = ops.layer_norm(xg, rg, lng)
    y.backward(go.to(DEV))
    close(y, ref, 2e-5, "ln y")
    close(xg.grad, xr.grad, 5e-5, "ln dx")
    if res:
        close(rg.grad, rr.grad, 5e-5, "ln dres")
    close(lng.weight.grad, lnd.weight.grad, 5e-5, "ln dgamma")
    close(lng.bias.grad, lnd.bias.grad, 5e-5, "ln dbeta")


@pytest.mark.parametrize("M,H", [(130, 768), (64, 128), (5, 512)])
def test_ln_bf16_dense_input_and_bf16_saved_sum(M, H):
    """hamt_ln_desc.io16: a bf16 dense output as `x` gives exactly what the same values give in fp32 (the kernel widens them);
    the saved pre-LN sum kept in bf16 changes the backward by its rounding only (it is re-normalised with the exact fp32 mean /
    rstd): bounded against the fp32-saved backward."""
    ops = _ops()
    x16 = rnd(M, H, seed=1).to(DEV).to(torch.bfloat16)
    r = rnd(M, H, seed=2).to(DEV)
    g, b = (1.0 + 0.1 * rnd(H, seed=3)).to(DEV), (0.1 * rnd(H, seed=4)).to(DEV)
    ya, ya16, za, mean_a, rstd_a, _ = ops._ln_fwd(x16, r, g, b, 1e-12, 0.0, 0.0, True)
    yb, yb16, zb, mean_b, rstd_b, _ = ops._ln_fwd(x16.float(), r, g, b, 1e-12, 0.0, 0.0, True)
    assert za.dtype == torch.bfloat16 and zb.dtype == torch.float32
    assert torch.equal(ya, yb) and torch.equal(ya16[:M], yb16[:M]) and torch.equal(mean_a, mean_b) and torch.equal(rstd_a, rstd_b)
    assert torch.equal(za, zb.to(torch.bfloat16))
    dy = rnd(M, H, seed=5).to(DEV)
    dza, _, dxa16, dga, dba, _ = ops._ln_bwd(dy, za, mean_a, rstd_a, g, 1e-12, 0.0, 0.0, 0, False, True, False)
    dzb, _, dxb16, dgb, dbb, _ = ops._ln_bwd(dy, zb, mean_b, rstd_b, g, 1e-12, 0.0, 0.0, 0, False, True, False)
    assert torch.equal(dba, dbb)                                       # dbeta does not involve z
    scale = float(dzb.abs().max())
    assert float((dza - dzb).abs().max()) < 2e-2 * scale               # |z| rstd 2^-9 ~ 1e-2 of a unit-variance x_hat at the tails
    assert float((dga - dgb).abs().max()) < 2e-2 * float(dgb.abs().max())
    cos = float((dza * dzb).sum() / (dza.norm() * dzb.norm()))
    assert cos > 0.9999, cos


@pytest.mark.parametrize("M,N,K", [(5120, 768, 768), (2752, 768, 3072), (130, 96, 128), (33, 136, 64)])
def test_half_dense_output_interface(M, N, K):
    """HAMT_F16 (round 6): the dense layer in front of a LayerNorm stores IEEE half -- `C` of a plain / bias / accumulate hamt_gemm
    (interior tiles through epi_fast8, ragged ones through epi_store), read back by hamt_ln_fwd as `x` and saved as `z` in the same
    format.  Half's rounding (2^-11) instead of bf16's (2^-8); values beyond +-65504 saturate (never inf); any other epilogue is refused."""
    ops = _ops()
    from vln_hamt_amd import _lib as L
    a, w, bias = rnd(M, K, seed=1), rnd(N, K, seed=2), rnd(N, seed=3)
    a16, w16, b = a.to(DEV).to(torch.bfloat16), w.to(DEV).to(torch.bfloat16), bias.to(DEV)
    ref = bf16_round(a).double() @ bf16_round(w).double().t() + bias.double()
    oh = torch.empty(M, N, dtype=torch.float16, device=DEV)
    ob = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    ops.gemm(a16, w16, oh, bias=b)
    ops.gemm(a16, w16, ob, bias=b)
    scale = float(ref.abs().max())
    eh, eb = float((oh.double().cpu() - ref).abs().max()) / scale, float((ob.double().cpu() - ref).abs().max()) / scale
    assert eh <= 2.0 ** -11 * 1.05 + 3e-6 * math.sqrt(K) and eb > 4 * eh, (eh, eb)        # one rounding of the fp32 accumulator, 8 x finer than bf16's
    base = rnd(M, N, seed=4)
    acc = base.to(DEV).to(torch.float16)
    ops.gemm(a16, w16, acc, epilogue=L.EPI_ACCUM)
    close(acc.float(), base.to(torch.float16).double() + (ref - bias.double()), 2e-3, "half C +=")
    big = torch.empty(M, N, dtype=torch.float16, device=DEV)
    ops.gemm(a16, w16, big, bias=b, alpha=1e6)
    assert bool(torch.isfinite(big).all()) and float(big.abs().max()) == 65504.0, "half C must saturate, not overflow"
    with pytest.raises(L.HamtError):
        ops.gemm(a16, w16, oh, bias=b, epilogue=L.EPI_GELU)
    # the LayerNorm side: x as half == the same values given in fp32; z saved as half
    if N % 4 == 0:
        r = rnd(M, N, seed=5).to(DEV)
        g, be = (1.0 + 0.1 * rnd(N, seed=6)).to(DEV), (0.1 * rnd(N, seed=7)).to(DEV)
        ya, ya16, za, mean_a, rstd_a, _ = ops._ln_fwd(oh, r, g, be, 1e-12, 0.0, 0.0, True)
        yb, yb16, zb, mean_b, rstd_b, _ = ops._ln_fwd(oh.float(), r, g, be, 1e-12, 0.0, 0.0, True)
        assert za.dtype == torch.float16 and zb.dtype == torch.float32
        assert torch.equal(ya, yb) and torch.equal(ya16[:M], yb16[:M]) and torch.equal(mean_a, mean_b) and torch.equal(rstd_a, rstd_b)
        assert torch.equal(za, zb.to(torch.float16))
        dy = rnd(M, N, seed=8).to(DEV)
        dza, _, _, dga, dba, _ = ops._ln_bwd(dy, za, mean_a, rstd_a, g, 1e-12, 0.0, 0.0, 0, False, True, False)
        dzb, _, _, dgb, dbb, _ = ops._ln_bwd(dy, zb, mean_b, rstd_b, g, 1e-12, 0.0, 0.0, 0, False, True, False)
        assert torch.equal(dba, dbb)
        assert float((dza - dzb).abs().max()) < 3e-3 * float(dzb.abs().max())          # (the bf16-saved sum: 2e-2, test above)
        assert float((dga - dgb).abs().max()) < 3e-3 * float(dgb.abs().max())


def test_ln_dropout_pre_post():
    ops = _ops()
    M, H, p = 512, 768, 0.1
    lng = torch.nn.LayerNorm(H, eps=1e-12).to(DEV)
    x, r = rnd(M, H, seed=1).to(DEV), rnd(M, H, seed=2).to(DEV)
    from vln_hamt_amd import ops as o
    # p_post: zeros at rate p, survivors scaled by 1/(1-p); identical mask in backward
    o._call_counter[0] = 5
    xg = x.clone().requires_grad_(True)
    y = ops.layer_norm(xg, None, lng, p_post=p)
    y0 = ops.layer_norm(x, None, lng)
    dropped = (y == 0) & (y0 != 0)
    rate = float(dropped.float().mean())
    assert abs(rate - p) < 0.01, rate
    close(y[~dropped], y0[~dropped] / (1 - p), 1e-5, "post scale")
    y.backward(torch.ones_like(y))
    # p_pre: LN(dropout(x) + r); adjoint identity of the masked branch: dx == dz * mask/(1-p)
    o._call_counter[0] = 9
    xg2, rg2 = x.clone().requires_grad_(True), r.clone().requires_grad_(True)
    y2 = ops.layer_norm(xg2, rg2, lng, p_pre=p)
    y2.backward(rnd(M, H, seed=3).to(DEV))
    ratio = xg2.grad / rg2.grad
    z = ratio[torch.isfinite(ratio)]
    assert bool(((z.abs() < 1e-6) | ((z - 1 / (1 - p)).abs() < 1e-4)).all()), torch.unique(torch.round(z * 1e4) / 1e4)
    assert abs(float((xg2.grad == 0).float().mean()) - p) < 0.01


# ------------------------------------------------------------------------------------------ gathers & co
@pytest.mark.parametrize("M,K,H", [(320, 768, 768), (2368, 768, 768), (11520, 768, 768), (37, 64, 128), (1000, 512, 1024)])
@pytest.mark.parametrize("prec", ["fp32", "bf16", "bf16-x16"])
def test_vis_embed_fwd_bwd(M, K, H, prec, monkeypatch):
    """hamt_vis_embed_fwd / _bwd (img_layer_norm(img_linear(img)) + ang_layer_norm(ang_linear(ang)), vilmodel.py:498-500) at the
    step's row counts (history steps 320, observation tokens 2368, panorama views 11520 at B = 64) against fp64 autograd.  fp32 mode:
    everything <= 2e-5 of scale.  bf16 mode: the fp64 reference is fed the kernel's own bf16 dense output (so the LayerNorm part is
    checked tightly: 1e-4, the bf16 image to bf16 rounding) and the dense layer's gradients loosely (2e-2: bf16 operands)."""
    ops = _ops()
    nn = torch.nn
    x16 = prec == "bf16-x16"         # the dense layer's output handed over in bf16 (HAMT_VIS_EMBED_X16=1; off by default)
    prec = prec.split("-")[0]
    monkeypatch.setattr(ops, "VIS_EMBED_X16", x16)
    torch.manual_seed(M + H)
    img_lin, ang_lin = nn.Linear(K, H), nn.Linear(4, H)
    ln1, ln2 = nn.LayerNorm(H, eps=1e-12), nn.LayerNorm(H, eps=1e-12)
    for ln in (ln1, ln2):
        ln.weight.data = 1.0 + 0.3 * torch.randn(H)
        ln.bias.data = 0.2 * torch.randn(H)
    mods = nn.ModuleList([img_lin, ang_lin, ln1, ln2])
    img, ang = rnd(M, K, seed=1), rnd(M, 4, seed=2)
    gy = rnd(M, H, seed=3)
    # fp64 reference
    ref = nn.ModuleList([nn.Linear(K, H), nn.Linear(4, H), nn.LayerNorm(H, eps=1e-12), nn.LayerNorm(H, eps=1e-12)]).double()
    ref.load_state_dict({k: v.double() for k, v in mods.state_dict().items()})
    mods = mods.to(DEV)
    x = img.to(DEV).requires_grad_(True)
    assert ops.vis_embed_ok(x, ang.to(DEV), mods[0], mods[1])
    y = ops.vis_embed(x, ang.to(DEV), mods[0], mods[2], mods[1], mods[3], prec, want16=True)
    y16 = ops.shadow16(y)
    assert y16 is not None and y16.shape == ((M + 63) // 64 * 64, H)
    y.backward(gy.to(DEV))
    xr = img.double().requires_grad_(True)
    x1 = ref[0](xr)
    if prec == "bf16":       # the dense layer's output as the kernel saw it: bf16 operands, bf16 result
        with torch.no_grad():
            x1q = bf16_round(img) @ bf16_round(mods[0].weight.detach().cpu()).t() + mods[0].bias.detach().cpu()
            x1q = (x1q.to(torch.bfloat16) if x16 else x1q).double()
        x1 = x1 + (x1q - x1).detach()
    yr = ref[2](x1) + ref[3](ref[1](ang.double()))
    yr.backward(gy.double())
    tight = 2e-5 if prec == "fp32" else 1e-4
    close(y, yr, 2e-3 if x16 else tight, "y")      # (x16: x1q above is torch's rounding of ITS sum, a few ulps of bf16 off the kernel's)
    close(y16[:M], yr, 1e-2, "y16")
    assert float(y16[M:].float().abs().max() if y16.shape[0] > M else 0.0) == 0.0
    loose = 2e-5 if prec == "fp32" else 2e-2
    for name, a, b in (("ln_img.weight", mods[2].weight.grad, ref[2].weight.grad), ("ln_img.bias", mods[2].bias.grad, ref[2].bias.grad),
                       ("ln_ang.weight", mods[3].weight.grad, ref[3].weight.grad), ("ln_ang.bias", mods[3].bias.grad, ref[3].bias.grad),
                       ("ang.weight", mods[1].weight.grad, ref[1].weight.grad), ("ang.bias", mods[1].bias.grad, ref[1].bias.grad)):
        close(a, b, 2e-3 if (x16 and name.startswith("ln_img")) else tight, name)
    close(mods[0].weight.grad, ref[0].weight.grad, loose, "img.weight")
    close(mods[0].bias.grad, ref[0].bias.grad, loose, "img.bias")
    close(x.grad, xr.grad, loose, "d img")


def test_vis_embed_matches_the_four_launch_path(monkeypatch):
    """the fused embedding against the path it replaces (dense, two LayerNorms, K = 4 dense, add) on the same modules, bf16 mode,
    gradients into a real optimizer arena (slots) with the deferred weight gradients flushed"""
    ops = _ops()
    from vln_hamt_amd.model.vilmodel import _VisualLinears
    nn = torch.nn
    torch.manual_seed(0)
    mods = nn.ModuleList([nn.Linear(768, 768), nn.Linear(4, 768), nn.LayerNorm(768, eps=1e-12), nn.LayerNorm(768, eps=1e-12)]).to(DEV)
    img, ang, gy = rnd(640, 36, 768, seed=4).to(DEV), rnd(640, 36, 4, seed=5).to(DEV), rnd(640, 36, 768, seed=6).to(DEV)
    res = {}
    for fused in (True, False):
        monkeypatch.setattr(ops, "VIS_EMBED", fused)
        for p in mods.parameters():
            p.grad = None
        y = _VisualLinears.two_stream(mods[0], mods[2], mods[1], mods[3], img, ang, "bf16")
        y.backward(gy)
        from vln_hamt_amd import wgrad
        wgrad.flush()
        res[fused] = (y.detach().clone(), [p.grad.detach().clone() for p in mods.parameters()])
    close(res[True][0], res[False][0], 1e-2, "y")
    for a, b, (n, _) in zip(res[True][1], res[False][1], mods.named_parameters()):
        close(a, b, 2e-2, n)


def test_embed_sum_and_gather_scatter_exact():
    ops = _ops()
    B, L, H, V = 3, 20, 128, 500
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(0, V, (B, L), generator=g)
    ids[0, :5] = 7                                             # duplicates exercise the atomic scatter
    word, pos, typ = rnd(V, H, seed=1), rnd(64, H, seed=2), rnd(2, H, seed=3)
    w, p_, t = (x.to(DEV).requires_grad_(True) for x in (word, pos, typ))
    z = ops.embed_sum(ids.to(DEV), w, p_, t)
    ref = (word[ids] + pos[:L][None]) + typ[0][None, None]
    assert torch.equal(z.cpu(), ref)                           # same association order => bit exact
    go = rnd(B, L, H, seed=4)
    z.backward(go.to(DEV))
    wr, pr, tr = (x.clone().double().requires_grad_(True) for x in (word, pos, typ))
    ((wr[ids] + pr[:L][None]) + tr[0][None, None]).backward(go.double())
    close(w.grad, wr.grad, 1e-5, "dword")
    close(p_.grad, pr.grad, 1e-5, "dpos")
    close(t.grad, tr.grad, 1e-5, "dtype")
    # gather_rows (+base) / scatter
    tab = rnd(50, H, seed=5)
    idx = torch.randint(0, 50, (77,), generator=g)
    base = rnd(77, H, seed=6)
    tg, bg = tab.to(DEV).requires_grad_(True), base.to(DEV).requires_grad_(True)
    out = ops.gather_rows(tg, idx.to(DEV), base=bg)
    assert torch.equal(out.cpu(), base + tab[idx])
    go = rnd(77, H, seed=7)
    out.backward(go.to(DEV))
    tr_ = tab.clone().double().requires_grad_(True)
    (tr_[idx]).backward(go.double())
    close(tg.grad, tr_.grad, 1e-5, "gather dtab")
    assert torch.equal(bg.grad.cpu(), go)
    # a table of a few rows gathered by many rows (token / navigability types): fixed-order sums, bit-reproducible
    for T, R in ((3, 2368), (1, 64), (8, 5000)):
        tab = rnd(T, H, seed=8)
        idx = torch.randint(0, T, (R,), generator=g)
        go = rnd(R, H, seed=9)
        grads = []
        for rep in range(3):
            tg = tab.to(DEV).requires_grad_(True)
            junk = torch.randn(1000 * (rep + 1), device=DEV)      # (another allocator state per repetition)
            ops.gather_rows(tg, idx.to(DEV)).backward(go.to(DEV))
            grads.append(tg.grad.clone())
            del junk
        tr_ = tab.clone().double().requires_grad_(True)
        (tr_[idx]).backward(go.double())
        close(grads[0], tr_.grad, 1e-5, f"small-table dtab T={T}")
        assert torch.equal(grads[0], grads[1]) and torch.equal(grads[0], grads[2]), "small-table scatter is not bit-reproducible"


@pytest.mark.parametrize("R,T,W", [(5120, 30522, 768), (5120, 40, 768), (2368, 9, 768), (33000, 100, 64), (1, 5, 128), (1025, 1024, 132)])
def test_scatter_add_ordered_is_bit_reproducible(R, T, W):
    """dst[idx[r]] += src[r] with colliding rows summed in ROW order by one writer per table row (hamt_scatter_add_rows_ordered, the word
    embeddings' gradient in hamt_embed_sum_bwd): equal to a sequential fp32 sum in row order -- hence bit-identical from run to run,
    which the atomic scatter was not (tools/grad_bitwise_repeat.py) -- on top of what the table held.  33 000 rows: ordered as well since
    round 6 (the fallback to atomics starts at 262 144 rows, and is announced on stderr)."""
    import ctypes as C
    from vln_hamt_amd import _lib as L
    from vln_hamt_amd.ops import _p, _stream
    g = torch.Generator().manual_seed(R + T)
    idx = torch.randint(0, T, (R,), generator=g)
    if R > 100:
        idx[: R // 4] = idx[0]                                   # one table row hit by a quarter of the source rows
    src = rnd(R, W, seed=3)
    base = rnd(T, W, seed=4)
    outs = []
    src_d, idx_d = src.to(DEV), idx.to(DEV)                      # (held: a temporary's block would be handed to the next allocation)
    for rep in range(3):
        junk = torch.randn(1000 * (rep + 1), device=DEV)
        dst = base.to(DEV).clone()
        ws = torch.empty(34 * R, dtype=torch.int32, device=DEV)
        L.check(L.load().hamt_scatter_add_rows_ordered(R, W, _p(src_d), W, 0, _p(idx_d), _p(dst), W, T, _p(ws), _stream()), "scatter")
        torch.cuda.synchronize()
        outs.append(dst.cpu())
        del junk
    ref = base.double().index_add(0, idx, src.double())
    close(outs[0], ref, 2e-5, "ordered scatter")
    if R <= 262144:
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), "ordered scatter is not bit-reproducible"
        # (the defined order: up to 8 source rows of a table row are summed in row order; more: members j = 0 .. 7 (mod 8) in row order,
        # the eight sums added in that order; the total is added to the table row)
        if R <= 6000 and T <= 100:
            want = base.clone()
            for t_ in range(T):
                rows = (idx == t_).nonzero().flatten().tolist()
                if not rows:
                    continue
                if len(rows) <= 8:
                    tot = torch.zeros(W)
                    for r_ in rows:
                        tot = tot + src[r_]
                else:
                    parts = []
                    for k in range(8):
                        a_ = torch.zeros(W)
                        for r_ in rows[k::8]:
                            a_ = a_ + src[r_]
                        parts.append(a_)
                    tot = parts[0]
                    for k in range(1, 8):
                        tot = tot + parts[k]
                want[t_] = base[t_] + tot
            assert torch.equal(outs[0], want), "not the defined summation order"


def test_scatter_add_ordered_skips_rows_outside_the_table():
    """an index outside [0, T) (a padding id, a corrupted batch) used to make one wave add a whole row out of bounds; with T in
    the interface those source rows are skipped -- the table between two guard regions, ids -1, -7, T, T + 5 and 2^31 + 3 among valid
    ones.  Same for hamt_embed_sum_bwd (V), whose scratch is now bounded by ws_bytes: a ws sized by the old HAMT_WS_COLSUM rule makes it
    take the atomic kernel instead of writing 34 R ints into it."""
    from vln_hamt_amd import _lib as L
    from vln_hamt_amd.ops import _p, _stream
    R, T, W = 600, 37, 256
    g = torch.Generator().manual_seed(5)
    idx = torch.randint(0, T, (R,), generator=g)
    bad = torch.tensor([3, 77, 150, 151, 599])
    idx[bad] = torch.tensor([-1, -7, T, T + 5, 2 ** 31 + 3])
    src = rnd(R, W, seed=1)
    guard = 64
    buf = torch.full((guard + T + guard, W), 7.0, device=DEV)
    dst = buf[guard:guard + T]
    dst.zero_()
    ws = torch.empty(34 * R, dtype=torch.int32, device=DEV)
    src_d, idx_d = src.to(DEV), idx.to(DEV)                      # (held: a temporary's block would be handed to the next allocation)
    L.check(L.load().hamt_scatter_add_rows_ordered(R, W, _p(src_d), W, 0, _p(idx_d), _p(dst), W, T, _p(ws), _stream()), "scatter")
    torch.cuda.synchronize()
    ok = torch.ones(R, dtype=torch.bool); ok[bad] = False
    ref = torch.zeros(T, W, dtype=torch.float64).index_add(0, idx[ok], src[ok].double())
    close(dst.cpu(), ref, 2e-5, "ordered scatter with out-of-range ids")
    assert float((buf[:guard] - 7.0).abs().max()) == 0.0 and float((buf[guard + T:] - 7.0).abs().max()) == 0.0, "wrote outside the table"
    # hamt_embed_sum_bwd: V bounds, and a scratch too small for the ranking -> atomics, nothing past ws_bytes
    B, Lq, H, V = 8, 50, 128, 30          # (R = 400: the ranking needs 54 KB, HAMT_WS_COLSUM is 32 KB)
    ids = torch.randint(0, V, (B, Lq), generator=g)
    ids[1, 2], ids[3, 4] = -1, V
    dz = rnd(B * Lq, H, seed=2)
    ids_d, dz_d = ids.to(DEV), dz.to(DEV)
    for ws_bytes in (L.workspace_bytes(L.WS_EMBED_BWD, B * Lq, H), L.workspace_bytes(L.WS_COLSUM, B * Lq, H)):
        wsb = torch.full((ws_bytes // 4 + 4096,), 5.0, device=DEV)
        tab = torch.full((guard + V + guard, H), 7.0, device=DEV)
        dword = tab[guard:guard + V]
        dword.zero_()
        dtyp = torch.zeros(H, device=DEV)
        L.check(L.load().hamt_embed_sum_bwd(B, Lq, H, V, _p(ids_d), _p(dz_d), _p(dword), None, _p(dtyp), _p(wsb), ws_bytes, _stream()), "embed bwd")
        torch.cuda.synchronize()
        okm = ((ids >= 0) & (ids < V)).flatten()
        ref = torch.zeros(V, H, dtype=torch.float64).index_add(0, ids.flatten()[okm], dz[okm].double())
        close(dword.cpu(), ref, 2e-5, "dword with out-of-range ids")
        close(dtyp.cpu(), dz.double().sum(0), 2e-5, "dtype row")
        assert float((tab[:guard] - 7.0).abs().max()) == 0.0 and float((tab[guard + V:] - 7.0).abs().max()) == 0.0, "wrote outside the word table"
        assert float((wsb[ws_bytes // 4:] - 5.0).abs().max()) == 0.0, "wrote past ws_bytes"
    assert L.load().hamt_embed_sum_bwd(B, Lq, H, V, _p(ids_d), _p(dz_d), None, None, _p(dtyp), _p(wsb), 16, _stream()) != 0      # dtype_row needs COLSUM scratch


@pytest.mark.parametrize("B,L,H,V", [(64, 80, 768, 30522), (5, 33, 1024, 100), (2, 7, 132, 50), (3, 9, 1028, 40)])
def test_embed_sum_bwd_shapes(B, L, H, V):
    """hamt_embed_sum_bwd (the text embedder's backward: word rows by atomic adds, position sums through the slice-sum kernel, the
    token-type row through the column-sum kernels) at the step's shape, at wide / narrow / odd-multiple rows, against fp64 autograd;
    gradients ADD to what the tables' gradients hold."""
    ops = _ops()
    g = torch.Generator().manual_seed(B)
    ids = torch.randint(0, V, (B, L), generator=g)
    ids[0, :3] = 1
    word, pos, typ = rnd(V, H, seed=1), rnd(L + 3, H, seed=2), rnd(2, H, seed=3)
    w, p_, t = (x.to(DEV).requires_grad_(True) for x in (word, pos, typ))
    go = rnd(B, L, H, seed=4)
    for rep in range(2):          # second pass: accumulation into existing .grad (autograd adds the fresh tensors)
        ops.embed_sum(ids.to(DEV), w, p_, t).backward(go.to(DEV))
    wr, pr, tr = (x.clone().double().requires_grad_(True) for x in (word, pos, typ))
    (2 * ((wr[ids] + pr[:L][None]) + tr[0][None, None])).backward(go.double())
    close(w.grad, wr.grad, 2e-5, "dword")
    close(p_.grad, pr.grad, 2e-5, "dpos")
    close(t.grad, tr.grad, 2e-5, "dtype")
    assert float(p_.grad[L:].abs().max()) == 0.0 and float(t.grad[1].abs().max()) == 0.0


def test_mean_mulbcast_fill_add():
    ops = _ops()
    B, S, H = 5, 36, 128
    x = rnd(B, S, H, seed=1)
    xg = x.to(DEV).requires_grad_(True)
    y = ops.mean_mid(xg)
    close(y, x.double().mean(1), 1e-6, "mean")
    y.backward(torch.ones_like(y))
    close(xg.grad, torch.full_like(x, 1.0 / S), 1e-6, "mean bwd")
    a, c = rnd(B, S, H, seed=2), rnd(B, 7, H, seed=3)
    ag, cg = a.to(DEV).requires_grad_(True), c.to(DEV).requires_grad_(True)
    y = ops.mul_bcast(ag, cg[:, 0])                            # strided row view like txt_embeds[:, 0]
    assert torch.equal(y.cpu(), a * c[:, :1])
    go = rnd(B, S, H, seed=4)
    y.backward(go.to(DEV))
    close(ag.grad, go * c[:, :1], 1e-6, "mul da")
    close(cg.grad[:, 0], (go * a).sum(1), 1e-5, "mul dc")
    assert float(cg.grad[:, 1:].abs().max()) == 0.0
    flag = (torch.rand(B, S) > 0.5).long()
    s = rnd(B, S, seed=5)
    sg = s.to(DEV).requires_grad_(True)
    f = ops.fill_where_zero(sg, flag.to(DEV), -float("inf"))
    assert torch.equal(torch.isneginf(f).cpu(), flag == 0)
    assert torch.equal(f.cpu()[flag != 0], s[flag != 0])
    f.backward(torch.ones_like(f))
    assert torch.equal(sg.grad.cpu(), (flag != 0).float())
    close(ops.add3(a.to(DEV), x.to(DEV), go.to(DEV)), a + x + go, 1e-6, "add3")


# ------------------------------------------------------------------------------------------ losses
def test_losses():
    ops = _ops()
    R, Cc = 19, 1003
    x = rnd(R, Cc, seed=1, scale=3.0)
    x[2, 5:40] = -float("inf")
    lab = torch.randint(0, Cc, (R,), generator=torch.Generator().manual_seed(2))
    lab[2] = 100
    lab[7] = -100                                   # F.cross_entropy's ignore_index (pretrain_cmt.py:181): loss 0, zero gradient
    buf = torch.zeros(R, 1008, device=DEV)
    buf[:, :Cc] = x.to(DEV)
    xg = buf[:, :Cc].detach().requires_grad_(True)
    loss = ops.cross_entropy(xg, lab.to(DEV))
    xr = x.clone().double().requires_grad_(True)
    ref = F.cross_entropy(xr, lab, reduction="none")
    close(loss, ref, 1e-5, "ce")
    gw = rnd(R, seed=3)
    loss.backward(gw.to(DEV))
    ref.backward(gw.double())
    close(xg.grad, xr.grad, 1e-5, "ce bwd")
    assert float(loss[7]) == 0.0 and float(xg.grad[7].abs().max()) == 0.0
    # any other NEGATIVE label is an ignored row too; a label >= C (a corrupted input: torch raises on it) gives a NaN loss and a NaN
    # gradient row -- detectable, never an out-of-bounds read, never a silent zero
    lab2 = lab.clone(); lab2[3], lab2[4] = -1, Cc
    xg2 = buf[:, :Cc].detach().requires_grad_(True)
    loss2 = ops.cross_entropy(xg2, lab2.to(DEV))
    loss2.backward(gw.to(DEV))
    assert float(loss2[3]) == 0.0 and float(xg2.grad[3].abs().max()) == 0.0
    assert math.isnan(float(loss2[4])) and bool(torch.isnan(xg2.grad[4]).all())
    keep = [i for i in range(R) if i not in (3, 4)]
    close(loss2[keep], ref.detach()[keep], 1e-5, "ce with ignored rows")
    assert bool(torch.isfinite(xg2.grad[keep]).all())
    a, t = rnd(7, 36, 2, seed=4), rnd(7, 36, 2, seed=5)
    ag = a.to(DEV).requires_grad_(True)
    l2 = ops.mse_loss(ag, t.to(DEV))
    assert torch.allclose(l2.cpu(), (a - t) ** 2, atol=1e-6)
    l2.backward(torch.ones_like(l2))
    close(ag.grad, 2 * (a - t), 1e-6, "mse bwd")
    xk = rnd(11, 40, seed=6)
    tk = torch.softmax(rnd(11, 40, seed=7) * 2, -1)
    tk[0, :3] = 0.0
    xkg = xk.to(DEV).requires_grad_(True)
    lk = ops.kl_div_logsoftmax(xkg, tk.to(DEV))
    xkr = xk.clone().double().requires_grad_(True)
    refk = F.kl_div(F.log_softmax(xkr, -1), tk.double(), reduction="none").sum(1)
    close(lk, refk, 1e-5, "kl")
    lk.backward(gw[:11].to(DEV))
    refk.backward(gw[:11].double())
    close(xkg.grad, xkr.grad, 1e-5, "kl bwd")


# ------------------------------------------------------------------------------------------ linear Functions
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_linear_and_packed_linear_autograd(prec):
    ops = _ops()
    tol = 2e-5 if prec == "fp32" else 2e-2
    B, S, K, N = 3, 43, 128, 192
    x = rnd(B, S, K, seed=1)
    lins = [torch.nn.Linear(K, N) for _ in range(3)]
    xg = x.to(DEV).requires_grad_(True)
    glins = [torch.nn.Linear(K, N).to(DEV) for _ in range(3)]
    for a, b in zip(glins, lins):
        a.load_state_dict(b.state_dict())
    for act, tf in ((ops.ACT_NONE, lambda v: v), (ops.ACT_GELU, F.gelu), (ops.ACT_RELU, torch.relu)):
        xr = x.clone().double().requires_grad_(True)
        ld = torch.nn.Linear(K, N).double()
        ld.load_state_dict(lins[0].state_dict())
        ref = tf(ld(xr))
        go = rnd(B, S, N, seed=2)
        ref.backward(go.double())
        xg.grad = None
        glins[0].zero_grad()
        y = ops.linear(xg, glins[0].weight, glins[0].bias, act, prec)
        y.backward(go.to(DEV))
        close(y, ref, tol, f"linear act{act}")
        if prec == "bf16" and act == ops.ACT_RELU:
            # bf16 operand rounding flips the sign of a few near-zero pre-activations, and one flipped mask entry
            # moves dW/dx by a whole |dy * x| term: build the reference gradients with the kernel's OWN mask
            dh = go.double().reshape(-1, N) * (y.detach().cpu().reshape(-1, N) > 0)
            close(xg.grad.reshape(-1, K), dh @ ld.weight.detach(), tol, "linear dx relu")
            close(glins[0].weight.grad, dh.t() @ x.double().reshape(-1, K), tol, "linear dW relu")
            close(glins[0].bias.grad, dh.sum(0), tol, "linear db relu")
            continue
        close(xg.grad, xr.grad, tol, f"linear dx act{act}")
        close(glins[0].weight.grad, ld.weight.grad, tol, f"linear dW act{act}")
        close(glins[0].bias.grad, ld.bias.grad, tol, f"linear db act{act}")
    xr = x.clone().double().requires_grad_(True)
    lds = [torch.nn.Linear(K, N).double() for _ in range(3)]
    for a, b in zip(lds, lins):
        a.load_state_dict(b.state_dict())
    ref = torch.cat([l(xr) for l in lds], -1)
    go = rnd(B * S, 3 * N, seed=3)
    ref.reshape(B * S, 3 * N).backward(go.double())
    xg.grad = None
    for l in glins:
        l.zero_grad()
    y = ops.packed_linear(xg, prec, *glins)
    y.backward(go.to(DEV))
    close(y, ref.reshape(B * S, 3 * N), tol, "packed y")
    close(xg.grad, xr.grad, tol, "packed dx")
    for a, b in zip(glins, lds):
        close(a.weight.grad, b.weight.grad, tol, "packed dW")
        close(a.bias.grad, b.bias.grad, tol, "packed db")


def test_cpu_tensor_is_rejected_loudly():
    ops = _ops()
    from vln_hamt_amd._lib import HamtError
    with pytest.raises(HamtError):
        ops.linear(torch.randn(4, 8), torch.randn(8, 8), torch.randn(8), 0, "fp32")


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_linear_with_residual_epilogue(prec):
    """HAMT_EPI_ADD_AUX: y = x W^T + b + residual in the GEMM epilogue; the residual's gradient is dy."""
    ops = _ops()
    x = rnd(300, 256, seed=1).to(DEV).requires_grad_()
    r = rnd(300, 192, seed=2).to(DEV).requires_grad_()
    lin = torch.nn.Linear(256, 192).to(DEV)
    y = ops.linear(x, lin.weight, lin.bias, ops.ACT_NONE, prec, residual=r)
    go = rnd(300, 192, seed=3).to(DEV)
    y.backward(go)
    xr = bf16_round(x.detach()) if prec == "bf16" else x.detach()
    wr = bf16_round(lin.weight.detach()) if prec == "bf16" else lin.weight.detach()
    ref = xr.double() @ wr.double().t() + lin.bias.detach().double() + r.detach().double()
    close(y, ref, 2e-5, "y")
    close(r.grad, go, 0.0, "d residual")
    gr = bf16_round(go) if prec == "bf16" else go
    close(x.grad, gr.double() @ wr.double(), 2e-5 if prec == "fp32" else 3e-3, "dx")


@pytest.mark.parametrize("off,n", [(0, 1 << 20), (3, 1000003), (1, 2), (2, 7), (5, 4097)])
def test_wire_pack_unpack_bf16(off, n):
    """hamt_wire_pack_bf16 / _unpack_bf16 on views at odd arena offsets: y = bf16(x * scale) (round to nearest even, as
    torch), untouched neighbours, exact widening back."""
    import ctypes as C
    from vln_hamt_amd import _lib as L, ops
    lib = L.load()
    base = torch.randn(off + n + 5, device=DEV)
    stage = torch.full((off + n + 5,), 7.0, device=DEV, dtype=torch.bfloat16)
    x, y = base[off:off + n], stage[off:off + n]
    L.check(lib.hamt_wire_pack_bf16(n, ops._p(x), ops._p(y), 0.125, ops._stream()), "pack")
    assert torch.equal(y, (x * 0.125).to(torch.bfloat16))
    assert float((stage[:off].float() - 7).abs().sum()) == 0 and float((stage[off + n:].float() - 7).abs().sum()) == 0
    keep = base.clone()
    L.check(lib.hamt_wire_unpack_bf16(n, ops._p(y), ops._p(x), ops._stream()), "unpack")
    assert torch.equal(x, y.float())
    assert torch.equal(base[:off], keep[:off]) and torch.equal(base[off + n:], keep[off + n:])


def test_wire_unpack_sumsq():
    """hamt_wire_unpack_sumsq: widen a bf16 chunk of the gradient arena to fp32 (every element) and add the sum of squares of the
    ACTIVE parameters' elements (table flags as hamt_sumsq_table: 0 = no gradient this step, 3 = accounted for elsewhere)."""
    import ctypes as C
    from vln_hamt_amd import _lib as L
    ops = _ops()
    sizes, flags = [1024, 2048, 512, 4096, 1536], [1.0, 0.0, 2.0, 3.0, 1.0]
    ends = torch.tensor(np.cumsum(sizes), dtype=torch.int32, device=DEV)
    hyp = torch.zeros(len(sizes), 4, device=DEV)
    hyp[:, 3] = torch.tensor(flags)
    n_all = sum(sizes)
    y = rnd(n_all, seed=4).to(torch.bfloat16).to(DEV)
    first, n = 512, n_all - 512 - 1024                      # a chunk that starts and ends inside parameters
    g = torch.full((n_all,), float("nan"), device=DEV)
    out = torch.full((1,), 3.0, device=DEV)
    ws = torch.empty(1024, device=DEV)
    L.check(L.load().hamt_wire_unpack_sumsq(first, n, ops._p(y[first:first + n]), ops._p(g[first:first + n]), ops._p(ends), ops._p(hyp), len(sizes),
                                            ops._p(out), 1, ops._p(ws), ops._stream()), "hamt_wire_unpack_sumsq")
    torch.cuda.synchronize()
    assert torch.equal(g[first:first + n], y[first:first + n].float()) and bool(torch.isnan(g[:first]).all()) and bool(torch.isnan(g[first + n:]).all())
    act = torch.cat([torch.full((sz,), f not in (0.0, 3.0)) for sz, f in zip(sizes, flags)]).to(DEV)
    want = 3.0 + float((y.double() ** 2)[first:first + n][act[first:first + n]].sum())
    assert abs(float(out) - want) <= 1e-5 * want, (float(out), want)


@pytest.mark.parametrize("sparse", [0, 2])
@pytest.mark.parametrize("nparams", [5, 300, 1500])
def test_sumsq_table_active_only(sparse, nparams):
    """hamt_sumsq_table over an arena range that starts and ends inside parameters, with every kind of table flag (0 = no gradient: the
    slot holds NaN; 1 / 2 active; 3 = accounted for by the weight-gradient tiles), plain element ranges and HAMT_SUMSQ_SPARSE (active
    elements spread evenly over the blocks: the GEMM-weight region of a step) against fp64."""
    from vln_hamt_amd import _lib as L
    ops = _ops()
    rs = np.random.RandomState(nparams)
    sizes = (rs.randint(1, 2000, size=nparams) * 4).tolist()
    flags = rs.choice([0.0, 1.0, 2.0, 3.0], size=nparams, p=[0.3, 0.1, 0.1, 0.5]).tolist()
    ends = torch.tensor(np.cumsum(sizes), dtype=torch.int32, device=DEV)
    hyp = torch.zeros(nparams, 4, device=DEV)
    hyp[:, 3] = torch.tensor(flags)
    n_all = sum(sizes)
    g = rnd(n_all, seed=5).to(DEV)
    act = torch.cat([torch.full((sz,), f in (1.0, 2.0)) for sz, f in zip(sizes, flags)]).to(DEV)
    dead = torch.cat([torch.full((sz,), f == 0.0) for sz, f in zip(sizes, flags)]).to(DEV)
    g[dead] = float("nan")
    first = (sizes[0] // 8) * 4
    n = n_all - first - (sizes[-1] // 8) * 4
    out = torch.full((1,), 2.0, device=DEV)
    ws = torch.empty(1024, device=DEV)
    L.check(L.load().hamt_sumsq_table(first, n, ops._p(g[first:first + n]), ops._p(ends), ops._p(hyp), nparams, ops._p(out), 1 | sparse,
                                      ops._p(ws), ops._stream()), "hamt_sumsq_table")
    want = 2.0 + float((g.double() ** 2)[first:first + n][act[first:first + n]].sum())
    assert abs(float(out) - want) <= 1e-5 * want, (float(out), want)
    L.check(L.load().hamt_sumsq_table(first, n, ops._p(g[first:first + n]), ops._p(ends), ops._p(hyp), nparams, ops._p(out), sparse,
                                      ops._p(ws), ops._stream()), "hamt_sumsq_table")
    assert abs(float(out) - (want - 2.0)) <= 1e-5 * want, (float(out), want - 2.0)


@pytest.mark.parametrize("normalize", ["total", "batch", "none"])
@pytest.mark.parametrize("feedback", ["sample", "teacher"])
def test_a2c_loss_vs_reference_goldens(normalize, feedback):
    """ops.a2c_loss + models.model_HAMT.Critic (fp32 mode) against tests/golden/a2c.npz: the REFERENCE's own A2C statements
    (agent_cmt.py:476-517, compiled from its file at generation time, oracle/gen_goldens.py a2c) on scripted rollout lists -- loss, logged
    sums, gradients w.r.t. log-probabilities, hidden states (through the critic), entropies and the critic's parameters."""
    import types
    import numpy as np
    from _util import grad_probe, load_npz, sub
    from oracle.hamt_oracle import make_state_dict
    from vln_hamt_amd import ops
    from vln_hamt_amd.models.model_HAMT import Critic
    store = load_npz("a2c.npz")
    sd = make_state_dict({"state2value.0.weight": (512, 768), "state2value.0.bias": (512,), "state2value.3.weight": (1, 512),
                          "state2value.3.bias": (1,)}, seed=int(store["meta/critic_seed"]))
    critic = Critic(types.SimpleNamespace(dropout=0.5, hamt_precision="fp32"))
    critic.load_state_dict(sd, strict=True)
    critic = critic.to(DEV).eval()
    T, B = store["in/logp"].shape
    d = lambda a: torch.from_numpy(a).to(DEV).requires_grad_(True)
    logp, hidden, ent = d(store["in/logp"]), d(store["in/hidden"]), d(store["in/ent"])
    value = critic(hidden.reshape(T * B, -1)).view(T, B)
    with torch.no_grad():
        lv = critic(torch.from_numpy(store["in/last_h"]).to(DEV))
        lv = torch.where(torch.from_numpy(store["in/ended"]).to(DEV), torch.zeros_like(lv), lv)      # agent_cmt.py:480-484
    loss, parts = ops.a2c_loss(logp, value, torch.from_numpy(store["in/rewards"]).to(DEV), torch.from_numpy(store["in/masks"]).to(DEV), last_value=lv,
                               entropy=ent if feedback == "sample" else None, gamma=0.9, entropy_weight=0.01, normalize=normalize)
    loss.backward()
    pre = f"{normalize}_{feedback}/"
    ref = float(store[pre + "rl_loss"])
    assert abs(float(loss) - ref) <= 1e-5 * max(1.0, abs(ref)), (float(loss), ref)
    assert abs(float(parts["policy"]) - float(store[pre + "policy_sum"])) <= 1e-4 * max(1.0, abs(float(store[pre + "policy_sum"])))
    assert abs(float(parts["critic"]) - float(store[pre + "critic_sum"])) <= 1e-4 * max(1.0, abs(float(store[pre + "critic_sum"])))
    close(logp.grad, torch.from_numpy(store[pre + "d_logp"]), 1e-5, "d logp")
    close(hidden.grad, torch.from_numpy(store[pre + "d_hidden"]), 1e-4, "d hidden")
    if feedback == "sample":
        close(ent.grad, torch.from_numpy(store[pre + "d_ent"]), 1e-5, "d entropy")
    named = dict(critic.named_parameters())
    for k, v in sub(store, pre + "d_critic/").items():
        close(named[k].grad, torch.from_numpy(v), 1e-4, k)
    for k, v in sub(store, pre + "d_critic_norm/").items():
        assert abs(float(named[k].grad.double().norm()) - float(v)) <= 1e-4 * float(v), k
        pr = store[pre + "d_critic_probe/" + k]
        assert float(np.abs(grad_probe(named[k].grad) - pr).max()) <= 1e-4 * max(1.0, float(np.abs(pr).max())), k


@pytest.mark.parametrize("normalize,with_ent", [("total", True), ("batch", False), ("none", True)])
def test_a2c_loss_vs_reference_restatement(normalize, with_ent):
    """ops.a2c_loss (one scan kernel over [T, B]) against the statement-by-statement restatement of the agent's loop
    (oracle.hamt_oracle.a2c_loss_ref <- finetune_src/r2r/agent_cmt.py:476-518): loss, logged sums and the gradients w.r.t. the
    policy log-probabilities, the critic values and the entropies.  (The restatement itself is pinned against the reference's own
    statements: tests/test_oracle_goldens.py::test_a2c_restatement_matches_the_reference_block.)"""
    import numpy as np
    from oracle.hamt_oracle import a2c_loss_ref
    from vln_hamt_amd import ops
    T, B = 7, 8
    g = torch.Generator().manual_seed(3)
    logp = (-torch.rand(T, B, generator=g) * 3).requires_grad_(True)
    value = torch.randn(T, B, generator=g).requires_grad_(True)
    ent = torch.rand(T, B, generator=g).requires_grad_(True)
    rewards = (torch.randn(T, B, generator=g) * 2).numpy()
    ended_at = torch.randint(2, T + 2, (B,), generator=g).numpy()            # some episodes run past the rollout (not ended)
    masks = (np.arange(T)[:, None] < ended_at[None]).astype(np.float32)
    rewards = rewards * masks
    ended = ended_at <= T
    last_value = torch.randn(B, generator=g)
    ref, logs = a2c_loss_ref([logp[t] for t in range(T)], [value[t] for t in range(T)], [rewards[t] for t in range(T)],
                             [masks[t] for t in range(T)], last_value, ended, [ent[t] for t in range(T)] if with_ent else None,
                             gamma=0.9, entropy_loss_weight=0.01, normalize_loss=normalize)
    ref.backward()
    d = lambda t: t.detach().to("cuda").requires_grad_(True)
    lp, va, en = d(logp), d(value), d(ent)
    lv = torch.where(torch.from_numpy(ended), torch.zeros(B), last_value).to("cuda")
    loss, parts = ops.a2c_loss(lp, va, torch.from_numpy(rewards).to("cuda"), torch.from_numpy(masks).to("cuda"), last_value=lv,
                               entropy=en if with_ent else None, gamma=0.9, entropy_weight=0.01, normalize=normalize)
    loss.backward()
    assert abs(float(loss) - float(ref)) <= 1e-5 * max(1.0, abs(float(ref))), (float(loss), float(ref))
    assert abs(float(parts["policy"]) - sum(logs["policy_loss"])) <= 1e-4 * max(1.0, abs(sum(logs["policy_loss"])))
    assert abs(float(parts["critic"]) - sum(logs["critic_loss"])) <= 1e-4 * max(1.0, abs(sum(logs["critic_loss"])))
    for a, r, n in ((lp, logp, "logp"), (va, value, "value")) + (((en, ent, "ent"),) if with_ent else ()):
        err = float((a.grad.cpu() - r.grad).abs().max())
        assert err <= 1e-5 * max(1.0, float(r.grad.abs().max())), (n, err)
