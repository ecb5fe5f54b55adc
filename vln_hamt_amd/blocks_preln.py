"""Block-level autograd Functions for PRE-LayerNorm transformer blocks (the ViT backbone, vision_transformer.py:181-198):

    x + proj(attn(qkv(LN(x))))      -> PreLnAttnFn
    x + fc2(gelu(fc1(LN(x))))       -> PreLnMlpFn

Same construction as blocks.py (post-LN, BERT): every tensor between two kernels has the dtype/layout the next kernel
wants.  LayerNorm emits ONLY the bf16 image of its output (no fp32 y, no copy of its input: the block input itself is
what backward needs); the residual add rides in the last GEMM's epilogue (HAMT_EPI_ADD_AUX); GELU and its derivative come
from one erf evaluation in fc1's epilogue; in backward the residual gradient is added inside the LayerNorm-backward
kernel (hamt_ln_bwd_add); weight / bias gradients are queued for the grouped end-of-pass launch (wgrad.py).
The branch dropouts (proj_drop, Mlp.drop after the activation and after fc2) ride in the producing GEMM's epilogue
(HAMT_EPI_DROPOUT); backward re-creates the mask while casting dy to its bf16 operand image (hamt_cast_pad_bf16_dropout),
and the post-activation mask is folded into the stored gelu'.  Attention-probability dropout is handled in the kernel.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L
from .blocks import _attn_desc, _wgrad, _zeros_or_empty
from .ops import _p, _rup, _stream, cast_pad16, cast_pad16_dropout, gemm, next_call_id, rng_state, weight_operand


def _ln16(x2, gamma, beta, eps):
    """bf16 image [Mp, D] of LayerNorm(x2) + the row statistics"""
    M, D = x2.shape
    Mp = _rup(M)
    y16 = torch.empty(Mp, D, dtype=torch.bfloat16, device=x2.device)
    mean = torch.empty(M, dtype=torch.float32, device=x2.device)
    rstd = torch.empty(M, dtype=torch.float32, device=x2.device)
    d = L.LnDesc(M, D, float(eps), 0.0, 0.0, 0, Mp)
    L.check(L.load().hamt_ln_fwd(C.byref(d), _p(x2), None, _p(gamma), _p(beta), None, None, _p(y16), _p(mean), _p(rstd),
                                 _p(rng_state(x2.device)), _stream()), "hamt_ln_fwd")
    return y16, mean, rstd


def _ln_bwd_add(dln, x2, mean, rstd, gamma, eps, dy2, params=None):
    """(LayerNorm-backward(dln) + dy2, dgamma, dbeta); with `params` = (gamma, beta) PARAMETERS their gradients are summed by the
    grouped end-of-pass launch and published there (returned as None), as in ops._ln_bwd"""
    from .ops import _can_defer_ln, _defer_ln_reduce
    M, D = x2.shape
    dev = x2.device
    dx = torch.empty(M, D, dtype=torch.float32, device=dev)
    red = torch.empty(3, D, dtype=torch.float32, device=dev)
    ws = torch.empty(L.workspace_bytes(L.WS_LN_BWD, M, D) // 4, dtype=torch.float32, device=dev)
    d = L.LnDesc(M, D, float(eps), 0.0, 0.0, 0, 0)
    defer = params is not None and _can_defer_ln((params[0], params[1], None), dev, D)
    L.check(L.load().hamt_ln_bwd_add(C.byref(d), _p(dln), _p(x2), _p(mean), _p(rstd), _p(gamma), _p(dy2), _p(dx),
                                     None if defer else _p(red[0]), None if defer else _p(red[1]), _p(ws), _stream()), "hamt_ln_bwd_add")
    if defer:
        _defer_ln_reduce(M, D, ws, red, False, (params[0], params[1], None), dev)
        return dx, None, None
    return dx, red[0], red[1]


class PreLnAttnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, heads, p_attn, p_proj, eps, gamma, beta, wqkv, bqkv, wproj, bproj):
        B, S, D = x.shape
        M = B * S
        dev = x.device
        x2 = x.reshape(M, D)
        x2 = x2 if x2.is_contiguous() else x2.contiguous()
        y16, mean, rstd = _ln16(x2, gamma.detach(), beta.detach(), eps)
        Mp = y16.shape[0]
        qkv16 = torch.empty(Mp, 3 * D, dtype=torch.bfloat16, device=dev)
        gemm(y16[:M], weight_operand(wqkv, "bf16"), qkv16[:M], bias=bqkv.detach())
        ctx16 = _zeros_or_empty(Mp, M, D, dev)
        lse = torch.empty(B * heads * S, dtype=torch.float32, device=dev)
        cid = next_call_id()
        d = _attn_desc(B, heads, S, S, D, 3 * D, 3 * D, 3 * D, p_attn, cid)
        q, k, v = qkv16[:, :D], qkv16[:, D:2 * D], qkv16[:, 2 * D:]
        L.check(L.load().hamt_attn_small_fwd(C.byref(d), _p(q), _p(k), _p(v), None, _p(ctx16), _p(lse), _p(rng_state(dev)), _stream()),
                "hamt_attn_small_fwd")
        y = torch.empty(M, D, dtype=torch.float32, device=dev)
        cid_p = next_call_id() if p_proj > 0.0 else 0
        gemm(ctx16[:M], weight_operand(wproj, "bf16"), y, bias=bproj.detach(), epilogue=L.EPI_ADD_AUX, aux=x2.detach(), drop=(p_proj, cid_p))
        ctx.save_for_backward(x2, y16, qkv16, ctx16, lse, mean, rstd, gamma, wqkv, bqkv, wproj, bproj)
        ctx.ln_params = (gamma, beta)
        ctx.meta = (B, S, D, M, heads, float(p_attn), float(eps), cid, float(p_proj), cid_p)
        return y.view(B, S, D)

    @staticmethod
    def backward(ctx, dy):
        x2, y16, qkv16, ctx16, lse, mean, rstd, gamma, wqkv, bqkv, wproj, bproj = ctx.saved_tensors
        B, S, D, M, heads, p_attn, eps, cid, p_proj, cid_p = ctx.meta
        dev = dy.device
        Mp = y16.shape[0]
        dy2 = dy.reshape(M, D)
        dy2 = dy2 if dy2.is_contiguous() else dy2.contiguous()
        dy16 = cast_pad16_dropout(dy2, p_proj, cid_p) if p_proj > 0.0 else cast_pad16(dy2, D)
        dctx16 = torch.empty(Mp, D, dtype=torch.bfloat16, device=dev)
        gemm(dy16[:M], weight_operand(wproj, "bf16"), dctx16[:M], b_kmajor=True)
        dwp, dbp = _wgrad(wproj, bproj, dy16, ctx16, M)
        dqkv16 = _zeros_or_empty(Mp, M, 3 * D, dev)
        d = _attn_desc(B, heads, S, S, D, 3 * D, 3 * D, 3 * D, p_attn, cid)
        q, k, v = qkv16[:, :D], qkv16[:, D:2 * D], qkv16[:, 2 * D:]
        dq, dk, dv = dqkv16[:, :D], dqkv16[:, D:2 * D], dqkv16[:, 2 * D:]
        L.check(L.load().hamt_attn_small_bwd(C.byref(d), _p(q), _p(k), _p(v), None, _p(ctx16), _p(dctx16), _p(lse), None,
                                             _p(dq), _p(dk), _p(dv), _p(rng_state(dev)), _stream()), "hamt_attn_small_bwd")
        dln = torch.empty(M, D, dtype=torch.float32, device=dev)
        gemm(dqkv16[:M], weight_operand(wqkv, "bf16"), dln, b_kmajor=True)
        dwq, dbq = _wgrad(wqkv, bqkv, dqkv16, y16, M)
        dx, dgamma, dbeta = _ln_bwd_add(dln, x2, mean, rstd, gamma.detach(), eps, dy2, params=ctx.ln_params)
        return dx.view(B, S, D), None, None, None, None, dgamma, dbeta, dwq, dbq, dwp, dbp


class PreLnMlpFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p_drop, eps, gamma, beta, w1, b1, w2, b2, grad_on=True):
        shp = x.shape
        D = shp[-1]
        x2 = x.reshape(-1, D)
        x2 = x2 if x2.is_contiguous() else x2.contiguous()
        M = x2.shape[0]
        dev = x.device
        I = w1.shape[0]
        y16, mean, rstd = _ln16(x2, gamma.detach(), beta.detach(), eps)
        Mp = y16.shape[0]
        g16 = _zeros_or_empty(Mp, M, I, dev)
        cid1, cid2 = (next_call_id(), next_call_id()) if p_drop > 0.0 else (0, 0)
        # (needs_input_grad reports the parameters' requires_grad whatever the caller's grad mode, and forward itself always runs
        # with grad mode off: the caller passes its own torch.is_grad_enabled())
        if grad_on and any(ctx.needs_input_grad):
            pre = torch.empty(M, I, dtype=torch.bfloat16, device=dev)      # gelu'(fc1), from the same erf evaluation as gelu
            gemm(y16[:M], weight_operand(w1, "bf16"), g16[:M], bias=b1.detach(), epilogue=L.EPI_GELU_GRAD, aux=pre, drop=(p_drop, cid1))
        else:                                    # no-grad panorama pass (image_vilmodel.py:44-52): nobody reads gelu'
            pre = None
            gemm(y16[:M], weight_operand(w1, "bf16"), g16[:M], bias=b1.detach(), epilogue=L.EPI_GELU, drop=(p_drop, cid1))
        y = torch.empty(M, D, dtype=torch.float32, device=dev)
        gemm(g16[:M], weight_operand(w2, "bf16"), y, bias=b2.detach(), epilogue=L.EPI_ADD_AUX, aux=x2.detach(), drop=(p_drop, cid2))
        if pre is not None:
            ctx.save_for_backward(x2, y16, g16, pre, mean, rstd, gamma, w1, b1, w2, b2)
        ctx.ln_params = (gamma, beta)
        ctx.meta = (shp, M, D, I, float(eps), float(p_drop), cid2)
        return y.view(shp)

    @staticmethod
    def backward(ctx, dy):
        x2, y16, g16, pre, mean, rstd, gamma, w1, b1, w2, b2 = ctx.saved_tensors
        shp, M, D, I, eps, p_drop, cid2 = ctx.meta
        dev = dy.device
        Mp = y16.shape[0]
        dy2 = dy.reshape(M, D)
        dy2 = dy2 if dy2.is_contiguous() else dy2.contiguous()
        dy16 = cast_pad16_dropout(dy2, p_drop, cid2) if p_drop > 0.0 else cast_pad16(dy2, D)
        dh16 = _zeros_or_empty(Mp, M, I, dev)
        gemm(dy16[:M], weight_operand(w2, "bf16"), dh16[:M], b_kmajor=True, epilogue=L.EPI_MUL_AUX, aux=pre)       # dG * (masked) gelu'
        dw2, db2 = _wgrad(w2, b2, dy16, g16, M)
        dln = torch.empty(M, D, dtype=torch.float32, device=dev)
        gemm(dh16[:M], weight_operand(w1, "bf16"), dln, b_kmajor=True)
        dw1, db1 = _wgrad(w1, b1, dh16, y16, M)
        dx, dgamma, dbeta = _ln_bwd_add(dln, x2, mean, rstd, gamma.detach(), eps, dy2, params=ctx.ln_params)
        return dx.view(shp), None, None, dgamma, dbeta, dw1, db1, dw2, db2, None


def usable(prec: str, x: torch.Tensor) -> bool:
    from . import blocks
    return blocks.ENABLED and prec == "bf16" and x.is_cuda and x.shape[-1] % 64 == 0
