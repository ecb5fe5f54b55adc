"""Block-level autograd Functions for the bf16 path: one Function per transformer sub-block, so that every tensor
between two kernels of a block has exactly the dtype/layout the next kernel wants and nothing is cast, transposed or
re-added by a separate pass:

    BertAttention      (vilmodel.py:146-156)  -> SelfAttnBlockFn   = LN(dropout(W_o attn(W_qkv x)) + x)
    BertXAttention     (vilmodel.py:351-360)  -> CrossAttnBlockFn  = LN(dropout(W_o attn(W_q x, W_kv c)) + x)
    BertIntermediate + BertOutput (:159-185)  -> FfnBlockFn        = LN(dropout(W_2 gelu(W_1 x)) + x)

Data flow inside a block (M rows padded to a multiple of 64 only in the bf16 images):
  * GEMM inputs are bf16 images written by their producers: LayerNorm kernels emit y16 next to the fp32 residual
    stream, the QKV / GELU epilogues and the attention kernel write bf16 directly;
  * backward: the LayerNorm-backward kernel emits the dropout-masked gradient as a bf16 image plus its column sums
    (= the dense layer's bias gradient); dgrad GEMMs run in "NN" form on the bf16 weight arena, wgrad GEMMs in "TN"
    form on the saved bf16 images -- no transposes; GELU' is applied in the dgrad epilogue; the residual gradient is
    accumulated by the last dgrad's epilogue (C += ...), not by a separate add.
  * query/key/value (key/value) weights that sit back to back in the optimizer's arena are used as ONE [3H,H]
    ([2H,H]) operand: one projection GEMM, one dgrad, one wgrad.
The fine-grained Functions in ops.py remain the fp32-mode path and the stand-alone module path.
"""
from __future__ import annotations

import ctypes as C
import math
import os

import torch

from . import _lib as L
from . import ops
from . import wgrad
from .ops import _ln_bwd, _ln_fwd, _p, _rup, _stream, cast_pad16, colsum, gemm, next_call_id, rng_state, shadow16, weight_operand

ENABLED = os.environ.get("HAMT_NO_FUSED_BLOCKS") is None
NO_PACKED = os.environ.get("HAMT_NO_PACKED") is not None      # ablation switches
NO_SHADOW = os.environ.get("HAMT_NO_SHADOW") is not None


# dtype of the dense outputs that feed a LayerNorm (attention output projection, second FFN layer): two bytes per element like every
# other dense output of the bf16 path, but IEEE half, not bf16 -- the value is consumed once, by the LayerNorm, in fp32 (never an MFMA
# operand), it is O(1 .. 100), and half's 10 mantissa bits make the rounding of this interface 8 x finer than bf16's 7 at the same HBM
# bytes: the bf16-mode head outputs at B = 64 went from 0.99 / 1.03e-2 (SAR / ITM) to ~6e-3 of the fp32 reference (round 6; what fp32
# outputs give at + 4 bytes per element: HAMT_DENSE_OUT_F32=1).  HAMT_DENSE_OUT_BF16=1: bf16 as before (what a linear returns under autocast)
O_DTYPE = (torch.float32 if os.environ.get("HAMT_DENSE_OUT_F32") == "1" else
           (torch.bfloat16 if os.environ.get("HAMT_DENSE_OUT_BF16") == "1" else torch.float16))


# the saved gelu' image of the FFN blocks: one byte per element (hamt.h HAMT_U8G: 0.005 q - 0.13, |error| <= 0.0025 -- what bf16's rounding
# of a value near 1 is) instead of bf16: 16 MB less to write in the FFN-1 epilogue and to read in the FFN-2 dgrad epilogue per 5120 rows.
# HAMT_GELUP_U8=1 selects it; measured (profiles/r05_epilogue.txt) it buys nothing, the epilogue's cost was never its bytes
GELUP_DTYPE = torch.uint8 if os.environ.get("HAMT_GELUP_U8") == "1" else torch.bfloat16

ZERO_PAD = os.environ.get("HAMT_ZERO_PAD_ROWS") == "1"       # ablation: zero the padding rows as before (one fill launch per buffer)


def _zeros_or_empty(rows_total, rows_valid, cols, device, dtype=torch.bfloat16):
    """bf16 image whose rows >= rows_valid are reduction padding of the weight-gradient GEMMs.  Nothing has to be written there:
    every consumer names the valid rows (`_wgrad` / `_proj_bwd`: hamt_wgrad_desc.K_valid, `gemm(k_valid=)`), the kernels re-read
    the last valid row for the padding and zero its products -- at B = 16 these fills were 27 launches per step."""
    t = torch.empty(rows_total, cols, dtype=dtype, device=device)
    if ZERO_PAD and rows_total != rows_valid:
        t[rows_valid:].zero_()
    return t


def _x16_of(x, x2):
    s = None if NO_SHADOW else shadow16(x)
    if s is not None and s.shape[1] == x2.shape[1]:
        return s
    return cast_pad16(x2, x2.shape[1])


def _packed(ws, bs):
    """If the parameters `ws` (and biases `bs`) are adjacent in the optimizer's arena return ([sumN,K] bf16 view,
    [sumN] fp32 bias view), else None."""
    a0 = getattr(ws[0], "_hamt_arena16", None)
    if NO_PACKED or a0 is None or not all(getattr(w, "_hamt_arena16", None) is not None and ops.arena16_valid(w, w._hamt_arena16) for w in ws):
        return None
    flat_p = a0[1]
    K = ws[0].shape[1]
    for w0, w1 in zip(ws[:-1], ws[1:]):
        if w0.data_ptr() + w0.numel() * 4 != w1.data_ptr() or w1.shape[1] != K:
            return None
    for b0, b1 in zip(bs[:-1], bs[1:]):
        if b0.data_ptr() + b0.numel() * 4 != b1.data_ptr():
            return None
    base, n = flat_p.data_ptr(), flat_p.numel() * 4
    if not (base <= ws[0].data_ptr() < base + n and base <= bs[0].data_ptr() < base + n):
        return None
    flat16 = a0[0]._base if a0[0]._base is not None else a0[0]
    woff = (ws[0].data_ptr() - base) // 4
    boff = (bs[0].data_ptr() - base) // 4
    N = sum(w.shape[0] for w in ws)
    return flat16[woff:woff + N * K].view(N, K), flat_p[boff:boff + N].detach()


def _attn_desc(B, heads, Sq, Sk, H, ldq, ldk, ldv, p_drop, cid):
    return L.AttnDesc(B, heads, Sq, Sk, H // heads, ldq, ldk, ldv, H, L.HAMT_BF16, L.HAMT_BF16, 1.0 / math.sqrt(H // heads),
                      float(p_drop), cid, L.PREC_BF16)


def _proj(x16, M, ws, bs, out16):
    """out16[:M, :] = x16[:M] @ [W_0; W_1; ...]^T + [b_0; b_1; ...] (bf16 out), one GEMM when the weights are adjacent"""
    pk = _packed(ws, bs)
    if pk is not None:
        gemm(x16[:M], pk[0], out16[:M], bias=pk[1])
        return
    c = 0
    for w, b in zip(ws, bs):
        n = w.shape[0]
        gemm(x16[:M], weight_operand(w, "bf16"), out16[:M, c:c + n], bias=b.detach())
        c += n


def _proj_bwd(d16, M, x16, ws, bs, dx_accum_into=None, need_dx=True):
    """gradients of _proj: d16 [Mp, sumN] bf16 (rows >= M zero).  Returns (dx fp32 [M,K] or None, [dW_i], [db_i]);
    with dx_accum_into the input gradient is accumulated into that fp32 buffer (residual-gradient fusion)."""
    K = ws[0].shape[1]
    dev = d16.device
    pk = _packed(ws, bs)
    N = d16.shape[1]
    dx = None
    if need_dx:
        dx = dx_accum_into if dx_accum_into is not None else torch.empty(M, K, dtype=torch.float32, device=dev)
        acc = L.EPI_ACCUM if dx_accum_into is not None else 0
        if pk is not None:
            gemm(d16[:M], pk[0], dx, b_kmajor=True, epilogue=acc)
        else:
            c = 0
            for i, w in enumerate(ws):
                n = w.shape[0]
                gemm(d16[:M, c:c + n], weight_operand(w, "bf16"), dx, b_kmajor=True, epilogue=acc if i == 0 else L.EPI_ACCUM)
                c += n
    if all(wgrad.eligible(w, d16[:, :w.shape[0]], x16) for w in ws):
        c = 0
        for w, b in zip(ws, bs):                 # queued: one grouped launch per backward pass (wgrad.py)
            n = w.shape[0]
            wgrad.defer(w, b, d16[:, c:c + n], x16, rows=M)
            c += n
        return dx, [None] * len(ws), [None] * len(ws)
    dW = torch.empty(N, K, dtype=torch.float32, device=dev)
    gemm(d16, x16, dW, a_kmajor=True, b_kmajor=True, k_valid=M)
    db = colsum(d16[:M])
    dws, dbs, c = [], [], 0
    for w in ws:
        n = w.shape[0]
        dws.append(dW[c:c + n])
        dbs.append(db[c:c + n])
        c += n
    return dx, dws, dbs


def _wgrad(w, b, dy16, x16, M):
    """(dW, db) now, or (None, None) after queueing them for the grouped end-of-pass launch"""
    if wgrad.eligible(w, dy16, x16):
        wgrad.defer(w, b, dy16, x16, rows=M)
        return None, None
    dw = torch.empty(w.shape, dtype=torch.float32, device=dy16.device)
    gemm(dy16, x16, dw, a_kmajor=True, b_kmajor=True, k_valid=M)
    return dw, (colsum(dy16[:M]) if b is not None else None)


# ================================================================================================= self attention block
class SelfAttnBlockFn(torch.autograd.Function):
    """`seq` = None for a padded batch x [B, S, H] with an additive key mask, or (cu_seqlens int32 [B + 1], B, S_max) for a PACKED
    batch x [M, H]: B sequences back to back, sequence b = rows [cu[b], cu[b + 1]), cu[B] == M (rows added to reach a bucketed M
    form sequences of their own: every row is written by every kernel, nothing is ever NaN), attention by hamt_attn_varlen_* --
    every other kernel of the block is row-wise and does not care."""

    @staticmethod
    def forward(ctx, x, add_mask, heads, p_attn, p_hidden, eps, wq, bq, wk, bk, wv, bv, wo, bo, gamma, beta, seq=None):
        if seq is not None:
            cu, B, S = seq
            M, H = x.shape
            assert add_mask is None
        else:
            B, S, H = x.shape
            M = B * S
        dev = x.device
        x2 = x.reshape(M, H)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        x16 = _x16_of(x, x2)
        Mp = x16.shape[0]
        qkv16 = torch.empty(Mp, 3 * H, dtype=torch.bfloat16, device=dev)
        _proj(x16, M, (wq, wk, wv), (bq, bk, bv), qkv16)
        mask2 = add_mask.reshape(B, S).to(torch.float32).contiguous() if add_mask is not None else None
        ctx16 = _zeros_or_empty(Mp, M, H, dev)
        lse = torch.empty(B * heads * S, dtype=torch.float32, device=dev)
        cid = next_call_id()
        rng = rng_state(dev)
        d = _attn_desc(B, heads, S, S, H, 3 * H, 3 * H, 3 * H, p_attn, cid)
        q, k, v = qkv16[:, :H], qkv16[:, H:2 * H], qkv16[:, 2 * H:]
        if seq is not None:      # (every row of x belongs to a sequence: bucket-padding rows form sequences of their own)
            L.check(L.load().hamt_attn_varlen_fwd(C.byref(d), _p(q), _p(k), _p(v), _p(cu), _p(ctx16), _p(lse), _p(rng), _stream()),
                    "hamt_attn_varlen_fwd")
        else:
            L.check(L.load().hamt_attn_small_fwd(C.byref(d), _p(q), _p(k), _p(v), _p(mask2), _p(ctx16), _p(lse), _p(rng), _stream()),
                    "hamt_attn_small_fwd")
        o = torch.empty(M, H, dtype=O_DTYPE, device=dev)      # bf16: what a linear returns under autocast (ops._ln_fwd reads it as such)
        gemm(ctx16[:M], weight_operand(wo, "bf16"), o, bias=bo.detach())
        y, y16, z, mean, rstd, cid_ln = _ln_fwd(o, x2, gamma.detach(), beta.detach(), eps, p_hidden, 0.0, True)
        ctx.save_for_backward(x16, qkv16, ctx16, lse, mask2, z, mean, rstd, wq, bq, wk, bk, wv, bv, wo, bo, gamma)
        ctx.ln_params = (gamma, beta, bo)          # their gradients come from the LayerNorm-backward partials (ops._ln_bwd)
        ctx.meta = (B, S, H, M, heads, float(p_attn), float(p_hidden), float(eps), cid, cid_ln)
        ctx.cu = seq[0] if seq is not None else None
        ctx.mark_non_differentiable(y16)
        ctx.set_materialize_grads(False)     # else autograd zero-fills a bf16 [Mp,H] "gradient" of y16 per backward
        return y.view(x.shape), y16

    @staticmethod
    def backward(ctx, dy, _unused=None):
        if dy is None:
            return (None,) * 17
        x16, qkv16, ctx16, lse, mask2, z, mean, rstd, wq, bq, wk, bk, wv, bv, wo, bo, gamma = ctx.saved_tensors
        B, S, H, M, heads, p_attn, p_hidden, eps, cid, cid_ln = ctx.meta
        dev = dy.device
        Mp = x16.shape[0]
        dz, _, dx16, dgamma, dbeta, dbo = _ln_bwd(dy.reshape(M, H).contiguous(), z, mean, rstd, gamma.detach(), eps, p_hidden, 0.0,
                                                  cid_ln, False, True, True, params=ctx.ln_params)
        dctx16 = torch.empty(Mp, H, dtype=torch.bfloat16, device=dev)
        gemm(dx16[:M], weight_operand(wo, "bf16"), dctx16[:M], b_kmajor=True)
        dwo, _ = _wgrad(wo, None, dx16, ctx16, M)
        dqkv16 = _zeros_or_empty(Mp, M, 3 * H, dev)
        d = _attn_desc(B, heads, S, S, H, 3 * H, 3 * H, 3 * H, p_attn, cid)
        q, k, v = qkv16[:, :H], qkv16[:, H:2 * H], qkv16[:, 2 * H:]
        dq, dk, dv = dqkv16[:, :H], dqkv16[:, H:2 * H], dqkv16[:, 2 * H:]
        if ctx.cu is not None:
            L.check(L.load().hamt_attn_varlen_bwd(C.byref(d), _p(q), _p(k), _p(v), _p(ctx.cu), _p(ctx16), _p(dctx16), _p(lse),
                                                  _p(dq), _p(dk), _p(dv), _p(rng_state(dev)), _stream()), "hamt_attn_varlen_bwd")
        else:
            L.check(L.load().hamt_attn_small_bwd(C.byref(d), _p(q), _p(k), _p(v), _p(mask2), _p(ctx16), _p(dctx16), _p(lse), None,
                                                 _p(dq), _p(dk), _p(dv), _p(rng_state(dev)), _stream()), "hamt_attn_small_bwd")
        dx, dws, dbs = _proj_bwd(dqkv16, M, x16, (wq, wk, wv), (bq, bk, bv), dx_accum_into=dz)
        return (dx.view(dy.shape), None, None, None, None, None, dws[0], dbs[0], dws[1], dbs[1], dws[2], dbs[2], dwo, dbo, dgamma, dbeta, None)


# ================================================================================================= cross attention block
class KvProjFn(torch.autograd.Function):
    """kv16 [Mkp, 2H] bf16 = c [W_k; W_v]^T + [b_k; b_v] -- the context side of a cross attention as its own node, so that a
    context that does not change between calls (the step-invariant text stream of the `no_lang_ca` rollout, vilmodel_cmt.py:
    645-652 / 701-709) is projected ONCE and shared by every step's attention: autograd sums the steps' dkv16 and this node
    runs one dgrad + one queued wgrad for the whole rollout."""

    @staticmethod
    def forward(ctx, c, wk, bk, wv, bv):
        H = c.shape[-1]
        Mk = c.numel() // H              # [B, Sk, H], or a packed context [M, H] (SelfAttnBlockFn: `seq`)
        c2 = c.reshape(Mk, H)
        if not c2.is_contiguous():
            c2 = c2.contiguous()
        c16 = _x16_of(c, c2)
        kv16 = torch.empty(c16.shape[0], 2 * H, dtype=torch.bfloat16, device=c.device)
        _proj(c16, Mk, (wk, wv), (bk, bv), kv16)
        ctx.save_for_backward(c16, wk, bk, wv, bv)
        ctx.meta = (Mk, tuple(c.shape))
        return kv16

    @staticmethod
    def backward(ctx, dkv16):
        if dkv16 is None:
            return (None,) * 5
        c16, wk, bk, wv, bv = ctx.saved_tensors
        Mk, shape = ctx.meta
        if not dkv16.is_contiguous():
            dkv16 = dkv16.contiguous()
        dc, dwkv, dbkv = _proj_bwd(dkv16, Mk, c16, (wk, wv), (bk, bv), need_dx=ctx.needs_input_grad[0])
        return (dc.view(shape) if dc is not None else None, dwkv[0], dbkv[0], dwkv[1], dbkv[1])


class CrossAttnBlockFn(torch.autograd.Function):
    """`seq_q` / `seq_k` (at most one; SelfAttnBlockFn's `seq` = (cu_seqlens, sequences, S_max)): the queries x [M, H], or the context
    behind kv16, are a PACKED batch (the instruction tokens of a ragged batch in the x-layers) -- hamt_attn_varlen_cross_*.  `pairs` =
    the number of samples (= the fixed-stride side's batch): packed query sequences behind it are fillers without keys.  `pair` (int32
    device tensor, optional) names each query sequence's key side instead (hamt.h: packed copies of a batch, fillers between them)."""

    @staticmethod
    def forward(ctx, x, kv16, Sk, add_mask, heads, p_attn, p_hidden, eps, wq, bq, wo, bo, gamma, beta, seq_q=None, seq_k=None, pairs=0, pair=None):
        if seq_q is not None:
            cu, B, Sq = seq_q
            Mq, H = x.shape
            Mk = pairs * Sk
        else:
            B, Sq, H = x.shape
            Mq, Mk = B * Sq, B * Sk
            if seq_k is not None:
                assert add_mask is None and (pairs == B or pair is not None)
                Sk = seq_k[2]
                Mk = -1                  # (a packed context: its row count is kv16's business)
        dev = x.device
        x2 = x.reshape(Mq, H)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        x16 = _x16_of(x, x2)
        Mqp = x16.shape[0]
        q16 = torch.empty(Mqp, H, dtype=torch.bfloat16, device=dev)
        gemm(x16[:Mq], weight_operand(wq, "bf16"), q16[:Mq], bias=bq.detach())
        mask2 = add_mask.reshape(-1, Sk).to(torch.float32).contiguous() if add_mask is not None else None
        ctx16 = _zeros_or_empty(Mqp, Mq, H, dev)
        lse = torch.empty(B * heads * Sq, dtype=torch.float32, device=dev)
        cid = next_call_id()
        d = _attn_desc(B, heads, Sq, Sk, H, H, 2 * H, 2 * H, p_attn, cid)
        cu_q = seq_q[0] if seq_q is not None else None
        cu_k = seq_k[0] if seq_k is not None else None
        if cu_q is not None or cu_k is not None:
            L.check(L.load().hamt_attn_varlen_cross_fwd(C.byref(d), _p(q16), _p(kv16[:, :H]), _p(kv16[:, H:]), _p(cu_q), _p(cu_k), pairs, _p(pair),
                                                        _p(mask2), _p(ctx16), _p(lse), _p(rng_state(dev)), _stream()),
                    "hamt_attn_varlen_cross_fwd")
        else:
            L.check(L.load().hamt_attn_small_fwd(C.byref(d), _p(q16), _p(kv16[:, :H]), _p(kv16[:, H:]), _p(mask2), _p(ctx16), _p(lse),
                                                 _p(rng_state(dev)), _stream()), "hamt_attn_small_fwd")
        o = torch.empty(Mq, H, dtype=O_DTYPE, device=dev)
        gemm(ctx16[:Mq], weight_operand(wo, "bf16"), o, bias=bo.detach())
        y, y16, z, mean, rstd, cid_ln = _ln_fwd(o, x2, gamma.detach(), beta.detach(), eps, p_hidden, 0.0, True)
        ctx.save_for_backward(x16, q16, kv16, ctx16, lse, mask2, z, mean, rstd, wq, bq, wo, bo, gamma)
        ctx.ln_params = (gamma, beta, bo)
        ctx.meta = (B, Sq, Sk, H, heads, float(p_attn), float(p_hidden), float(eps), cid, cid_ln, Mq, Mk, pairs)
        ctx.cu_q, ctx.cu_k, ctx.pair = cu_q, cu_k, pair
        ctx.mark_non_differentiable(y16)
        ctx.set_materialize_grads(False)     # else autograd zero-fills a bf16 [Mp,H] "gradient" of y16 per backward
        return y.view(x.shape), y16

    @staticmethod
    def backward(ctx, dy, _unused=None):
        if dy is None:
            return (None,) * 18
        x16, q16, kv16, ctx16, lse, mask2, z, mean, rstd, wq, bq, wo, bo, gamma = ctx.saved_tensors
        B, Sq, Sk, H, heads, p_attn, p_hidden, eps, cid, cid_ln, Mq, Mk, pairs = ctx.meta
        Mqp, Mkp = x16.shape[0], kv16.shape[0]
        dev = dy.device
        dz, _, dx16, dgamma, dbeta, dbo = _ln_bwd(dy.reshape(Mq, H).contiguous(), z, mean, rstd, gamma.detach(), eps, p_hidden, 0.0,
                                                  cid_ln, False, True, True, params=ctx.ln_params)
        dctx16 = torch.empty(Mqp, H, dtype=torch.bfloat16, device=dev)
        gemm(dx16[:Mq], weight_operand(wo, "bf16"), dctx16[:Mq], b_kmajor=True)
        dwo, _ = _wgrad(wo, None, dx16, ctx16, Mq)
        dq16 = _zeros_or_empty(Mqp, Mq, H, dev)
        d = _attn_desc(B, heads, Sq, Sk, H, H, 2 * H, 2 * H, p_attn, cid)
        if ctx.cu_q is not None or ctx.cu_k is not None:
            # packed keys: the filler rows behind the last real sequence belong to no sample -- nobody writes their dK / dV
            dkv16 = torch.zeros(Mkp, 2 * H, dtype=torch.bfloat16, device=dev) if ctx.cu_k is not None else _zeros_or_empty(Mkp, Mk, 2 * H, dev)
            L.check(L.load().hamt_attn_varlen_cross_bwd(C.byref(d), _p(q16), _p(kv16[:, :H]), _p(kv16[:, H:]), _p(ctx.cu_q), _p(ctx.cu_k), pairs, _p(ctx.pair),
                                                        _p(mask2), _p(ctx16), _p(dctx16), _p(lse), _p(dq16), _p(dkv16[:, :H]), _p(dkv16[:, H:]),
                                                        _p(rng_state(dev)), _stream()), "hamt_attn_varlen_cross_bwd")
        else:
            dkv16 = _zeros_or_empty(Mkp, Mk, 2 * H, dev)
            L.check(L.load().hamt_attn_small_bwd(C.byref(d), _p(q16), _p(kv16[:, :H]), _p(kv16[:, H:]), _p(mask2), _p(ctx16), _p(dctx16),
                                                 _p(lse), None, _p(dq16), _p(dkv16[:, :H]), _p(dkv16[:, H:]), _p(rng_state(dev)), _stream()),
                    "hamt_attn_small_bwd")
        dx, dwqs, dbqs = _proj_bwd(dq16, Mq, x16, (wq,), (bq,), dx_accum_into=dz)
        return (dx.view(dy.shape), dkv16 if ctx.needs_input_grad[1] else None, None, None, None, None, None, None,
                dwqs[0], dbqs[0], dwo, dbo, dgamma, dbeta, None, None, None, None)


def _kv_key(c, att):
    a = getattr(att.key.weight, "_hamt_arena16", None)
    return (id(att), c._version, att.key.weight._version, att.value.weight._version, a[2] if a is not None else -1, ops._cache_epoch[0],
            torch.is_grad_enabled())


def precompute_cross_kv(c, att):
    """Project the context `c` [B, Sk, H] with cross-attention module `att`'s key / value weights now and remember the result on
    the tensor (`c._hamt_xkv`): later cross_attn_block calls with this very context (and unchanged weights) skip the projection.
    Used by NavCMT's `language` mode for the step-invariant text stream of a `no_lang_ca` rollout."""
    kv16 = KvProjFn.apply(c, att.key.weight, att.key.bias, att.value.weight, att.value.bias)
    c._hamt_xkv = (_kv_key(c, att), kv16)
    return kv16


# ================================================================================================= feed-forward block
class FfnBlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p_hidden, eps, w1, b1, w2, b2, gamma, beta, grad_on=True):
        shp = x.shape
        H = shp[-1]
        x2 = x.reshape(-1, H)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        M = x2.shape[0]
        dev = x.device
        I = w1.shape[0]
        x16 = _x16_of(x, x2)
        Mp = x16.shape[0]
        g16 = _zeros_or_empty(Mp, M, I, dev)
        if grad_on and any(ctx.needs_input_grad):
            pre = torch.empty(M, I, dtype=GELUP_DTYPE, device=dev)         # gelu'(W1 x + b1), from the same erf/exp as gelu
            gemm(x16[:M], weight_operand(w1, "bf16"), g16[:M], bias=b1.detach(), epilogue=L.EPI_GELU_GRAD, aux=pre)
        else:        # inference (rollout, validation): nobody reads gelu' -- one M x 3072 image less to write
            pre = None
            gemm(x16[:M], weight_operand(w1, "bf16"), g16[:M], bias=b1.detach(), epilogue=L.EPI_GELU)
        o = torch.empty(M, H, dtype=O_DTYPE, device=dev)      # bf16: what a linear returns under autocast (ops._ln_fwd reads it as such)
        gemm(g16[:M], weight_operand(w2, "bf16"), o, bias=b2.detach())
        y, y16, z, mean, rstd, cid_ln = _ln_fwd(o, x2, gamma.detach(), beta.detach(), eps, p_hidden, 0.0, True)
        ctx.save_for_backward(x16, g16, pre, z, mean, rstd, w1, b1, w2, b2, gamma)
        ctx.ln_params = (gamma, beta, b2)
        ctx.meta = (shp, M, H, I, float(p_hidden), float(eps), cid_ln)
        ctx.mark_non_differentiable(y16)
        ctx.set_materialize_grads(False)     # else autograd zero-fills a bf16 [Mp,H] "gradient" of y16 per backward
        return y.view(shp), y16

    @staticmethod
    def backward(ctx, dy, _unused=None):
        if dy is None:
            return (None,) * 10
        x16, g16, pre, z, mean, rstd, w1, b1, w2, b2, gamma = ctx.saved_tensors
        shp, M, H, I, p_hidden, eps, cid_ln = ctx.meta
        dev = dy.device
        Mp = x16.shape[0]
        dz, _, dx16, dgamma, dbeta, db2 = _ln_bwd(dy.reshape(M, H).contiguous(), z, mean, rstd, gamma.detach(), eps, p_hidden, 0.0,
                                                  cid_ln, False, True, True, params=ctx.ln_params)
        dh16 = _zeros_or_empty(Mp, M, I, dev)
        gemm(dx16[:M], weight_operand(w2, "bf16"), dh16[:M], b_kmajor=True, epilogue=L.EPI_MUL_AUX, aux=pre)     # dG * gelu'(pre)
        dw2, _ = _wgrad(w2, None, dx16, g16, M)
        gemm(dh16[:M], weight_operand(w1, "bf16"), dz, b_kmajor=True, epilogue=L.EPI_ACCUM)                     # dx = dz + dH W1
        dw1, db1 = _wgrad(w1, b1, dh16, x16, M)
        return dz.view(shp), None, None, dw1, db1, dw2, db2, dgamma, dbeta, None


# ================================================================================================= module-facing wrappers
def _tag(y, y16):
    y._hamt_bf16 = (y16, y._version, y.data_ptr())
    return y


def usable(prec: str, x: torch.Tensor) -> bool:
    return ENABLED and prec == "bf16" and x.is_cuda and x.shape[-1] % 64 == 0


def self_attn_block(x, add_mask, att_self, att_out, training):
    pa = float(att_self.dropout.p) if training else 0.0
    ph = float(att_out.dropout.p) if training else 0.0
    seq = getattr(x, "_hamt_seq", None)      # a packed batch (model.vilmodel.NavPreTrainedModel._text): see SelfAttnBlockFn
    y, y16 = SelfAttnBlockFn.apply(x, None if seq is not None else add_mask, att_self.num_attention_heads, pa, ph, att_out.LayerNorm.eps,
                                   att_self.query.weight, att_self.query.bias, att_self.key.weight, att_self.key.bias,
                                   att_self.value.weight, att_self.value.bias, att_out.dense.weight, att_out.dense.bias,
                                   att_out.LayerNorm.weight, att_out.LayerNorm.bias, seq)
    y = _tag(y, y16)
    if seq is not None:
        y._hamt_seq = seq
        if getattr(x, "_hamt_pair", None) is not None:
            y._hamt_pair = x._hamt_pair
    return y


def cross_attn_block(x, c, add_mask, att, att_out, training):
    pa = float(att.dropout.p) if training else 0.0
    ph = float(att_out.dropout.p) if training else 0.0
    cached = getattr(c, "_hamt_xkv", None)
    if cached is not None and cached[0] == _kv_key(c, att):
        kv16 = cached[1]
    else:
        kv16 = KvProjFn.apply(c, att.key.weight, att.key.bias, att.value.weight, att.value.bias)
    seq_q, seq_k = getattr(x, "_hamt_seq", None), getattr(c, "_hamt_seq", None)     # one side packed: see CrossAttnBlockFn
    pairs = c.shape[0] if seq_q is not None else (x.shape[0] if seq_k is not None else 0)
    # packed COPIES of a batch (forward_itm): (key sample of each packed sequence, packed sequence of each fixed-stride sample)
    pm = getattr(x if seq_q is not None else c, "_hamt_pair", None) if (seq_q is not None or seq_k is not None) else None
    pair = None if pm is None else (pm[0] if seq_q is not None else pm[1])
    y, y16 = CrossAttnBlockFn.apply(x, kv16, seq_k[2] if seq_k is not None else c.shape[1], None if seq_k is not None else add_mask,
                                    att.num_attention_heads, pa, ph, att_out.LayerNorm.eps,
                                    att.query.weight, att.query.bias, att_out.dense.weight, att_out.dense.bias,
                                    att_out.LayerNorm.weight, att_out.LayerNorm.bias, seq_q, seq_k, pairs, pair)
    y = _tag(y, y16)
    if seq_q is not None:
        y._hamt_seq = seq_q
        if pm is not None:
            y._hamt_pair = pm
    return y


def ffn_block(x, inter, out, training):
    ph = float(out.dropout.p) if training else 0.0
    y, y16 = FfnBlockFn.apply(x, ph, out.LayerNorm.eps, inter.dense.weight, inter.dense.bias, out.dense.weight, out.dense.bias,
                              out.LayerNorm.weight, out.LayerNorm.bias, torch.is_grad_enabled())
    y = _tag(y, y16)
    seq = getattr(x, "_hamt_seq", None)
    if seq is not None:
        y._hamt_seq = seq
        if getattr(x, "_hamt_pair", None) is not None:
            y._hamt_pair = x._hamt_pair
    return y
