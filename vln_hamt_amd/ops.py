"""autograd Functions over the C-ABI of libhamt_hip.so.

Every Function's forward and backward are HIP kernel launches on torch's *current* stream (raw
pointers + hipStream_t through ctypes); torch only owns the memory and the autograd graph.
There is no CPU or eager-PyTorch fallback: tensors must live on an AMD GPU and the library must load.

Precision: ``prec`` is ``"bf16"`` (bf16 MFMA operands, fp32 accumulate/epilogue; activations stay
fp32 in HBM and are rounded while staged) or ``"fp32"`` (exact fp32 MFMA).
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Optional

import torch

from . import _lib as L

# ------------------------------------------------------------------------------------------ plumbing
_rng_state = {}
_call_counter = [0]


def rng_state(device) -> torch.Tensor:
    """Per-device dropout RNG words [seed, epoch] (uint64 stored as int64)."""
    d = torch.device(device)
    key = d.index
    if key is None:                      # "cuda" without an index means the CURRENT device (cuda:LOCAL_RANK), not cuda:0
        key = torch.cuda.current_device() if d.type == "cuda" else 0
    t = _rng_state.get(key)
    if t is None:
        t = torch.tensor([0x243F6A8885A308D3 & 0x7FFFFFFFFFFFFFFF, 0], dtype=torch.int64,
                         device=torch.device(d.type, key) if d.type == "cuda" else d)
        _rng_state[key] = t
    return t


def manual_seed(seed: int, device="cuda"):
    rng_state(device).copy_(torch.tensor([seed & 0x7FFFFFFFFFFFFFFF, 0], dtype=torch.int64))
    _call_counter[0] = 0


def advance_rng_epoch(device="cuda"):
    """New dropout epoch (one kernel; graph-capturable).  Call once per optimisation step."""
    st = rng_state(device)
    L.check(L.load().hamt_rng_advance(_p(st), _stream()), "hamt_rng_advance")


def next_call_id() -> int:
    _call_counter[0] = (_call_counter[0] + 1) & 0xFFFFFFFF
    return _call_counter[0]


def _p(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(t: torch.Tensor, name: str):
    if not t.is_cuda:
        raise L.HamtError(f"{name}: tensor is on {t.device}; the HAMT kernels run on the GPU only (no CPU fallback)")


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return L.HAMT_F32
    if t.dtype == torch.bfloat16:
        return L.HAMT_BF16
    if t.dtype == torch.float16:     # IEEE half: `C` of a dense layer in front of a LayerNorm (hamt.h: HAMT_F16) -- hamt_gemm refuses it elsewhere
        return L.HAMT_F16
    if t.dtype == torch.uint8:       # the one-byte gelu' image (hamt.h: HAMT_U8G) -- `aux` of EPI_GELU_GRAD / EPI_MUL_AUX only
        return L.HAMT_U8G
    raise L.HamtError(f"unsupported dtype {t.dtype}")


def _prec(prec: str) -> int:
    return L.PREC_F32 if prec == "fp32" else L.PREC_BF16


def _ld(t: torch.Tensor) -> int:
    """Leading dimension (elements) of a 2-D row-major view whose last dim is contiguous."""
    assert t.dim() == 2 and (t.stride(1) == 1 or t.shape[1] == 1), (t.shape, t.stride())
    return t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])


def _pad_cols(n: int, elem_size: int) -> int:
    q = 16 // elem_size
    return (n + q - 1) // q * q


def empty_rows(M: int, N: int, device, dtype=torch.float32) -> torch.Tensor:
    """[M,N] view whose rows start 16-byte aligned (row stride padded), e.g. the 30522-wide MLM logits."""
    return torch.empty(M, _pad_cols(N, torch.empty((), dtype=dtype).element_size()), dtype=dtype, device=device)[:, :N]


def _operand(t: torch.Tensor) -> torch.Tensor:
    """Return `t` if it satisfies the GEMM operand contract (unit inner stride, 16-byte aligned rows and
    base); otherwise a padded copy (only skinny / odd-width tensors such as [M,1] head gradients)."""
    es = t.element_size()
    ok = (t.stride(1) == 1 or t.shape[1] == 1) and t.data_ptr() % 16 == 0 and (t.shape[0] <= 1 or (t.stride(0) * es) % 16 == 0)
    if ok and not (t.shape[0] <= 1 and (t.shape[1] * es) % 16):
        return t
    o = empty_rows(t.shape[0], t.shape[1], t.device, t.dtype)
    o.copy_(t)
    return o


# ------------------------------------------------------------------------------------------ raw GEMM
def gemm(a, b, out, *, a_kmajor=False, b_kmajor=False, bias=None, epilogue=0, aux=None, prec="bf16", alpha=1.0,
         k_red=None, drop=None, k_valid=None):
    """out[M,N] = epi(alpha * op(a) @ op(b)); 2-D views with unit inner stride (strided rows allowed).
    drop = (p, call_id): HAMT_EPI_DROPOUT on the epilogue value (before the residual add).
    k_valid (K-strided operands: the weight-gradient form): only the first k_valid reduction rows count, the rest is padding of any
    content (the kernels re-read the last valid row instead and zero its products)."""
    _chk(a, "gemm")
    a, b = _operand(a), _operand(b)
    M, N = out.shape
    # reduction length: K-contiguous operands define it; K-strided operands may store fewer rows (ka/kb_rows)
    ka = a.shape[0] if a_kmajor else a.shape[1]
    kb = b.shape[0] if b_kmajor else b.shape[1]
    K = k_red if k_red is not None else max(ka, kb)
    assert (a.shape[1] == M if a_kmajor else a.shape == (M, K)), (a.shape, out.shape, K)
    assert (b.shape[1] == N if b_kmajor else b.shape == (N, K)), (b.shape, out.shape, K)
    assert ka <= K and kb <= K and (a_kmajor or ka == K) and (b_kmajor or kb == K)
    if _prec(prec) == L.PREC_F32:
        assert a.dtype == torch.float32 and b.dtype == torch.float32
    d = L.GemmDesc(M, N, K, _ld(a), _ld(b), _ld(out), _ld(aux) if aux is not None else 0, int(a_kmajor), int(b_kmajor),
                   _dt(a), _dt(b), _dt(out), _dt(aux) if aux is not None else 0, _prec(prec),
                   epilogue | (L.EPI_BIAS if bias is not None else 0), alpha, ka if ka < K else 0, kb if kb < K else 0, 0.0, 0, None)
    if k_valid is not None and k_valid < K:
        assert a_kmajor and b_kmajor and 0 < k_valid
        d.ka_rows, d.kb_rows = min(ka, k_valid), min(kb, k_valid)
    if drop is not None and drop[0] > 0.0:
        d.epilogue |= L.EPI_DROPOUT
        d.p_drop, d.call_id, d.rng = float(drop[0]), int(drop[1]), rng_state(out.device).data_ptr()
    lib = L.load()
    ks = 1
    if a.dtype == torch.bfloat16 and b.dtype == torch.bfloat16:
        ks = lib.hamt_gemm_ksplit(C.byref(d))
    if ks > 1:      # deterministic split-K: fp32 partial tiles in a scratch buffer, summed in slice order
        ws = torch.empty(ks * M * N, dtype=torch.float32, device=out.device)
        L.check(lib.hamt_gemm_ws(C.byref(d), _p(a), _p(b), _p(out), _p(bias), _p(aux), _p(ws), ws.numel() * 4, _stream()), "hamt_gemm_ws")
    else:
        L.check(lib.hamt_gemm(C.byref(d), _p(a), _p(b), _p(out), _p(bias), _p(aux), _stream()), "hamt_gemm")
    return out


def colsum(x2d: torch.Tensor, out: Optional[torch.Tensor] = None, accumulate=False) -> torch.Tensor:
    M, N = x2d.shape
    if out is None:
        out = torch.empty(N, dtype=torch.float32, device=x2d.device)
    ws = torch.empty(L.workspace_bytes(L.WS_COLSUM, M, N) // 4, dtype=torch.float32, device=x2d.device)
    L.check(L.load().hamt_colsum(M, N, _p(x2d), _ld(x2d), _dt(x2d), _p(out), int(accumulate), _p(ws), _stream()), "hamt_colsum")
    return out


def cast_bf16(x: torch.Tensor) -> torch.Tensor:
    x = x.contiguous()
    y = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    L.check(L.load().hamt_cast_f32_bf16(x.numel(), _p(x), _p(y), _stream()), "hamt_cast_f32_bf16")
    return y


def arena16_valid(w: torch.Tensor, a) -> bool:
    """Is the optimizer's bf16 shadow `a` (= w._hamt_arena16) still the image of `w`?  The arena's version counter moves
    with the optimizer's own updates; an in-place write to the parameter itself (load_state_dict, p.data.copy_, re-init:
    `p.data` is a view of the arena with its OWN version counter) moves only the parameter's, so both are compared."""
    return a[1]._version == a[2] and w._version == a[3] and a[1].data_ptr() <= w.data_ptr() < a[1].data_ptr() + a[1].numel() * 4


def weight_operand(w: torch.Tensor, prec: str) -> torch.Tensor:
    """GEMM B-operand for a parameter: fp32 master in fp32 mode, cached bf16 shadow in bf16 mode.
    The shadow is refreshed when the parameter's version counter or storage changes."""
    wd = w.detach()
    if prec == "fp32":
        return wd
    a = getattr(w, "_hamt_arena16", None)       # (bf16 view, fp32 arena, arena version, parameter version at last sync): optim.AdamW
    if a is not None and arena16_valid(w, a):
        return a[0]
    c = getattr(w, "_hamt_w16", None)
    key = (w._version, _cache_epoch[0])
    if c is None or c[0] != key or c[1] != w.data_ptr():
        c = (key, w.data_ptr(), cast_bf16(wd)) + _cache_stamp(w)
        try:
            w._hamt_w16 = c
        except Exception:
            pass
    _cache_sync(c)
    return c[2]


ACT_NONE, ACT_GELU, ACT_RELU = 0, 1, 2


# ---- operand preparation for the bf16 fast GEMM (both operands bf16, K-contiguous, K % 64 == 0)
def _rup(n: int, m: int = 64) -> int:
    return (n + m - 1) // m * m


def cast_pad16(x2: torch.Tensor, cpad: Optional[int] = None, rpad: Optional[int] = None) -> torch.Tensor:
    """fp32 [R,C] (strided rows ok) -> bf16 [Rpad,Cpad], padding rows/columns zero-filled (defaults: multiples of 64)."""
    R, Cc = x2.shape
    cpad = _rup(Cc) if cpad is None else cpad
    rpad = _rup(R) if rpad is None else rpad
    if x2.stride(1) != 1:
        x2 = x2.contiguous()
    y = torch.empty(rpad, cpad, dtype=torch.bfloat16, device=x2.device)
    L.check(L.load().hamt_cast_pad_bf16(R, Cc, rpad, cpad, _p(x2), _ld(x2), _p(y), cpad, _stream()), "hamt_cast_pad_bf16")
    return y


def cast_pad16_dropout(x2: torch.Tensor, p: float, call_id: int) -> torch.Tensor:
    """bf16 [Rpad,C] image of x2 * keep-mask of the HAMT_EPI_DROPOUT GEMM with the same call id (C % 8 == 0)."""
    R, Cc = x2.shape
    if x2.stride(1) != 1:
        x2 = x2.contiguous()
    y = torch.empty(_rup(R), Cc, dtype=torch.bfloat16, device=x2.device)
    L.check(L.load().hamt_cast_pad_bf16_dropout(R, Cc, y.shape[0], _p(x2), _ld(x2), _p(y), Cc, float(p), int(call_id),
                                                _p(rng_state(x2.device)), _stream()), "hamt_cast_pad_bf16_dropout")
    return y


def cast_t16(x2: torch.Tensor, rpad: Optional[int] = None) -> torch.Tensor:
    """[R,C] fp32/bf16 (strided rows ok) -> bf16 [C,Rpad] = x^T, zero-filled padding columns."""
    R, Cc = x2.shape
    rpad = _rup(R) if rpad is None else rpad
    if x2.stride(1) != 1:
        x2 = x2.contiguous()
    y = torch.empty(Cc, rpad, dtype=torch.bfloat16, device=x2.device)
    L.check(L.load().hamt_cast_transpose(R, Cc, _p(x2), _ld(x2), _dt(x2), _p(y), rpad, rpad, _stream()), "hamt_cast_transpose")
    return y


_cache_epoch = [0]


def _cache_stamp(w):
    """(event, stream handle) of the stream that just built a lazily cached weight image: another stream (the model runs
    independent branches on two, streams.py) waits for the event before its first read."""
    if not w.is_cuda:
        return (None, None)
    st = torch.cuda.current_stream()
    ev = torch.cuda.Event()
    ev.record(st)
    return (ev, st.cuda_stream)


def _cache_sync(c):
    if c[3] is not None:
        st = torch.cuda.current_stream()
        if st.cuda_stream != c[4]:
            st.wait_event(c[3])


def invalidate_weight_caches():
    """Force every lazily cached weight shadow to be rebuilt at its next use (call right before capturing a graph so
    that the rebuild kernels are part of the captured step)."""
    _cache_epoch[0] += 1


def weight_t16(w: torch.Tensor) -> torch.Tensor:
    """cached bf16 W^T [K, Npad64] of a [N,K] parameter (B operand of the dgrad dX = dY W in NT form)."""
    c = getattr(w, "_hamt_wt16", None)
    key = (w._version, _cache_epoch[0])
    if c is None or c[0] != key or c[1] != w.data_ptr():
        c = (key, w.data_ptr(), cast_t16(w.detach())) + _cache_stamp(w)
        try:
            w._hamt_wt16 = c
        except Exception:
            pass
    _cache_sync(c)
    return c[2]


def _fast_ok(K: int) -> bool:
    return K % 64 == 0 and K >= 64


def prep_x16(x2: torch.Tensor, prec: str):
    """bf16 operand image of the activations [M,K] for the fast GEMMs: rows padded to a multiple of 64 with zeros so
    the same buffer serves as A of the forward (first M rows) and as the K-strided B of the weight gradient."""
    if prec != "bf16" or not _fast_ok(x2.shape[1]):
        return None
    if x2.dtype == torch.bfloat16 and x2.shape[0] % 64 == 0 and x2.is_contiguous():
        return x2
    assert x2.dtype == torch.float32
    return cast_pad16(x2, x2.shape[1])


def _linear_fwd(x2, weight, bias, out, act, prec, pre=None, x16=None):
    """out = act(x2 @ W^T + b).  bf16 mode with K % 64 == 0: x is cast once (x16 may be passed in to share it) and the
    GEMM takes the glds fast path; otherwise the generic kernel converts while staging."""
    epi = 0
    if act == ACT_GELU:
        epi |= L.EPI_GELU | (L.EPI_SAVE_PRE if pre is not None else 0)
    elif act == ACT_RELU:
        epi |= L.EPI_RELU
    a = x16[:x2.shape[0]] if x16 is not None else x2
    gemm(a, weight_operand(weight, prec), out, bias=bias, epilogue=epi, aux=pre if act == ACT_GELU else None, prec=prec)


def _linear_bwd(dy2, x2, x16, weight, prec, need_dx, need_dw, need_db, dx_out=None, dx_accumulate=False, dy16=None,
                bias_param=False):
    """dy2 [M,N] (strided rows ok), x [M,K] (x16 = its padded bf16 image or None), weight [N,K] parameter
    -> (dx [M,K], dW [N,K], db [N]); `bias_param` = the bias parameter (or None) enables queueing dW/db (wgrad.py), in
    which case dW and db come back as None.  bf16 mode: ONE padded bf16 image of dY feeds both contractions of the fast kernel:
    dX = dY16[M,Np] * W16 (W k-strided, "NN"), dW = dY16^T * X16 (both k-strided, "TN") -- no transposed copies."""
    M, N = dy2.shape
    K = weight.shape[1]
    dx = dw = db = None
    fast = prec == "bf16" and x16 is not None and N >= 8
    if fast and dy16 is None:
        dy16 = cast_pad16(dy2)                                   # [Mp, Np], zero padded
    if need_dx:
        dx = dx_out if dx_out is not None else torch.empty(M, K, dtype=torch.float32, device=dy2.device)
        epi = L.EPI_ACCUM if dx_accumulate else 0
        if fast:
            gemm(dy16[:M], weight_operand(weight, prec), dx, b_kmajor=True, epilogue=epi, prec=prec, k_red=dy16.shape[1])
        else:
            gemm(dy2, weight_operand(weight, prec), dx, b_kmajor=True, epilogue=epi, prec=prec)
    if need_dw and fast and bias_param is not False:
        from . import wgrad
        if wgrad.eligible(weight, dy16[:, :N], x16):    # queued for the grouped end-of-pass launch (wgrad.py); db rides along
            wgrad.defer(weight, bias_param if need_db else None, dy16[:, :N], x16)
            return dx, None, None
    slot_w = _accum_slot(weight) if (need_dw and not (K <= 8)) else None
    if slot_w is not None:
        # a parameter outside the grouped launch (the fp32 prediction heads) that owns a zero-at-the-start-of-a-step slot in the
        # optimizer's gradient arena: ADD dW (and db) there -- no autograd tensor for optim.AdamW._pack_grads to copy in afterwards
        from . import wgrad
        wgrad.queue(dy2.device).current()
        if fast:
            gemm(dy16[:, :N], x16, slot_w, a_kmajor=True, b_kmajor=True, prec=prec, epilogue=L.EPI_ACCUM)
        else:
            gemm(dy2, x2 if x2 is not None else x16[:M], slot_w, a_kmajor=True, b_kmajor=True, prec=prec, epilogue=L.EPI_ACCUM)
        wgrad.publish_slot_grad(weight, slot_w)
        need_dw = False
        slot_b = _accum_slot(bias_param) if (need_db and torch.is_tensor(bias_param)) else None
        if slot_b is not None:
            colsum(dy2, out=slot_b, accumulate=True)
            wgrad.publish_slot_grad(bias_param, slot_b)
            need_db = False
    if need_dw:
        dw = torch.empty(N, K, dtype=torch.float32, device=dy2.device)
        if fast:
            gemm(dy16[:, :N], x16, dw, a_kmajor=True, b_kmajor=True, prec=prec)
        elif K <= 8 and x2 is not None and dy2.dtype == torch.float32 and x2.dtype == torch.float32:
            ws = torch.empty(64 * N * K, dtype=torch.float32, device=dy2.device)
            L.check(L.load().hamt_smallk_wgrad(M, N, K, _p(dy2), _ld(dy2), _p(x2), _ld(x2), _p(dw), 0, _p(ws), _stream()), "hamt_smallk_wgrad")
        else:
            gemm(dy2, x2 if x2 is not None else x16[:M], dw, a_kmajor=True, b_kmajor=True, prec=prec)
    if need_db:
        db = colsum(dy2)
    return dx, dw, db


SLOT_ACCUM = os.environ.get("HAMT_NO_SLOT_ACCUM") is None


def _accum_slot(p):
    """p's slot in the optimizer's flat gradient arena if a producer may ADD into it during this backward pass (a leaf parameter
    whose slot the update kernel leaves zero, inside a pass, .grad unset or already the slot), else None."""
    if p is None or not torch.is_tensor(p) or not p.is_leaf or not p.requires_grad or not SLOT_ACCUM:
        return None
    from . import wgrad
    slot = getattr(p, "_hamt_grad_slot", None)
    if (slot is None or not wgrad.ENABLED or torch._C._current_graph_task_id() < 0 or slot.shape != p.shape or not slot.is_contiguous()
            or not getattr(p, "_hamt_slot_zeroed", False) or not (p.grad is None or p.grad.data_ptr() == slot.data_ptr())):
        return None
    return slot


# ------------------------------------------------------------------------------------------ Linear
class LinearFn(torch.autograd.Function):
    """y = act(x W^T + b)  (nn.Linear + optional erf-GELU / ReLU; vilmodel.py:140, 168-171, 182, 263-264)."""

    @staticmethod
    def forward(ctx, x, weight, bias, act, prec, residual=None):
        _chk(x, "LinearFn")
        K = x.shape[-1]
        x2 = x.reshape(-1, K)
        if x2.stride(-1) != 1:
            x2 = x2.contiguous()
        N = weight.shape[0]
        M = x2.shape[0]
        x16 = prep_x16(x2, prec)
        # wide odd-width outputs (the 30522-column MLM logits) get a padded row stride so that they can feed the
        # backward GEMMs as 16-byte aligned operands; everything else is plain contiguous
        y = empty_rows(M, N, x.device) if (N % 4 and N >= 256) else torch.empty(M, N, dtype=torch.float32, device=x.device)
        pre = torch.empty(M, N, dtype=torch.float32, device=x.device) if act == ACT_GELU else None
        if residual is not None:                 # y = x W^T + b + residual in the GEMM epilogue (pre-LN residual add)
            assert act == ACT_NONE and residual.numel() == M * N
            r2 = residual.reshape(M, N)
            r2 = r2 if r2.is_contiguous() else r2.contiguous()
            a = x16[:M] if x16 is not None else x2
            gemm(a, weight_operand(weight, prec), y, bias=bias.detach() if bias is not None else None, epilogue=L.EPI_ADD_AUX,
                 aux=r2.detach(), prec=prec)
        else:
            _linear_fwd(x2, weight, bias.detach() if bias is not None else None, y, act, prec, pre, x16)
        # the bf16 image is all backward needs of x (weight gradient operand); keep fp32 x only on the generic path
        ctx.save_for_backward(x2 if x16 is None else None, x16, weight, pre if act == ACT_GELU else (y if act == ACT_RELU else None))
        ctx.act, ctx.prec, ctx.has_bias, ctx.xshape, ctx.M = act, prec, bias is not None, x.shape, M
        ctx.bias_param = bias if (bias is not None and bias.is_leaf) else (None if bias is None else False)
        ctx.res_shape = residual.shape if residual is not None else None
        return y if x.dim() == 2 else y.reshape(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, x16, weight, h = ctx.saved_tensors
        N = weight.shape[0]
        dy2 = dy.reshape(-1, N)
        if ctx.act != ACT_NONE:
            dy2 = dy2.contiguous()
            h = h.contiguous()
            dh = torch.empty_like(dy2)
            L.check(L.load().hamt_act_bwd(dy2.numel(), _p(dy2), _p(h), ctx.act, _p(dh), _stream()), "hamt_act_bwd")
            dy2 = dh
        dy2 = _operand(dy2)
        dx, dw, db = _linear_bwd(dy2, x2, x16, weight, ctx.prec, ctx.needs_input_grad[0], ctx.needs_input_grad[1],
                                 ctx.has_bias and ctx.needs_input_grad[2], bias_param=ctx.bias_param)
        dres = dy.reshape(ctx.res_shape) if (ctx.res_shape is not None and ctx.needs_input_grad[5]) else None
        return (dx.view(ctx.xshape) if dx is not None else None), dw, db, None, None, dres


def linear(x, weight, bias, act=ACT_NONE, prec="bf16", residual=None):
    """act(x W^T + b) (+ residual: added in the GEMM epilogue, act must be ACT_NONE)"""
    return LinearFn.apply(x, weight, bias, act, prec, residual)


class PackedLinearFn(torch.autograd.Function):
    """Several nn.Linear layers applied to the same input, written side by side into ONE packed buffer
    [M, sum(N_i)] (query/key/value of vilmodel.py:97-99 -> qkv; key/value of :324-325 -> kv) so the
    attention kernel reads heads by pointer arithmetic and no permute/contiguous copy is needed."""

    @staticmethod
    def forward(ctx, x, prec, *wb):
        _chk(x, "PackedLinearFn")
        K = x.shape[-1]
        x2 = x.reshape(-1, K)
        if x2.stride(-1) != 1:
            x2 = x2.contiguous()
        ws, bs = wb[0::2], wb[1::2]
        ns = [w.shape[0] for w in ws]
        out = torch.empty(x2.shape[0], sum(ns), dtype=torch.float32, device=x.device)
        x16 = prep_x16(x2, prec)
        c = 0
        for w, b, n in zip(ws, bs, ns):
            _linear_fwd(x2, w, b.detach(), out[:, c:c + n], ACT_NONE, prec, x16=x16)
            c += n
        ctx.save_for_backward(x2 if x16 is None else None, x16, *ws)
        ctx.prec, ctx.ns, ctx.xshape, ctx.M = prec, ns, x.shape, x2.shape[0]
        ctx.bias_params = [b if b.is_leaf else False for b in bs]
        return out

    @staticmethod
    def backward(ctx, dout):
        x2, x16, *ws = ctx.saved_tensors
        dout = dout.contiguous()
        grads = []
        M = ctx.M
        K = ws[0].shape[1]
        dx = torch.empty(M, K, dtype=torch.float32, device=dout.device) if ctx.needs_input_grad[0] else None
        fast = ctx.prec == "bf16" and x16 is not None and all(n % 64 == 0 for n in ctx.ns)
        d16 = cast_pad16(dout) if fast else None                  # [Mp, sum n]: column slices are the per-layer dY16
        c = 0
        for i, (w, n) in enumerate(zip(ws, ctx.ns)):
            _, dw, db = _linear_bwd(dout[:, c:c + n], x2, x16, w, ctx.prec, dx is not None, True, True, dx_out=dx,
                                    dx_accumulate=i > 0, dy16=d16[:, c:c + n] if fast else None, bias_param=ctx.bias_params[i])
            grads += [dw, db]
            c += n
        return (dx.view(ctx.xshape) if dx is not None else None), None, *grads


def packed_linear(x, prec, *linears):
    wb = []
    for lin in linears:
        wb += [lin.weight, lin.bias]
    return PackedLinearFn.apply(x, prec, *wb)


# ------------------------------------------------------------------------------------------ attention
class AttnFn(torch.autograd.Function):
    """softmax(Q K^T / sqrt(d) + mask) V with dropout on the probabilities (vilmodel.py:101-126, 327-348).
    `q_src` is either the packed self-attention buffer [B*S, 3H] (kv_src None) or the query projection
    [B*Sq, H] with `kv_src` = packed [B*Sk, 2H].  `add_mask` is the reference's additive (B,1,1,Sk) mask."""

    @staticmethod
    def forward(ctx, q_src, kv_src, add_mask, B, heads, p_drop, prec="fp32"):
        _chk(q_src, "AttnFn")
        packed = kv_src is None
        H = q_src.shape[1] // 3 if packed else q_src.shape[1]
        Sq = q_src.shape[0] // B
        if packed:
            q, k, v, Sk = q_src[:, :H], q_src[:, H:2 * H], q_src[:, 2 * H:], Sq
        else:
            q, k, v, Sk = q_src, kv_src[:, :H], kv_src[:, H:], kv_src.shape[0] // B
        mask2 = None
        if add_mask is not None:
            mask2 = add_mask.reshape(B, Sk).to(torch.float32).contiguous()
        out = torch.empty(B * Sq, H, dtype=torch.float32, device=q_src.device)
        lse = torch.empty(B * heads * Sq, dtype=torch.float32, device=q_src.device)
        cid = next_call_id()
        rng = rng_state(q_src.device)
        d = L.AttnDesc(B, heads, Sq, Sk, H // heads, _ld(q), _ld(k), _ld(v), H, _dt(q), L.HAMT_F32,
                       1.0 / math.sqrt(H // heads), float(p_drop), cid, _prec(prec))
        L.check(L.load().hamt_attn_small_fwd(C.byref(d), _p(q), _p(k), _p(v), _p(mask2), _p(out), _p(lse), _p(rng), _stream()),
                "hamt_attn_small_fwd")
        ctx.save_for_backward(q_src, kv_src, mask2, out, lse)
        ctx.desc_args = (B, heads, Sq, Sk, H, float(p_drop), cid, prec)
        return out

    @staticmethod
    def backward(ctx, dout):
        q_src, kv_src, mask2, out, lse = ctx.saved_tensors
        B, heads, Sq, Sk, H, p_drop, cid, prec = ctx.desc_args
        packed = kv_src is None
        dout = dout.contiguous()
        dq_src = torch.empty_like(q_src)
        dkv_src = None if packed else torch.empty_like(kv_src)
        if packed:
            q, k, v = q_src[:, :H], q_src[:, H:2 * H], q_src[:, 2 * H:]
            dq, dk, dv = dq_src[:, :H], dq_src[:, H:2 * H], dq_src[:, 2 * H:]
        else:
            q, k, v = q_src, kv_src[:, :H], kv_src[:, H:]
            dq, dk, dv = dq_src, dkv_src[:, :H], dkv_src[:, H:]
        d = L.AttnDesc(B, heads, Sq, Sk, H // heads, _ld(q), _ld(k), _ld(v), H, _dt(q), L.HAMT_F32,
                       1.0 / math.sqrt(H // heads), p_drop, cid, _prec(prec))
        L.check(L.load().hamt_attn_small_bwd(C.byref(d), _p(q), _p(k), _p(v), _p(mask2), _p(out), _p(dout), _p(lse), None,
                                             _p(dq), _p(dk), _p(dv), _p(rng_state(q_src.device)), _stream()),
                "hamt_attn_small_bwd")
        return dq_src, dkv_src, None, None, None, None, None


def attention(q_src, kv_src, add_mask, B, heads, p_drop, prec="fp32"):
    return AttnFn.apply(q_src, kv_src, add_mask, B, heads, p_drop, prec)


# ------------------------------------------------------------------------------------------ LayerNorm family
def shadow16(x: torch.Tensor):
    """bf16 image [Mpad64, H] of an activation tensor if its producer emitted one (LayerNorm kernels do), else None."""
    s = getattr(x, "_hamt_bf16", None)
    if s is not None and s[1] == x._version and s[2] == x.data_ptr():
        return s[0]
    return None


LN_Z16 = os.environ.get("HAMT_LN_Z32") is None       # ablation: keep the saved pre-LN sum in fp32 behind a bf16 dense output as well


def _ln_io(t, flag_bf16, flag_f16):
    return flag_bf16 if t.dtype == torch.bfloat16 else (flag_f16 if t.dtype == torch.float16 else 0)


def _ln_fwd(x2, r2, gamma, beta, eps, p_pre, p_post, want16):
    """`x2` may be bf16 (the fused blocks' dense outputs in bf16 mode: what a linear returns under autocast); the pre-LN sum
    saved for backward is then kept in bf16 too (backward re-normalises it with the exact fp32 mean / rstd): 14 instead of
    18 bytes per element through this kernel, 14 instead of 16 through its backward."""
    M, H = x2.shape
    dev = x2.device
    x16_in = x2.dtype in (torch.bfloat16, torch.float16)
    # (the saved pre-LN sum takes the dense output's 2-byte format: half behind a half output -- |z| is O(1 .. 100) --, bf16 behind bf16)
    z = torch.empty(M, H, dtype=x2.dtype if (x16_in and LN_Z16) else torch.float32, device=dev)
    y = torch.empty(M, H, dtype=torch.float32, device=dev)
    mean = torch.empty(M, dtype=torch.float32, device=dev)
    rstd = torch.empty(M, dtype=torch.float32, device=dev)
    Mp = _rup(M) if want16 else 0
    y16 = torch.empty(Mp, H, dtype=torch.bfloat16, device=dev) if want16 else None
    cid = next_call_id() if (p_pre > 0 or p_post > 0) else 0
    d = L.LnDesc(M, H, float(eps), float(p_pre), float(p_post), cid, Mp, _ln_io(x2, L.LN_X_BF16, L.LN_X_F16) | _ln_io(z, L.LN_Z_BF16, L.LN_Z_F16))
    L.check(L.load().hamt_ln_fwd(C.byref(d), _p(x2), _p(r2), _p(gamma), _p(beta), _p(z), _p(y), _p(y16),
                                 _p(mean), _p(rstd), _p(rng_state(dev)), _stream()), "hamt_ln_fwd")
    return y, y16, z, mean, rstd, cid


LN_DEFER = os.environ.get("HAMT_NO_DEFER_LNRED") is None      # ablation: reduce the parameter-gradient partials in line


def _defer_ln_reduce(M, H, ws, red, want_dxsum, params, dev):
    """The partials are in `ws`: queue their summation for the ONE grouped launch at the end of the pass (wgrad.flush) and
    the publication of the results as the parameters' gradients."""
    from . import wgrad
    gamma_p, beta_p, bias_p = params
    wgrad.defer_ln_reduce(ws, red, M, H, want_dxsum, [(gamma_p, red[0]), (beta_p, red[1]), (bias_p, red[2] if want_dxsum else None)])


def _can_defer_ln(params, dev, H=64):
    from . import wgrad
    if not (LN_DEFER and wgrad.ENABLED and params is not None and dev.type == "cuda" and H % 64 == 0):
        return False
    return all(p is None or (isinstance(p, torch.Tensor) and p.is_leaf) for p in params)


def _ln_bwd(dy2, z, mean, rstd, gamma, eps, p_pre, p_post, cid, want_dx32, want_dx16, want_dxsum, params=None, dx16_out=None):
    """-> (dz, dx32 | None, dx16 | None, dgamma, dbeta, dxsum | None).  `params` = (gamma, beta, bias) PARAMETERS (bias:
    of the dense layer whose output fed the LayerNorm, or None): the per-block partials of their gradients are then
    summed by ONE grouped launch at the end of the pass (54 tiny reductions leave the critical path of a step), published
    there as `.grad` (wgrad.defer_ln_reduce) and returned as None here."""
    M, H = dy2.shape
    dev = dy2.device
    dz = torch.empty(M, H, dtype=torch.float32, device=dev)
    dx = torch.empty(M, H, dtype=torch.float32, device=dev) if (want_dx32 and p_pre > 0) else None
    Mp = _rup(M) if want_dx16 else 0
    if dx16_out is not None:      # caller-provided rows of a larger bf16 image
        assert want_dx16 and dx16_out.shape == (Mp, H) and dx16_out.is_contiguous()
        dx16 = dx16_out
    else:
        dx16 = torch.empty(Mp, H, dtype=torch.bfloat16, device=dev) if want_dx16 else None
    red = torch.empty(3, H, dtype=torch.float32, device=dev)      # stored (not accumulated) by the reduce kernel
    ws = torch.empty(L.workspace_bytes(L.WS_LN_BWD, M, H) // 4, dtype=torch.float32, device=dev)
    d = L.LnDesc(M, H, float(eps), float(p_pre), float(p_post), cid, Mp, _ln_io(z, L.LN_Z_BF16, L.LN_Z_F16))
    defer = _can_defer_ln(params, dev, H)
    L.check(L.load().hamt_ln_bwd(C.byref(d), _p(dy2), _p(z), _p(mean), _p(rstd), _p(gamma), _p(dz), _p(dx), _p(dx16),
                                 None if defer else _p(red[0]), None if defer else _p(red[1]),
                                 _p(red[2]) if (want_dxsum and not defer) else None, _p(ws), _p(rng_state(dev)), _stream()),
            "hamt_ln_bwd")
    if defer:
        _defer_ln_reduce(M, H, ws, red, want_dxsum, params, dev)
        return dz, dx, dx16, None, None, None
    return dz, dx, dx16, red[0], red[1], (red[2] if want_dxsum else None)


class LnFn(torch.autograd.Function):
    """y = dropout_post(LayerNorm(dropout_pre(x) + residual)) -- see hamt_ln_fwd in include/hamt.h.
    Second output: the bf16 image of y (rows padded to a multiple of 64) for the next GEMM, or None."""

    @staticmethod
    def forward(ctx, x, residual, gamma, beta, eps, p_pre, p_post, want16):
        _chk(x, "LnFn")
        H = x.shape[-1]
        x2 = x.reshape(-1, H).contiguous()
        r2 = residual.reshape(-1, H).contiguous() if residual is not None else None
        y, y16, z, mean, rstd, cid = _ln_fwd(x2, r2, gamma.detach(), beta.detach(), eps, p_pre, p_post, want16)
        ctx.save_for_backward(z, mean, rstd, gamma)
        ctx.ln_params = (gamma, beta, None)
        ctx.args = (float(eps), float(p_pre), float(p_post), cid, residual is not None, x.shape)
        if y16 is not None:
            ctx.mark_non_differentiable(y16)
        ctx.set_materialize_grads(False)         # no zero-filled stand-in for the bf16 image's (non-existent) gradient
        return y.view(x.shape), y16

    @staticmethod
    def backward(ctx, dy, _d16=None):
        if dy is None:
            return (None,) * 8
        z, mean, rstd, gamma = ctx.saved_tensors
        eps, p_pre, p_post, cid, has_res, xshape = ctx.args
        dy2 = dy.reshape(z.shape).contiguous()
        dz, dx, _, dgamma, dbeta, _ = _ln_bwd(dy2, z, mean, rstd, gamma.detach(), eps, p_pre, p_post, cid, True, False, False,
                                              params=ctx.ln_params)
        gx = (dx if dx is not None else dz).view(xshape)
        return gx, (dz.view(xshape) if has_res else None), dgamma, dbeta, None, None, None, None


def layer_norm(x, residual, ln_module, p_pre=0.0, p_post=0.0, eps=None, want16=False):
    y, y16 = LnFn.apply(x, residual, ln_module.weight, ln_module.bias, ln_module.eps if eps is None else eps, p_pre, p_post, want16)
    if y16 is not None:
        y._hamt_bf16 = (y16, y._version, y.data_ptr())
    return y


# ------------------------------------------------------------------------------------------ two-stream visual embedding
VIS_EMBED = os.environ.get("HAMT_VIS_EMBED", "1") == "1"
# bf16 dense output in front of the fused kernel (-6 us per step at B = 64): OFF -- the history embedder is where ITM's candidates
# differ, and its logits sit at 0.9 of the 1e-2 parity bound already (1.02e-2 on one canon_ragged draw with this on)
VIS_EMBED_X16 = os.environ.get("HAMT_VIS_EMBED_X16", "0") == "1"


SLOT_ACCUM_VE = os.environ.get("HAMT_VIS_EMBED_NO_SLOT") is None


def _grad_dst(p, shape, dev, in_pass):
    """-> (tensor the kernel ADDS p's gradient to, what backward returns for p): p's slot in the optimizer's gradient arena (zero at
    the start of a step; published as `.grad` at the end of the pass, as ops.GatherRowsFn does) or a fresh zero tensor."""
    slot = getattr(p, "_hamt_grad_slot", None) if (in_pass and p is not None and p.is_leaf and p.requires_grad and SLOT_ACCUM_VE) else None
    if (slot is not None and slot.shape == p.shape and slot.is_contiguous() and getattr(p, "_hamt_slot_zeroed", False)
            and (p.grad is None or p.grad.data_ptr() == slot.data_ptr())):
        return slot, None
    t = torch.zeros(shape, dtype=torch.float32, device=dev)
    return t, t


class VisEmbedFn(torch.autograd.Function):
    """e = img_layer_norm(img_linear(img)) + ang_layer_norm(ang_linear(ang))  (vilmodel.py:498-500, 549-551, 557-558): the dense
    layer, then ONE launch for the K = 4 angle projection, both LayerNorms and the sum
    (hamt_vis_embed_fwd; csrc/vis_embed.hip); backward one launch + a small reduction, the image stream's gradient leaving as
    the bf16 image img_linear's queued weight gradient reads.  Second output: the bf16 image of e (or None)."""

    @staticmethod
    def forward(ctx, img, ang, w1, b1, g1, be1, w2, b2, g2, be2, eps1, eps2, prec, want16):
        _chk(img, "VisEmbedFn")
        K, H, A = img.shape[-1], w1.shape[0], ang.shape[-1]
        x2 = img.reshape(-1, K)
        if x2.stride(-1) != 1:
            x2 = x2.contiguous()
        M = x2.shape[0]
        dev = img.device
        a2 = ang.reshape(M, A).to(torch.float32).contiguous()
        x16 = prep_x16(x2, prec)
        x1 = torch.empty(M, H, dtype=torch.bfloat16 if (x16 is not None and VIS_EMBED_X16) else torch.float32, device=dev)
        _linear_fwd(x2, w1, b1.detach() if b1 is not None else None, x1, ACT_NONE, prec, None, x16)
        y = torch.empty(M, H, dtype=torch.float32, device=dev)
        Mp = _rup(M) if want16 else 0
        y16 = torch.empty(Mp, H, dtype=torch.bfloat16, device=dev) if want16 else None
        stats = torch.empty(4, M, dtype=torch.float32, device=dev)
        d = L.VisEmbedDesc(M, H, A, A, float(eps1), float(eps2), int(x1.dtype == torch.bfloat16), Mp)
        L.check(L.load().hamt_vis_embed_fwd(C.byref(d), _p(x1), _p(a2), _p(w2.detach()), _p(b2.detach()), _p(g1.detach()), _p(be1.detach()),
                                            _p(g2.detach()), _p(be2.detach()), _p(y), _p(y16), _p(stats), _stream()), "hamt_vis_embed_fwd")
        ctx.save_for_backward(x2 if x16 is None else None, x16, x1, a2, stats, w1, g1, w2, b2, g2)
        ctx.params = (g1, be1, g2, be2, b2, w2)
        ctx.args = (float(eps1), float(eps2), prec, img.shape, b1 is not None)
        ctx.bias_param = b1 if (b1 is not None and b1.is_leaf) else (None if b1 is None else False)
        if y16 is not None:
            ctx.mark_non_differentiable(y16)
        ctx.set_materialize_grads(False)
        return y.view(*img.shape[:-1], H), y16

    @staticmethod
    def backward(ctx, dy, _d16=None):
        if dy is None:
            return (None,) * 14
        x2, x16, x1, a2, stats, w1, g1, w2, b2, g2 = ctx.saved_tensors
        eps1, eps2, prec, ishape, has_b1 = ctx.args
        M, H = x1.shape
        A = a2.shape[1]
        dev = dy.device
        dy2 = dy.reshape(M, H).contiguous()
        from . import wgrad
        in_pass = wgrad.ENABLED and torch._C._current_graph_task_id() >= 0
        fast = x16 is not None
        Mp = _rup(M) if fast else 0
        dx16 = torch.empty(Mp, H, dtype=torch.bfloat16, device=dev) if fast else None
        dx = None if fast else torch.empty(M, H, dtype=torch.float32, device=dev)
        dst = [_grad_dst(p, p.shape, dev, in_pass) if ctx.needs_input_grad[i] else (None, None)
               for p, i in zip(ctx.params, (4, 5, 8, 9, 7, 6))]          # gamma_img, beta_img, gamma_ang, beta_ang, b_ang, w_ang
        scratch = [None if t is not None else torch.zeros(p.shape, dtype=torch.float32, device=dev) for (t, _), p in zip(dst, ctx.params)]
        out = [t if t is not None else s_ for (t, _), s_ in zip(dst, scratch)]
        if any(t is not None and r is None for t, r in dst):
            wgrad.queue(dev).current()              # (opens the pass: orders this stream behind an overlapped optimizer update)
        ws = torch.empty(L.workspace_bytes(L.WS_VIS_EMBED_BWD, M, H) // 4, dtype=torch.float32, device=dev)
        d = L.VisEmbedDesc(M, H, A, A, eps1, eps2, int(x1.dtype == torch.bfloat16), Mp)
        L.check(L.load().hamt_vis_embed_bwd(C.byref(d), _p(dy2), _p(x1), _p(a2), _p(w2.detach()), _p(b2.detach()), _p(g1.detach()), _p(g2.detach()),
                                            _p(stats), _p(dx), _p(dx16), _p(out[0]), _p(out[1]), _p(out[2]), _p(out[3]), _p(out[4]), _p(out[5]),
                                            _p(ws), _stream()), "hamt_vis_embed_bwd")
        for (t, r), p in zip(dst, ctx.params):
            if t is not None and r is None:
                wgrad.publish_slot_grad(p, t)
        rets = [r for _, r in dst]
        dimg, dw1, db1 = _linear_bwd(dx16[:M] if fast else dx, x2, x16, w1, prec, ctx.needs_input_grad[0], ctx.needs_input_grad[2],
                                     has_b1 and ctx.needs_input_grad[3], dy16=dx16, bias_param=ctx.bias_param)
        return ((dimg.view(ishape) if dimg is not None else None), None, dw1, db1, rets[0], rets[1], rets[5], rets[4], rets[2], rets[3],
                None, None, None, None)


def vis_embed_ok(img, ang, img_lin, ang_lin) -> bool:
    H = img_lin.weight.shape[0]
    return (VIS_EMBED and img.is_cuda and ang.shape[-1] == 4 and H % 64 == 0 and H <= 1024 and img.dtype == torch.float32
            and ang_lin.bias is not None and img_lin.weight.dtype == torch.float32)


def vis_embed(img, ang, img_lin, img_ln, ang_lin, ang_ln, prec, want16=False):
    y, y16 = VisEmbedFn.apply(img, ang, img_lin.weight, img_lin.bias, img_ln.weight, img_ln.bias, ang_lin.weight, ang_lin.bias,
                              ang_ln.weight, ang_ln.bias, img_ln.eps, ang_ln.eps, prec, want16)
    if y16 is not None:
        y._hamt_bf16 = (y16, y._version, y.data_ptr())
    return y


# ------------------------------------------------------------------------------------------ gathers / embeddings
class EmbedSumFn(torch.autograd.Function):
    """word[ids] + position[:L] + token_type[0]  (vilmodel.py:62-66; int64 gather is bit exact)."""

    @staticmethod
    def forward(ctx, ids, word, pos, typ):
        _chk(word, "EmbedSumFn")
        B, Lq = ids.shape
        H = word.shape[1]
        ids = ids.contiguous()
        z = torch.empty(B, Lq, H, dtype=torch.float32, device=word.device)
        L.check(L.load().hamt_embed_sum_fwd(B, Lq, H, _p(ids), _p(word.detach()), _p(pos.detach()), _p(typ.detach()), _p(z), _stream()),
                "hamt_embed_sum_fwd")
        ctx.save_for_backward(ids)
        ctx.shapes = (word.shape, pos.shape, typ.shape)
        ctx.tables = (word, pos, typ)
        return z

    @staticmethod
    def backward(ctx, dz):
        (ids,) = ctx.saved_tensors
        B, Lq = ids.shape
        dz = dz.contiguous()
        H = dz.shape[-1]
        dev = dz.device
        from . import wgrad
        in_pass = wgrad.ENABLED and torch._C._current_graph_task_id() >= 0
        outs, rets = [], []
        for p, shape in zip(ctx.tables, ctx.shapes):
            # a table that owns a (zero at the start of a step) slot in the optimizer's gradient arena: add there -- no zero fill
            # of the 94 MB word table, no copy into the arena afterwards (as ops.GatherRowsFn does); published as `.grad` at the
            # end of the pass, the tied MLM decoder's queued weight gradient then accumulates on top
            slot = getattr(p, "_hamt_grad_slot", None) if (in_pass and p.is_leaf and p.requires_grad) else None
            if (slot is not None and slot.shape == p.shape and slot.is_contiguous() and getattr(p, "_hamt_slot_zeroed", False)
                    and (p.grad is None or p.grad.data_ptr() == slot.data_ptr())):
                outs.append(slot)
                rets.append(None)
            else:
                t = torch.zeros(shape, dtype=torch.float32, device=dev)
                outs.append(t)
                rets.append(t)
        dword, dpos, dtyp = outs
        if any(r is None for r in rets):
            wgrad.queue(dev).current()              # (opens the pass: orders this stream behind an overlapped optimizer update)
        ws = torch.empty(L.workspace_bytes(L.WS_EMBED_BWD, B * Lq, H) // 4, dtype=torch.float32, device=dev)
        L.check(L.load().hamt_embed_sum_bwd(B, Lq, H, dword.shape[0], _p(ids), _p(dz), _p(dword), _p(dpos), _p(dtyp), _p(ws), ws.numel() * 4, _stream()),
                "hamt_embed_sum_bwd")      # (dtyp: row 0 of the type table)
        for p, r, o in zip(ctx.tables, rets, outs):
            if r is None:
                wgrad.publish_slot_grad(p, o)
        return None, rets[0], rets[1], rets[2]


def _scatter_add(R, W, dout, idx, table, unique=False):
    """table[idx[r]] += dout[r] in a fixed summation order (bit-reproducible): tables of a few rows (token / navigability types, the cls
    token) by per-block partial sums (hamt_scatter_add_rows_small), larger ones (position tables, compaction scatters) by one writer per
    table row walking its colliding source rows in row order (hamt_scatter_add_rows_ordered)"""
    T = table.numel() // W
    if unique:        # no two source rows share a table row (a compaction / slice): every element is added once, order cannot matter
        L.check(L.load().hamt_scatter_add_rows(R, W, _p(dout), W, 0, _p(idx), _p(table), W, _stream()), "hamt_scatter_add_rows")
    elif T <= 8 and table.is_contiguous():
        ws = torch.empty(64 * T * W, dtype=torch.float32, device=dout.device)
        L.check(L.load().hamt_scatter_add_rows_small(R, W, _p(dout), W, 0, _p(idx), T, _p(table), _p(ws), _stream()), "hamt_scatter_add_rows_small")
    else:
        ws = torch.empty(34 * max(R, 1), dtype=torch.int32, device=dout.device)
        L.check(L.load().hamt_scatter_add_rows_ordered(R, W, _p(dout), W, 0, _p(idx), _p(table), W, T, _p(ws), _stream()), "hamt_scatter_add_rows_ordered")


class GatherRowsFn(torch.autograd.Function):
    """out[r] = (base[r] if base is not None else 0) + table[idx[r]]  over rows of width W.
    Serves embedding lookups, boolean-mask compaction, anchor gathers and slices (idx arithmetic)."""

    @staticmethod
    def forward(ctx, table, idx, base, unique=False):
        _chk(table, "GatherRowsFn")
        ctx.unique = bool(unique)
        W = table.shape[-1]
        t2 = table.reshape(-1, W)
        if t2.stride(-1) != 1:
            t2 = t2.contiguous()
        idx = idx.reshape(-1).contiguous()
        R = idx.numel()
        b2 = base.reshape(R, W).contiguous() if base is not None else None
        out = torch.empty(R, W, dtype=torch.float32, device=table.device)
        L.check(L.load().hamt_gather_rows(R, W, _p(t2.detach()), _ld(t2), _p(idx), _p(b2), W, _p(out), W, 0, _stream()),
                "hamt_gather_rows")
        ctx.save_for_backward(idx)
        ctx.tshape, ctx.has_base, ctx.bshape = table.shape, base is not None, (base.shape if base is not None else None)
        ctx.table_param = table if (table.is_leaf and table.requires_grad and getattr(table, "_hamt_grad_slot", None) is not None) else None
        return out

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        dout = dout.contiguous()
        R, W = dout.shape
        dtab = None
        if ctx.needs_input_grad[0]:
            p = ctx.table_param
            slot = getattr(p, "_hamt_grad_slot", None) if p is not None else None
            from . import wgrad
            if (slot is not None and wgrad.ENABLED and slot.shape == p.shape and slot.is_contiguous() and getattr(p, "_hamt_slot_zeroed", False)
                    and (p.grad is None or p.grad.data_ptr() == slot.data_ptr()) and torch._C._current_graph_task_id() >= 0):
                # an embedding table that owns a slot in the optimizer's gradient arena (zero at the start of a step): add the rows
                # there -- no 94 MB zero fill for the 30 522 x 768 word table, no copy into the arena afterwards; published as
                # `.grad` at the end of the pass (the tied MLM decoder's queued weight gradient then accumulates on top)
                wgrad.queue(slot.device).current()         # (opens the pass: orders this stream behind an overlapped optimizer update)
                _scatter_add(R, W, dout, idx, slot, ctx.unique)
                wgrad.publish_slot_grad(p, slot)
            else:
                dtab = torch.zeros(ctx.tshape, dtype=torch.float32, device=dout.device)
                _scatter_add(R, W, dout, idx, dtab, ctx.unique)
        return dtab, None, (dout.view(ctx.bshape) if ctx.has_base else None), None


def gather_rows(table, idx, base=None, unique=False):
    """unique: the caller guarantees that no index repeats (a compaction of masked positions, a slice, one row per sample): the
    backward then adds every gradient row to its own table row, where the order of the adds cannot matter"""
    return GatherRowsFn.apply(table, idx, base, unique)


_CONST_IDX: dict = {}


def const_index(kind: str, *args, device) -> torch.Tensor:
    """A constant int64 index tensor, built once per (kind, arguments, device) and never modified: `zeros` n / `ones` n / `arange` n
    (0 .. n-1) / `arange_mul` (n, k) = arange(n) * k / `bcast` (n_outer, n_inner) = row -> outer index / `rows_but_last` (B, S) = b * S +
    s for s < S - 1.  A forward pass used to rebuild these with 1-3 tiny fill / arange / multiply launches each, every step (and every
    replay of a captured step)."""
    key = (kind, args, str(torch.device(device)))
    t = _CONST_IDX.get(key)
    if t is None:
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():      # (first use inside a capture: no persistent memory there)
            return _build_const(kind, args, device)
        with torch.no_grad():
            t = _CONST_IDX[key] = _build_const(kind, args, device)
        if t.is_cuda:       # (once per constant: whichever stream uses it next finds it written)
            torch.cuda.current_stream(t.device).synchronize()
    return t


def _build_const(kind, args, device):
    if kind == "zeros":
        return torch.zeros(args[0], dtype=torch.long, device=device)
    if kind == "ones":
        return torch.ones(args[0], dtype=torch.long, device=device)
    if kind == "arange":
        return torch.arange(args[0], dtype=torch.long, device=device)
    if kind == "arange_mul":
        return torch.arange(args[0], dtype=torch.long, device=device) * args[1]
    if kind == "bcast":
        return torch.arange(args[0], device=device).repeat_interleave(args[1])
    if kind == "rows_but_last":
        B, S = args
        return (torch.arange(B, device=device)[:, None] * S + torch.arange(S - 1, device=device)[None]).reshape(-1)
    raise ValueError(kind)


@torch.no_grad()
def extend_mask(mask: torch.Tensor) -> torch.Tensor:
    """(B, S) bool keep-mask -> additive (B, 1, 1, S) fp32 = (1 - m) * -10000 (vilmodel.py:597-599) in one launch"""
    if mask.dtype != torch.bool or not mask.is_cuda:
        return (1.0 - mask[:, None, None, :].to(torch.float32)) * -10000.0
    m = mask.contiguous()
    out = torch.empty(m.shape[0], 1, 1, m.shape[1], dtype=torch.float32, device=m.device)
    L.check(L.load().hamt_extend_mask(m.numel(), _p(m), _p(out), _stream()), "hamt_extend_mask")
    return out


@torch.no_grad()
def copy_rows_into(dst2d: torch.Tensor, col0: int, src2d: torch.Tensor):
    """dst2d[:, col0:col0 + W] = src2d  (fp32, rows of either side may be strided): one hamt_gather_rows launch with the
    identity index -- the in-place `cat` / `stack` of the no-grad paths (history cache, vision buffer)."""
    R, W = src2d.shape
    assert dst2d.shape[0] == R and dst2d.stride(1) == 1 and src2d.stride(1) == 1 and col0 + W <= dst2d.shape[1]
    L.check(L.load().hamt_gather_rows(R, W, _p(src2d), _ld(src2d), None, None, W, _p(dst2d), _ld(dst2d), col0, _stream()), "hamt_gather_rows")
    return dst2d


class Add3Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, c):
        _chk(a, "Add3Fn")
        a, b = a.contiguous(), b.contiguous()
        c = c.contiguous() if c is not None else None
        out = torch.empty_like(a)
        L.check(L.load().hamt_add3(a.numel(), _p(a), _p(b), _p(c), _p(out), _stream()), "hamt_add3")
        ctx.has_c = c is not None
        return out

    @staticmethod
    def backward(ctx, g):
        return g, g, (g if ctx.has_c else None)


def add3(a, b, c=None):
    return Add3Fn.apply(a, b, c)


class MeanMidFn(torch.autograd.Function):
    """(B,S,H) -> (B,H) mean over S (panorama mean pooling, vilmodel.py:563-564)."""

    @staticmethod
    def forward(ctx, x):
        _chk(x, "MeanMidFn")
        x = x.contiguous()
        B, S, H = x.shape
        y = torch.empty(B, H, dtype=torch.float32, device=x.device)
        L.check(L.load().hamt_mean_mid_fwd(B, S, H, _p(x), _p(y), _stream()), "hamt_mean_mid_fwd")
        ctx.shape = (B, S, H)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, S, H = ctx.shape
        dy = dy.contiguous()
        dx = torch.empty(B, S, H, dtype=torch.float32, device=dy.device)
        L.check(L.load().hamt_mean_mid_bwd(B, S, H, _p(dy), _p(dx), _stream()), "hamt_mean_mid_bwd")
        return dx


class MulBcastFn(torch.autograd.Function):
    """y[b,s,:] = a[b,s,:] * c[b,:]  (SAP fusion ob*txt_cls pretrain_cmt.py:176; ITM cls product vilmodel.py:722).
    `c` may be a strided row view such as txt_embeds[:, 0]."""

    @staticmethod
    def forward(ctx, a, c):
        _chk(a, "MulBcastFn")
        a = a.contiguous()
        B, S, H = a.shape
        if c.stride(-1) != 1 or (c.stride(0) % 4):
            c = c.contiguous()
        y = torch.empty_like(a)
        L.check(L.load().hamt_mul_bcast_fwd(B, S, H, _p(a), _p(c), c.stride(0), _p(y), _stream()), "hamt_mul_bcast_fwd")
        ctx.save_for_backward(a, c)
        return y

    @staticmethod
    def backward(ctx, dy):
        a, c = ctx.saved_tensors
        B, S, H = a.shape
        dy = dy.contiguous()
        da = torch.empty_like(a)
        dc = torch.empty(B, H, dtype=torch.float32, device=a.device)
        L.check(L.load().hamt_mul_bcast_bwd(B, S, H, _p(a), _p(c), c.stride(0), _p(dy), _p(da), _p(dc), _stream()), "hamt_mul_bcast_bwd")
        return da, dc


class FillWhereZeroFn(torch.autograd.Function):
    """x.masked_fill(flag == 0, value) (pretrain_cmt.py:177 does it in place on a fresh tensor; the result
    is the same object-wise for callers); backward zeroes the gradient at the filled positions."""

    @staticmethod
    def forward(ctx, x, flag, value):
        _chk(x, "FillWhereZeroFn")
        flag = flag.contiguous()
        assert flag.numel() == x.numel()
        y = x.contiguous().clone()
        L.check(L.load().hamt_fill_where_zero(y.numel(), _p(flag), _p(y), float(value), _stream()), "hamt_fill_where_zero")
        ctx.save_for_backward(flag)
        return y

    @staticmethod
    def backward(ctx, g):
        (flag,) = ctx.saved_tensors
        g = g.contiguous().clone()
        L.check(L.load().hamt_fill_where_zero(g.numel(), _p(flag), _p(g), 0.0, _stream()), "hamt_fill_where_zero")
        return g, None, None


class DropoutFn(torch.autograd.Function):
    """Feature dropout (finetune model_HAMT.py:32-52); same counter-based mask replayed in backward."""

    @staticmethod
    def forward(ctx, x, p):
        _chk(x, "DropoutFn")
        x = x.contiguous()
        y = torch.empty_like(x)
        ctx.cid, ctx.p = next_call_id(), float(p)
        L.check(L.load().hamt_dropout(x.numel(), _p(x), _p(y), ctx.p, ctx.cid, _p(rng_state(x.device)), _stream()), "hamt_dropout")
        return y

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        dx = torch.empty_like(g)
        L.check(L.load().hamt_dropout(g.numel(), _p(g), _p(dx), ctx.p, ctx.cid, _p(rng_state(g.device)), _stream()), "hamt_dropout")
        return dx, None


def fill_where_zero(x, flag, value):
    return FillWhereZeroFn.apply(x, flag, value)


def mean_mid(x):
    return MeanMidFn.apply(x)


def mul_bcast(a, c):
    return MulBcastFn.apply(a, c)


def embed_sum(ids, word, pos, typ):
    return EmbedSumFn.apply(ids, word, pos, typ)


def dropout(x, p, training):
    if not training or p <= 0:
        return x
    return DropoutFn.apply(x, p)


# ------------------------------------------------------------------------------------------ losses
class CrossEntropyFn(torch.autograd.Function):
    """F.cross_entropy(x, label, reduction='none') (pretrain_cmt.py:154, 180, 259)."""

    @staticmethod
    def forward(ctx, x, label):
        _chk(x, "CrossEntropyFn")
        assert x.dim() == 2
        if x.stride(1) != 1:
            x = x.contiguous()
        R, Cc = x.shape
        label = label.contiguous()
        loss = torch.empty(R, dtype=torch.float32, device=x.device)
        lse = torch.empty(R, dtype=torch.float32, device=x.device)
        L.check(L.load().hamt_ce_fwd(R, Cc, _p(x), x.stride(0), _p(label), _p(loss), _p(lse), _stream()), "hamt_ce_fwd")
        ctx.save_for_backward(x, label, lse)
        return loss

    @staticmethod
    def backward(ctx, g):
        x, label, lse = ctx.saved_tensors
        R, Cc = x.shape
        g = g.contiguous()
        ld = (Cc + 7) // 8 * 8                       # keep rows 16-byte aligned for the following GEMMs
        buf = torch.empty(R, ld, dtype=torch.float32, device=x.device)
        L.check(L.load().hamt_ce_bwd(R, Cc, _p(x), x.stride(0), _p(label), _p(lse), _p(g), _p(buf), ld, _stream()), "hamt_ce_bwd")
        return buf[:, :Cc], None


class MseFn(torch.autograd.Function):
    """F.mse_loss(x, t, reduction='none') (pretrain_cmt.py:197, 219)."""

    @staticmethod
    def forward(ctx, x, t):
        _chk(x, "MseFn")
        x, t = x.contiguous(), t.contiguous().to(torch.float32)
        loss = torch.empty_like(x)
        L.check(L.load().hamt_mse_fwd(x.numel(), _p(x), _p(t), _p(loss), _stream()), "hamt_mse_fwd")
        ctx.save_for_backward(x, t)
        return loss

    @staticmethod
    def backward(ctx, g):
        x, t = ctx.saved_tensors
        g = g.contiguous()
        dx = torch.empty_like(x)
        L.check(L.load().hamt_mse_bwd(x.numel(), _p(x), _p(t), _p(g), _p(dx), _stream()), "hamt_mse_bwd")
        return dx, None


class KlFn(torch.autograd.Function):
    """F.kl_div(F.log_softmax(x, -1), t, reduction='none').sum(1) (pretrain_cmt.py:239-240)."""

    @staticmethod
    def forward(ctx, x, t):
        _chk(x, "KlFn")
        assert x.dim() == 2
        if x.stride(1) != 1:
            x = x.contiguous()
        t = t.contiguous().to(torch.float32)
        R, Cc = x.shape
        loss = torch.empty(R, dtype=torch.float32, device=x.device)
        lse = torch.empty(R, dtype=torch.float32, device=x.device)
        L.check(L.load().hamt_kl_fwd(R, Cc, _p(x), x.stride(0), _p(t), Cc, _p(loss), _p(lse), _stream()), "hamt_kl_fwd")
        ctx.save_for_backward(x, t, lse)
        return loss

    @staticmethod
    def backward(ctx, g):
        x, t, lse = ctx.saved_tensors
        R, Cc = x.shape
        g = g.contiguous()
        ld = (Cc + 7) // 8 * 8
        buf = torch.empty(R, ld, dtype=torch.float32, device=x.device)
        L.check(L.load().hamt_kl_bwd(R, Cc, _p(x), x.stride(0), _p(t), Cc, _p(lse), _p(g), _p(buf), ld, _stream()), "hamt_kl_bwd")
        return buf[:, :Cc], None


def cross_entropy(x, label):
    return CrossEntropyFn.apply(x, label)


def mse_loss(x, t):
    return MseFn.apply(x, t)


class A2cFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logp, value, ent, reward, mask, last_value, gamma, ent_w):
        T, B = logp.shape
        dev = logp.device
        f = lambda t: None if t is None else t.detach().to(torch.float32).contiguous()
        logp_, value_, ent_, reward_, mask_, last_ = f(logp), f(value), f(ent), f(reward), f(mask), f(last_value)
        ret = torch.empty(T, B, dtype=torch.float32, device=dev)
        out = torch.empty(B, 3, dtype=torch.float32, device=dev)
        L.check(L.load().hamt_a2c_fwd(T, B, _p(reward_), _p(mask_), _p(value_), _p(logp_), _p(ent_), _p(last_), float(gamma), float(ent_w),
                                      _p(ret), _p(out), _stream()), "hamt_a2c_fwd")
        ctx.save_for_backward(ret, mask_, value_)
        ctx.meta = (T, B, float(ent_w), ent is not None)
        return out

    @staticmethod
    def backward(ctx, dout):
        ret, mask_, value_ = ctx.saved_tensors
        T, B, ent_w, has_ent = ctx.meta
        dev = ret.device
        # the caller reduces `out` with ONE scalar weight per rollout (sum, / total, / batch: agent_cmt.py:508-514)
        g = dout.reshape(-1)[:1].to(torch.float32).contiguous()
        dlogp = torch.empty(T, B, dtype=torch.float32, device=dev)
        dvalue = torch.empty(T, B, dtype=torch.float32, device=dev)
        dent = torch.empty(T, B, dtype=torch.float32, device=dev) if has_ent else None
        L.check(L.load().hamt_a2c_bwd(T, B, _p(ret), _p(mask_), _p(value_), ent_w, _p(g), _p(dlogp), _p(dvalue), _p(dent), _stream()), "hamt_a2c_bwd")
        return dlogp, dvalue, dent, None, None, None, None, None


def a2c_loss(logp, value, reward, mask, last_value=None, entropy=None, gamma=0.9, entropy_weight=0.01, normalize="total"):
    """The agent's RL loss over a whole rollout in two launches (finetune_src/r2r/agent_cmt.py:476-518): `logp`, `value`,
    `reward`, `mask` (and `entropy` for feedback == 'sample') are [T, B]; `last_value` [B] is the critic's value of the last
    state for episodes that have not ended and 0 for the others (:480-484).  normalize: 'total' (/ mask.sum()), 'batch'
    (/ B) or 'none' (:508-514).  Returns (loss, {policy, critic, entropy sums}) -- the sums are what the reference logs."""
    out = A2cFn.apply(logp, value, entropy, reward, mask, last_value, gamma, entropy_weight)
    parts = out.sum(0)
    loss = parts.sum()
    if normalize == "total":
        loss = loss / mask.to(torch.float32).sum()
    elif normalize == "batch":
        loss = loss / logp.shape[1]
    else:
        assert normalize == "none", normalize
    return loss, {"policy": parts[0].detach(), "critic": parts[1].detach(), "entropy": parts[2].detach()}


def kl_div_logsoftmax(x, t):
    return KlFn.apply(x, t)
