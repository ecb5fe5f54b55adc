"""HF-style AdamW (pretrain_src/optim/adamw.py:13-112) as HIP kernels over flat fp32 arenas.

On the first step the parameters of every group are re-homed into ONE contiguous fp32 arena per group
(``p.data`` becomes a view; names, shapes and values are unchanged), with matching flat gradient / exp_avg /
exp_avg_sq arenas.  ``step()`` then costs a handful of launches: gradients are packed into the arena, the
update runs per maximal run of parameters that (a) received a gradient -- parameters with ``grad is None``
are skipped like the reference's ``continue`` (:70-71) -- and (b) share a step count (bias correction :93-97).
``clip_grad_norm_`` computes the global L2 norm on device (one reduction over the arena) and defers the
scaling into the update kernel, so the clip costs no extra pass over the gradients.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Iterable

import torch
from torch.optim import Optimizer

from .. import _lib as L
from ..ops import _p, _stream


class AdamW(Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0, correct_bias=True):
        if lr < 0.0:
            raise ValueError("Invalid learning rate: {} - should be >= 0.0".format(lr))
        if not 0.0 <= betas[0] < 1.0:
            raise ValueError("Invalid beta parameter: {} - should be in [0.0, 1.0[".format(betas[0]))
        if not 0.0 <= betas[1] < 1.0:
            raise ValueError("Invalid beta parameter: {} - should be in [0.0, 1.0[".format(betas[1]))
        if not 0.0 <= eps:
            raise ValueError("Invalid epsilon value: {} - should be >= 0.0".format(eps))
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, correct_bias=correct_bias))
        self._flat = None          # per group: dict(p=, g=, m=, v=, offs=[...], steps=[...])
        self._pending_clip = None  # (gnorm_sq device scalar, max_norm)
        self._packed = False

    # ---------------------------------------------------------------- arenas
    def _build(self):
        self._flat = []
        for group in self.param_groups:
            ps = [p for p in group["params"]]
            if not ps:
                self._flat.append(None)
                continue
            dev = ps[0].device
            if not ps[0].is_cuda:
                raise L.HamtError("AdamW: parameters must live on the GPU (no CPU fallback)")
            offs, n = [], 0
            for p in ps:
                offs.append(n)
                n += (p.numel() + 3) // 4 * 4            # keep every tensor 16-byte aligned inside the arena
            flat_p = torch.zeros(n, dtype=torch.float32, device=dev)
            for p, o in zip(ps, offs):
                flat_p[o:o + p.numel()].copy_(p.data.reshape(-1))
                p.data = flat_p[o:o + p.numel()].view(p.shape)
            self._flat.append(dict(p=flat_p, g=torch.zeros_like(flat_p), m=torch.zeros_like(flat_p),
                                   v=torch.zeros_like(flat_p), offs=offs, n=n, steps=[0] * len(ps), params=ps))
        self._hyper = torch.zeros(3, dtype=torch.float32, device=dev)
        self._hyper_host = torch.zeros(3, dtype=torch.float32).pin_memory() if torch.cuda.is_available() else torch.zeros(3)
        self._gnorm = torch.zeros(1, dtype=torch.float32, device=dev)
        self._ws = torch.empty(1024, dtype=torch.float32, device=dev)

    def materialize(self):
        """Build the flat arenas now (re-homes p.data); call before wrapping the model in DDP."""
        if self._flat is None:
            self._build()
        return self

    def _pack_grads(self):
        """Copy the autograd-produced gradients into the flat arenas (one fused multi-tensor copy per group)."""
        if self._flat is None:
            self._build()
        for fl in self._flat:
            if fl is None:
                continue
            src, dst = [], []
            for p, o in zip(fl["params"], fl["offs"]):
                if p.grad is not None:
                    src.append(p.grad.reshape(-1))
                    dst.append(fl["g"][o:o + p.numel()])
            fl["active"] = [p.grad is not None for p in fl["params"]]
            if src:
                torch._foreach_copy_(dst, src)
        self._packed = True

    def global_grad_sumsq(self) -> torch.Tensor:
        """device scalar sum(g^2) over every parameter that has a gradient."""
        self._pack_grads()
        lib = L.load()
        first = True
        for fl in self._flat:
            if fl is None:
                continue
            for (a, b) in self._runs(fl, by_step=False):
                L.check(lib.hamt_sumsq(b - a, C.c_void_p(fl["g"].data_ptr() + 4 * a), _p(self._gnorm), int(not first),
                                       _p(self._ws), _stream()), "hamt_sumsq")
                first = False
        if first:
            self._gnorm.zero_()
        return self._gnorm

    @staticmethod
    def _runs(fl, by_step=True):
        """maximal [start, end) element ranges of consecutive active parameters (sharing a step count)."""
        runs, cur = [], None
        ps, offs, act, steps = fl["params"], fl["offs"], fl["active"], fl["steps"]
        for i, p in enumerate(ps):
            if not act[i]:
                if cur:
                    runs.append(cur)
                    cur = None
                continue
            end = offs[i + 1] if i + 1 < len(ps) else fl["n"]
            if cur and (not by_step or cur[2] == steps[i]):
                cur[1] = end
            else:
                if cur:
                    runs.append(cur)
                cur = [offs[i], end, steps[i], i]
        if cur:
            runs.append(cur)
        return [(r[0], r[1]) if not by_step else tuple(r) for r in runs]

    # ---------------------------------------------------------------- step
    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        if not self._packed:
            self._pack_grads()
        lib = L.load()
        gn, max_norm = (self._pending_clip if self._pending_clip is not None else (None, 0.0))
        touched = []
        for group, fl in zip(self.param_groups, self._flat):
            if fl is None:
                continue
            b1, b2 = group["betas"]
            for i, a in enumerate(fl["active"]):
                if a:
                    fl["steps"][i] += 1
            for (s, e, t, _) in self._runs(fl, by_step=True):
                step_size = group["lr"]
                if group["correct_bias"]:
                    step_size = step_size * math.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)
                self._hyper_host[0], self._hyper_host[1], self._hyper_host[2] = group["lr"], step_size, max_norm
                hyper = self._hyper_host.to(self._hyper.device, non_blocking=False)   # tiny H2D; value frozen per launch
                off = 4 * s
                L.check(lib.hamt_adamw_flat(e - s, C.c_void_p(fl["p"].data_ptr() + off), C.c_void_p(fl["g"].data_ptr() + off),
                                            C.c_void_p(fl["m"].data_ptr() + off), C.c_void_p(fl["v"].data_ptr() + off), None,
                                            _p(hyper), _p(gn), b1, b2, group["eps"], group["weight_decay"], 0, _stream()),
                        "hamt_adamw_flat")
            touched.append(fl["p"])
        for t in touched:                       # parameters changed outside autograd's view: bump versions so
            torch.autograd.graph.increment_version(t)   # cached bf16 weight shadows are refreshed
        self._pending_clip = None
        self._packed = False
        return loss

    def zero_grad(self, set_to_none: bool = True):
        super().zero_grad(set_to_none=set_to_none)
        self._packed = False

    def flat_state(self):
        return self._flat


def clip_grad_norm_(parameters: Iterable[torch.Tensor], max_norm: float, optimizer: AdamW = None) -> torch.Tensor:
    """torch.nn.utils.clip_grad_norm_ (main_r2r.py:271-273) on the GPU: returns the total L2 norm (device tensor,
    no host sync).  With `optimizer` (our AdamW) the scaling min(1, max_norm/(norm+1e-6)) is fused into the
    next ``optimizer.step()``; without it the gradients are scaled in place."""
    lib = L.load()
    if optimizer is not None:
        gsq = optimizer.global_grad_sumsq()
        optimizer._pending_clip = (gsq, float(max_norm))
        return gsq.sqrt()[0]
    grads = [p.grad for p in parameters if p.grad is not None]
    if not grads:
        return torch.zeros(())
    dev = grads[0].device
    gsq = torch.zeros(1, dtype=torch.float32, device=dev)
    ws = torch.empty(1024, dtype=torch.float32, device=dev)
    flat = [g.contiguous() for g in grads]
    for i, g in enumerate(flat):
        if g.data_ptr() % 16:
            g = g.clone()
            flat[i] = g
        L.check(lib.hamt_sumsq(g.numel(), _p(g), _p(gsq), int(i > 0), _p(ws), _stream()), "hamt_sumsq")
    for g, orig in zip(flat, grads):
        L.check(lib.hamt_clip_scale(g.numel(), _p(g), _p(gsq), float(max_norm), _stream()), "hamt_clip_scale")
        if g.data_ptr() != orig.data_ptr():
            orig.copy_(g)
    return gsq.sqrt()[0]
