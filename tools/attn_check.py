#!/usr/bin/env python3
"""A/B of the attention kernels against an fp64 reference: prints max errors of out, dq, dk, dv per shape."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vln_hamt_amd import ops

def run(B, heads, Sq, Sk, prec="bf16"):
    H = heads * 64
    g = torch.Generator().manual_seed(0)
    q = torch.randn(B, Sq, H, generator=g); k = torch.randn(B, Sk, H, generator=g); v = torch.randn(B, Sk, H, generator=g)
    go = torch.randn(B, Sq, H, generator=g)
    mask = torch.zeros(B, Sk); mask[:, Sk - Sk // 4:] = -10000.0
    qd, kd, vd = (x.double().requires_grad_() for x in (q, k, v))
    def heads_(x, S): return x.view(B, S, heads, 64).transpose(1, 2)
    s = heads_(qd, Sq) @ heads_(kd, Sk).transpose(-1, -2) / 8.0 + mask.double()[:, None, None, :]
    o = (s.softmax(-1) @ heads_(vd, Sk)).transpose(1, 2).reshape(B, Sq, H)
    o.backward(go.double())
    qs = q.reshape(B * Sq, H).cuda().requires_grad_()
    kvs = torch.cat([k, v], -1).reshape(B * Sk, 2 * H).cuda().requires_grad_()
    out = ops.attention(qs, kvs, mask.cuda(), B, heads, 0.0, prec)
    out.backward(go.reshape(B * Sq, H).cuda())
    e = lambda a, b: float((a.detach().cpu().double() - b).abs().max())
    gkv = kvs.grad.view(B, Sk, 2 * H)
    print(f"B{B} h{heads} Sq{Sq} Sk{Sk}: out {e(out.view(B,Sq,H), o.detach()):.3e} dq {e(qs.grad.view(B,Sq,H), qd.grad):.3e} "
          f"dk {e(gkv[..., :H], kd.grad):.3e} dv {e(gkv[..., H:], vd.grad):.3e}")

for shp in [(2, 2, 80, 80), (2, 2, 16, 16), (2, 2, 32, 32), (2, 2, 48, 48), (2, 2, 64, 64), (2, 1, 80, 6), (2, 1, 6, 80), (1, 1, 128, 128)]:
    run(*shp)
