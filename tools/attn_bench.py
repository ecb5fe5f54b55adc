#!/usr/bin/env python3
"""Micro-benchmark of hamt_attn_small_fwd/bwd (bf16) on the HAMT shapes.  usage: attn_bench.py [B,heads,Sq,Sk ...]"""
import ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vln_hamt_amd import _lib as L, ops

def bench(B, heads, Sq, Sk, p_drop=0.1, use_mask=True, iters=30):
    H = heads * 64
    dev = "cuda"
    lib = L.load()
    packed = Sq == Sk
    if packed:
        qkv = torch.randn(B * Sq, 3 * H, device=dev).to(torch.bfloat16)
        q, k, v = qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:]
        dqkv = torch.empty_like(qkv); dq, dk, dv = dqkv[:, :H], dqkv[:, H:2 * H], dqkv[:, 2 * H:]
        ldq = ldk = ldv = 3 * H
    else:
        q = torch.randn(B * Sq, H, device=dev).to(torch.bfloat16)
        kv = torch.randn(B * Sk, 2 * H, device=dev).to(torch.bfloat16)
        k, v = kv[:, :H], kv[:, H:]
        dq = torch.empty_like(q); dkv = torch.empty_like(kv); dk, dv = dkv[:, :H], dkv[:, H:]
        ldq, ldk, ldv = H, 2 * H, 2 * H
    o = torch.empty(B * Sq, H, device=dev, dtype=torch.bfloat16)
    do = torch.randn(B * Sq, H, device=dev).to(torch.bfloat16)
    lse = torch.empty(B * heads * Sq, device=dev)
    mask = torch.zeros(B, Sk, device=dev) if use_mask else None
    d = L.AttnDesc(B, heads, Sq, Sk, 64, ldq, ldk, ldv, H, L.HAMT_BF16, L.HAMT_BF16, 0.125, p_drop, 7, L.PREC_BF16)
    rng = ops.rng_state(torch.device(dev))
    p = ops._p
    def fwd(): L.check(lib.hamt_attn_small_fwd(C.byref(d), p(q), p(k), p(v), p(mask), p(o), p(lse), p(rng), ops._stream()), "fwd")
    def bwd(): L.check(lib.hamt_attn_small_bwd(C.byref(d), p(q), p(k), p(v), p(mask), p(o), p(do), p(lse), None, p(dq), p(dk), p(dv), p(rng), ops._stream()), "bwd")
    res = []
    for fn in (fwd, bwd):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters): fn()
        e.record(); torch.cuda.synchronize()
        res.append(s.elapsed_time(e) / iters * 1e3)
    return res

if __name__ == "__main__":
    shapes = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]] or [(64, 12, 80, 80), (320, 12, 36, 36), (64, 12, 80, 6), (64, 12, 6, 80), (64, 12, 37, 80)]
    for shp in shapes:
        for pd, um in ((0.1, True), (0.0, True), (0.0, False)):
            f, b = bench(*shp, p_drop=pd, use_mask=um)
            print(f"B{shp[0]} h{shp[1]} Sq{shp[2]} Sk{shp[3]} p_drop {pd} mask {int(um)}: fwd {f:7.1f} us  bwd {b:7.1f} us")
