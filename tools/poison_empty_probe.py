#!/usr/bin/env python3
"""Find reads of uninitialised memory: every torch.empty() of a floating dtype is filled with NaN before it is handed out, then the
two-stream visual embedding runs forward + backward at the tiny model's shapes -- a NaN in any result means some kernel read an element
nobody wrote.  usage: poison_empty_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
_empty = torch.empty
def poisoned(*a, **k):
    t = _empty(*a, **k)
    if t.is_floating_point() and t.is_cuda:
        t.fill_(float("nan"))
    return t
torch.empty = poisoned
from vln_hamt_amd import ops
nn = torch.nn
dev = "cuda"
torch.manual_seed(0)
for (M, K, H) in ((16, 64, 128), (12, 64, 128), (148, 64, 128), (576, 64, 128), (20, 64, 128), (2368, 768, 768)):
    mods = nn.ModuleList([nn.Linear(K, H), nn.Linear(4, H), nn.LayerNorm(H, eps=1e-12), nn.LayerNorm(H, eps=1e-12)]).to(dev)
    img, ang, gy = torch.randn(M, K, device=dev), torch.randn(M, 4, device=dev), torch.randn(M, H, device=dev)
    for want16 in (False, True):
        for p in mods.parameters():
            p.grad = None
        x = img.clone().requires_grad_(True)
        y = ops.vis_embed(x, ang, mods[0], mods[2], mods[1], mods[3], "bf16", want16=want16)
        y.backward(gy)
        torch.cuda.synchronize()
        bad = [n for n, t in [("y", y), ("dimg", x.grad)] + [(n, p.grad) for n, p in mods.named_parameters()] if t is None or bool(torch.isnan(t).any())]
        y16 = ops.shadow16(y)
        if y16 is not None and bool(torch.isnan(y16.float()).any()):
            bad.append("y16")
        print((M, K, H), "want16" if want16 else "", "NaN in:", bad or "nothing")

# phase 2: the tiny model's training loop (both embedder variants are whatever HAMT_VIS_EMBED selects) with every torch.empty poisoned
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from _util import tiny_cfg
from oracle.hamt_oracle import make_state_dict, pretrain_param_shapes
from vln_hamt_amd.optim import AdamW, clip_grad_norm_
from test_gpu_model import build, _two_rank_schedule, _two_rank_batch
cfg = tiny_cfg()
seq, shapes, hyp = _two_rank_schedule(True)
m = build(cfg, make_state_dict(pretrain_param_shapes(cfg), seed=5), "bf16", train=True)
for mod in m.modules():
    if isinstance(mod, torch.nn.Dropout):
        mod.p = 0.0
named = list(m.named_parameters())
o = AdamW([{"params": [p for _, p in named], "weight_decay": 0.0}], betas=(0.9, 0.98), **hyp)
o.materialize()
for r in range(2):
    for t in ("sap", "mlm", "sar", "mrc"):
        b = _two_rank_batch(t, r, cfg, shapes)
        l_ = m(b, t, True).mean()
        l_.backward()
        o._pack_grads()
        torch.cuda.synchronize()
        nan_g = [n for n, p in named if p.grad is not None and bool(torch.isnan(p.grad).any())]
        print(f"[model rank-{r} batch {t}] loss {float(l_):.6f}; NaN gradients: {nan_g[:8] or 'none'}")
        clip_grad_norm_(m.parameters(), 5.0, optimizer=o)
        o.step(); o.zero_grad()
print("NaN parameters after the loop:", [n for n, p in named if bool(torch.isnan(p).any())][:8] or "none")
