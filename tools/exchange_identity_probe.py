#!/usr/bin/env python3
"""One rank, fp32 wire: the gradient arena after a backward pass whose weight gradients go through the exchange's plan (launch groups,
ranges, an all-reduce over one rank) must equal, bit for bit, the arena after a plain backward pass.  Prints the parameters that differ.
usage: exchange_identity_probe.py [task ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("gloo", rank=0, world_size=1)
from _util import tiny_cfg
from oracle.hamt_oracle import make_state_dict, pretrain_param_shapes
from vln_hamt_amd.optim import AdamW
from vln_hamt_amd.parallel import OverlappedGradSync, ShardedGradSync
from vln_hamt_amd.synth import make_batch
from test_gpu_model import build
dev = torch.device("cuda", 0)
cfg = tiny_cfg()
sd = make_state_dict(pretrain_param_shapes(cfg), seed=5)
tasks = sys.argv[1:] or ["sap", "mlm", "sar", "mrc"]
for sharded in (False, True):
    m = build(cfg, sd, "bf16", train=True)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    named = list(m.named_parameters())
    o = AdamW([{"params": [p for _, p in named], "weight_decay": 0.0}], lr=1e-6, eps=1e-6)
    o.materialize()
    for t in tasks:
        b = make_batch(t, 4, cfg, seed=sum(map(ord, t)), ragged=True, device=dev, txt_len=20, hist_len=4)
        o.zero_grad()
        m(b, t, True).mean().backward()
        o._pack_grads()
        plain = o._flat_g.clone()
        act0 = list(o._active)
        o.zero_grad()
        sync = (ShardedGradSync if sharded else OverlappedGradSync)(o, n_groups=3, wire="fp32")
        try:
            m(b, t, True).mean().backward()
            sync(o)
            torch.cuda.synchronize()
            got = o._flat_g.clone()
        finally:
            sync.close()
        bad = []
        for (n, p), off, a in zip(named, [o._offs[o._index_of[id(p)]] for _, p in named], act0):
            if a and not torch.equal(plain[off:off + p.numel()], got[off:off + p.numel()]):
                d = (plain[off:off + p.numel()] - got[off:off + p.numel()]).abs()
                bad.append((n, float(d.max()), float(plain[off:off + p.numel()].abs().max())))
        print(f"[sharded={sharded} {t}] parameters whose gradient differs from the plain pass: {len(bad)}")
        for n, d, s in bad[:12]:
            print(f"      {n}: max |d| {d:.3e} (scale {s:.3e})")
dist.destroy_process_group()
