#!/usr/bin/env python3
"""Micro-benchmark of the flat-arena optimizer step on the R2R-canon model (174.8 M parameters): sum of squares + AdamW table kernel,
HIP-event timed; bytes = 34 B/param for the update (p, g, m, v read; p, m, v, zero-g, bf16 shadow written), 4 B/param for the norm."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_model
from vln_hamt_amd.optim import AdamW
from vln_hamt_amd.optim.misc import NO_DECAY
dev = torch.device("cuda", 0)
model, cfg = build_model("bf16", dev)
named = list(model.named_parameters())
groups = [{"params": [p for n, p in named if not any(nd in n for nd in NO_DECAY)], "weight_decay": 0.01},
          {"params": [p for n, p in named if any(nd in n for nd in NO_DECAY)], "weight_decay": 0.0}]
opt = AdamW(groups, lr=5e-5, betas=(0.9, 0.98)).materialize()
opt._flat_g.normal_()
n = opt._n
act = [True] * len(opt._params)
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
opt._packed = True
opt.prepare_step(act)
def upd():
    opt._flat_g.normal_() if False else None
    opt.launch_step()
us = timeit(upd)
print(f"adamw_table: {us:8.1f} us  {34.0 * n / us / 1e6:6.2f} TB/s ({n/1e6:.1f} M params)")
def nrm():
    opt._packed = True
    opt.global_grad_sumsq()
us = timeit(nrm)
print(f"sumsq:       {us:8.1f} us  {4.0 * n / us / 1e6:6.2f} TB/s")
