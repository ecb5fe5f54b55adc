#!/usr/bin/env python3
"""Which parameters' gradients still arrive as autograd tensors and are copied into the gradient arena by optim.AdamW._pack_grads
(everything else is written in place by its producer)?  One eager step per task at B = 64.  usage: packed_grads_probe.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_model
from vln_hamt_amd.optim import AdamW
from vln_hamt_amd.optim.misc import NO_DECAY
from vln_hamt_amd.synth import make_batch, make_itm_rng

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda", 0)
model, cfg = build_model("bf16", dev)
named = list(model.named_parameters())
opt = AdamW([{"params": [p for n, p in named if not any(nd in n for nd in NO_DECAY)], "weight_decay": 0.01},
             {"params": [p for n, p in named if any(nd in n for nd in NO_DECAY)], "weight_decay": 0.0}], lr=5e-5)
opt.materialize()
name_of = {id(p): n for n, p in named}
for i, task in enumerate(("mlm", "sap", "sar", "sprel", "mrc", "itm")):
    b = make_batch(task, B if task != "itm" else 2 * B, cfg, seed=i, device=dev)
    if task == "itm":
        r = make_itm_rng(b, seed=i)
        b["itm_neg_idxs"], b["itm_shuffled_pos_ids"] = r["neg_idxs"], r["shuffled_pos_ids"]
    opt.zero_grad()
    model(b, task, True).mean().backward()
    base = opt._flat_g.data_ptr()
    rows = [(name_of[id(p)], p.numel()) for p, o in zip(opt._params, opt._offs) if p.grad is not None and p.grad.data_ptr() != base + 4 * o]
    print(f"== {task}: {len(rows)} gradients packed, {sum(n for _, n in rows) * 4 / 1e6:.2f} MB")
    for n, k in rows:
        print(f"   {k:9d}  {n}")
