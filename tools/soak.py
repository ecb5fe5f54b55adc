#!/usr/bin/env python3
"""Soak run of the graph-replayed training step: N steps of the 6-task mix over 12 fixed synthetic batches (B=32) --
the model must overfit them (losses fall, stay finite).  usage: soak.py [steps]"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from vln_hamt_amd import ops
from vln_hamt_amd.graph import GraphedTrainStep
from vln_hamt_amd.optim import AdamW
from vln_hamt_amd.optim.misc import NO_DECAY
from vln_hamt_amd.parallel import TaskSchedule
from vln_hamt_amd.synth import make_batch, make_itm_rng

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 360
dev = torch.device("cuda")
ops.manual_seed(1, dev)
model, cfg = bench.build_model("bf16", dev)
named = list(model.named_parameters())
opt = AdamW([{"params": [p for n, p in named if not any(nd in n for nd in NO_DECAY)], "weight_decay": 0.01},
             {"params": [p for n, p in named if any(nd in n for nd in NO_DECAY)], "weight_decay": 0.0}], lr=5e-5, betas=(0.9, 0.98))
gs = GraphedTrainStep(model, opt, 5.0)
sched = TaskSchedule(cyclic=True)
batches = {}
hist = collections.defaultdict(list)
for s in range(steps):
    task = sched.task_at(s)
    key = (task, s % 12)
    if key not in batches:
        ragged = os.environ.get("SOAK_RAGGED") == "1"      # ragged lengths: the text packing path (varlen attention, packed x-layers)
        b = make_batch(task, int(os.environ.get("SOAK_B", "32")), cfg, seed=100 + s % 12, txt_len=80, hist_len=7 if ragged else 5, ragged=ragged,
                       mlm_exact=12 if (task == "mlm" and not ragged) else None, device=dev)
        if task == "itm":
            r = make_itm_rng(b, seed=s); b["itm_neg_idxs"], b["itm_shuffled_pos_ids"] = r["neg_idxs"], r["shuffled_pos_ids"]
        batches[key] = b
    for g in opt.param_groups:
        g["lr"] = 5e-5 * min(1.0, (s + 1) / 50.0)
    loss = gs.step(key, batches[key], task)
    hist[task].append(float(loss))
for task, v in hist.items():
    n = max(1, len(v) // 5)
    first, last = sum(v[:n]) / n, sum(v[-n:]) / n
    ok = all(x == x and abs(x) < 1e6 for x in v)
    print(f"{task:6s} {len(v):4d} steps: mean loss first fifth {first:9.4f} -> last fifth {last:9.4f}  finite={ok}")
