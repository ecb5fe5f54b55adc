#!/usr/bin/env python3
"""Which scatter-adds (ops._scatter_add: the backward of ops.gather_rows) does a B = 64 step issue?  (source rows, width, table rows) per task."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_model
from vln_hamt_amd import ops
from vln_hamt_amd.synth import make_batch, make_itm_rng
dev = torch.device("cuda", 0)
model, cfg = build_model("bf16", dev)
model.train()
calls = collections.Counter()
orig = ops._scatter_add
def spy(R, W, dout, idx, table):
    calls[(R, W, table.numel() // W)] += 1
    return orig(R, W, dout, idx, table)
ops._scatter_add = spy
for task in ("mlm", "sap", "sar", "sprel", "mrc", "itm"):
    calls.clear()
    b = make_batch(task, 64 if task != "itm" else 32, cfg, seed=1, device=dev, txt_len=80, hist_len=5)
    if task == "itm":
        r = make_itm_rng(b, seed=3); b["itm_neg_idxs"], b["itm_shuffled_pos_ids"] = r["neg_idxs"], r["shuffled_pos_ids"]
    model(b, task, True).mean().backward()
    torch.cuda.synchronize()
    print(task, sorted(calls.items()))
