#!/usr/bin/env python3
"""Where do torch.zeros / Tensor.zero_ calls of one training step come from?  (monkeypatch + traceback)"""
import os, sys, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from vln_hamt_amd.synth import make_batch, make_itm_rng
from vln_hamt_amd.optim import AdamW, clip_grad_norm_
model, cfg = bench.build_model("bf16", torch.device("cuda"))
opt = AdamW(model.parameters(), lr=1e-5)
opt.materialize()
cnt = collections.Counter()
on = [False]
def wrap(name, fn):
    def f(*a, **k):
        if on[0]:
            st = [f"{os.path.basename(fr.filename)}:{fr.lineno}" for fr in traceback.extract_stack()[:-1] if "vln" in fr.filename][-2:]
            out = fn(*a, **k)
            cnt[(name, tuple(out.shape), str(out.dtype), " <- ".join(reversed(st)))] += 1
            return out
        return fn(*a, **k)
    return f
for n in ("zeros", "zeros_like", "full", "ones"):
    setattr(torch, n, wrap(n, getattr(torch, n)))
def wrapm(name):
    orig = getattr(torch.Tensor, name)
    def f(self, *a, **k):
        if on[0]:
            st = [f"{os.path.basename(fr.filename)}:{fr.lineno}" for fr in traceback.extract_stack()[:-1] if "vln" in fr.filename][-2:]
            cnt[("T." + name, tuple(self.shape), str(self.dtype), " <- ".join(reversed(st)))] += 1
        return orig(self, *a, **k)
    setattr(torch.Tensor, name, f)
for n in ("zero_", "fill_", "masked_fill", "masked_fill_", "contiguous", "clone", "to", "float", "add_", "__add__", "__mul__", "sum", "mean"):
    wrapm(n)
for task in sys.argv[1:] or ["mlm"]:
    b = make_batch(task, 64, cfg, seed=1, txt_len=80, hist_len=5, mlm_exact=12 if task == "mlm" else None, device="cuda")
    if task == "itm":
        r = make_itm_rng(b, seed=1); b["itm_neg_idxs"], b["itm_shuffled_pos_ids"] = r["neg_idxs"], r["shuffled_pos_ids"]
    def step():
        loss = model(b, task, True).mean(); loss.backward()
        clip_grad_norm_(model.parameters(), 5.0, optimizer=opt); opt.step(); opt.zero_grad()
    step(); step()
    cnt.clear(); on[0] = True; step(); on[0] = False
    print("==== task", task)
    for k, v in cnt.most_common(30):
        print(f"{v:4d} {k[0]:10s} {str(k[1]):18s} {k[2]:15s} {k[3]}")
