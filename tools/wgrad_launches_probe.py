#!/usr/bin/env python3
"""The weight-gradient launches of one backward pass per task at B = 64: tile class, work and time of each (hamt_debug_wgrad_timing).
usage: wgrad_launches_probe.py [batch]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_model
from vln_hamt_amd import _lib as L
from vln_hamt_amd.optim import AdamW
from vln_hamt_amd.synth import make_batch, make_itm_rng
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda", 0)
model, cfg = build_model("bf16", dev)
o = AdamW([{"params": list(model.parameters()), "weight_decay": 0.0}], lr=1e-5)
o.materialize()
lib = L.load()
for i, task in enumerate(("mlm", "sap", "mrc", "itm")):
    b = make_batch(task, B if task != "itm" else 2 * B, cfg, seed=i, device=dev)
    if task == "itm":
        r = make_itm_rng(b, seed=i); b["itm_neg_idxs"], b["itm_shuffled_pos_ids"] = r["neg_idxs"], r["shuffled_pos_ids"]
    for rep in range(2):
        o.zero_grad()
        loss = model(b, task, True).mean()
        torch.cuda.synchronize()
        lib.hamt_debug_wgrad_timing(1)
        loss.backward()
        torch.cuda.synchronize()
        us, rows, fl = (C.c_float * 64)(), (C.c_int * 64)(), (C.c_double * 64)()
        k = lib.hamt_debug_wgrad_times(us, rows, fl, 64)
        lib.hamt_debug_wgrad_timing(0)
    print(f"== {task}: " + "; ".join(f"{rows[j]}-row tiles {fl[j] / 1e9:.1f} GFLOP {us[j]:.0f} us = {fl[j] / us[j] / 1e6:.0f} TFLOP/s" for j in range(k)))
