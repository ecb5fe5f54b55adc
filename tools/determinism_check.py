#!/usr/bin/env python3
"""Run-to-run differences of the gradients of one forward + backward per task (same inputs, same weights, dropout off): what is
left must be explainable by the order of fp32 atomic adds (embedding tables, shared LayerNorm parameters).  usage: determinism_check.py [B]
(HAMT_NO_XSTREAM=1 for the single-stream order)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from vln_hamt_amd import ops
from vln_hamt_amd.synth import make_batch, make_itm_rng
dev = torch.device("cuda")
ops.manual_seed(1, dev)
model, cfg = bench.build_model("bf16", dev)
for mod in model.modules():
    if isinstance(mod, torch.nn.Dropout):
        mod.p = 0.0
named = list(model.named_parameters())
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
for it, task in enumerate(["mlm", "sap", "sar", "sprel", "mrc", "itm"]):
    b = make_batch(task, B, cfg, seed=300 + it, txt_len=80, hist_len=5, ragged=True, mlm_exact=7 if task == "mlm" else None, device=dev)
    if task == "itm":
        r = make_itm_rng(b, seed=it); b["itm_neg_idxs"], b["itm_shuffled_pos_ids"] = r["neg_idxs"], r["shuffled_pos_ids"]
    runs = []
    for rep in range(int(os.environ.get("HAMT_DET_RUNS", "3"))):
        for p in model.parameters():
            p.grad = None
        loss = model(b, task, True).mean()
        loss.backward()
        torch.cuda.synchronize()
        runs.append((float(loss), {n: p.grad.detach().clone() for n, p in named if p.grad is not None}))
    worst = []
    for n in runs[0][1]:
        a = runs[0][1][n].double()
        d = max(float((a - r[1][n].double()).abs().max()) for r in runs[1:])
        if d > 0 and not n.endswith("key.bias"):      # (an attention's key bias has a zero true gradient: rounding noise only)
            worst.append((d / max(float(a.abs().max()), 1e-30), d, n))
    worst.sort(reverse=True)
    print(f"{task}: losses {[r[0] for r in runs]}; parameters whose gradient differs between runs: {len(worst)} of {len(runs[0][1])}")
    for w in worst[:5]:
        print(f"     rel {w[0]:.2e} abs {w[1]:.2e}  {w[2]}")
