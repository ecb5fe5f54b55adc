#!/usr/bin/env python3
"""Is a graph replay of the training step host-bound?  For one task's captured step: the host time of each replay call (no
synchronisation between calls) against the GPU time per step, for the single-graph replay and for the chain-by-chain split;
and, inside the split launch, the host time of every segment launch (HAMT_SPLIT_TIMING)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from vln_hamt_amd import ops, graph
from vln_hamt_amd.optim import AdamW
from vln_hamt_amd.optim.misc import NO_DECAY
from vln_hamt_amd.synth import make_batch

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
task = sys.argv[2] if len(sys.argv) > 2 else "mlm"
dev = torch.device("cuda", 0)
ops.manual_seed(1, dev)
model, cfg = B.build_model("bf16", dev)
named = list(model.named_parameters())
opt = AdamW([{"params": [p for n, p in named if not any(nd in n for nd in NO_DECAY)], "weight_decay": 0.01},
             {"params": [p for n, p in named if any(nd in n for nd in NO_DECAY)], "weight_decay": 0.0}], lr=5e-5, betas=(0.9, 0.98))
opt.materialize()
b = make_batch(task, bs, cfg, seed=3, txt_len=80, hist_len=5, mlm_exact=12 if task == "mlm" else None, device=dev)
gs = graph.GraphedTrainStep(model, opt, 5.0)
for _ in range(3):
    gs.step(task, b, task)
torch.cuda.synchronize()
for n in (1, 4, 16):
    host = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        t1 = time.perf_counter()
        gs.step(task, b, task)
        host.append((time.perf_counter() - t1) * 1e3)
    t_issue = (time.perf_counter() - t0) * 1e3
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e3
    print(f"B={bs} {task} split={graph.SPLIT}: {n} steps back to back: host time per step call {sum(host) / n:.3f} ms (first {host[0]:.3f}), all issued after {t_issue:.3f} ms, "
          f"done after {wall:.3f} ms = {wall / n:.3f} ms per step", flush=True)
