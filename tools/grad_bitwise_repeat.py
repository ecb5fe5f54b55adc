#!/usr/bin/env python3
"""Bitwise repeatability of one forward + backward of the tiny model (two streams, dropout off): which parameters' gradients are NOT
bit-identical over repetitions, and how many distinct values they take.  usage: grad_bitwise_repeat.py [task] [reps]"""
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from _util import tiny_cfg
from oracle.hamt_oracle import make_state_dict, pretrain_param_shapes
from test_gpu_model import build, _two_rank_schedule, _two_rank_batch
task = sys.argv[1] if len(sys.argv) > 1 else "mlm"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cfg = tiny_cfg()
seq, shapes, hyp = _two_rank_schedule(True)
m = build(cfg, make_state_dict(pretrain_param_shapes(cfg), seed=5), "bf16", train=True)
for mod in m.modules():
    if isinstance(mod, torch.nn.Dropout):
        mod.p = 0.0
named = list(m.named_parameters())
b = _two_rank_batch(task, 0, cfg, shapes)
seen = {}
junk = []
for it in range(reps):
    junk.append(torch.randn(1000 * (1 + it % 7), device="cuda"))      # (shift the allocator's state between repetitions)
    if len(junk) > 4:
        junk.pop(0)
    for _, p in named:
        p.grad = None
    loss = m(b, task, True).mean()
    loss.backward()
    torch.cuda.synchronize()
    seen.setdefault("(loss)", set()).add(float(loss))
    for n, p in named:
        if p.grad is not None:
            seen.setdefault(n, set()).add(hashlib.md5(p.grad.detach().cpu().numpy().tobytes()).hexdigest())
var = {n: len(v) for n, v in seen.items() if len(v) > 1}
print(f"[{task}, HAMT_VIS_EMBED={os.environ.get('HAMT_VIS_EMBED', '1')}] {len(var)} of {len(seen)} tensors take more than one value over {reps} repetitions")
for n, k in sorted(var.items(), key=lambda t: -t[1])[:25]:
    print(f"    {k:3d} distinct values: {n}")
