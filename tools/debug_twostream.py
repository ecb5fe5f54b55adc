#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from oracle.hamt_oracle import OracleConfig, make_state_dict, pretrain_param_shapes
from vln_hamt_amd import ops, streams
from vln_hamt_amd.synth import make_batch
import test_gpu_model as T
cfg = OracleConfig(); sd = make_state_dict(pretrain_param_shapes(cfg), seed=11)
model = T.build(cfg, sd, "bf16", train=True)
for mod in model.modules():
    if isinstance(mod, torch.nn.Dropout): mod.p = 0.0
b = make_batch("sap", 16, cfg, seed=8, txt_len=80, hist_len=5, ragged=True, device="cuda")
def run(two):
    streams.set_two_stream(two)
    ops.manual_seed(77, torch.device("cuda")); model.zero_grad(set_to_none=True)
    model(b, "sap", True).mean().backward(); torch.cuda.synchronize()
    return {n: p.grad.double().clone() for n, p in model.named_parameters() if p.grad is not None}
r0 = run(False); r1 = run(False)
for rep in range(3):
    g = run(True)
    worst = sorted(((float((g[n] - r0[n]).abs().max()), float(r0[n].abs().max()), n) for n in r0), reverse=True)[:4]
    print("rep", rep, [(f"{a:.2e}", f"{s:.2e}", n) for a, s, n in worst])
print("single vs single", sorted(((float((r1[n] - r0[n]).abs().max()), n) for n in r0), reverse=True)[:2])
