#!/usr/bin/env python3
"""Repeat one forward + backward of the tiny model (two compute streams, dropout off) many times while a second process keeps the
GPU busy, and report every repetition whose loss or gradients differ from the first by more than float-atomics noise.
usage: step_repeat_stress.py [task=sap] [reps=300]     (spawns its own load process unless HAMT_STRESS_CHILD is set)"""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
task = sys.argv[1] if len(sys.argv) > 1 else "sap"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
child = os.environ.get("HAMT_STRESS_CHILD") == "1"
load = None
if not child:
    load = subprocess.Popen([sys.executable, os.path.abspath(__file__), task, str(reps * 3)], env=dict(os.environ, HAMT_STRESS_CHILD="1"),
                            stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
from _util import tiny_cfg
from oracle.hamt_oracle import make_state_dict, pretrain_param_shapes
from vln_hamt_amd.synth import make_batch
from test_gpu_model import build
dev = torch.device("cuda", 0)
cfg = tiny_cfg()
m = build(cfg, make_state_dict(pretrain_param_shapes(cfg), seed=5), "bf16", train=True)
for mod in m.modules():
    if isinstance(mod, torch.nn.Dropout):
        mod.p = 0.0
named = list(m.named_parameters())
b = make_batch(task, 4, cfg, seed=sum(map(ord, task)), ragged=True, device=dev, txt_len=20, hist_len=4)
ref, bad = None, 0
for it in range(reps):
    for _, p in named:
        p.grad = None
    loss = m(b, task, True).mean()
    loss.backward()
    torch.cuda.synchronize()
    cur = (float(loss), {n: p.grad.detach().clone() for n, p in named if p.grad is not None})
    if ref is None:
        ref = cur
        continue
    worst, who = 0.0, None
    for n, g in cur[1].items():
        r = ref[1][n]
        e = float((g - r).abs().max()) / max(1e-12, float(r.abs().max()))
        if e > worst:
            worst, who = e, n
    if abs(cur[0] - ref[0]) > 1e-6 * abs(ref[0]) or worst > 1e-4:
        bad += 1
        if not child and bad <= 10:
            print(f"  rep {it}: loss {cur[0]:.8f} vs {ref[0]:.8f}, worst gradient difference {worst:.2e} ({who})", flush=True)
if not child:
    print(f"[{task}, HAMT_VIS_EMBED={os.environ.get('HAMT_VIS_EMBED', '1')}] {bad} of {reps - 1} repetitions differ from the first", flush=True)
    load.wait()
