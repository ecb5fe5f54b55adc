#!/usr/bin/env python3
"""The one-process reference loop of tests/test_gpu_model.py::test_two_ranks_on_one_gpu_match_averaged_gradients (24 steps, both ranks'
batches per step, bf16-wire arithmetic mimicked, eps 1e-6) run several times from the same start: are the final parameters the same
bit for bit?  usage: reference_loop_repeat.py [runs=4]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from _util import tiny_cfg
from oracle.hamt_oracle import make_state_dict, pretrain_param_shapes
from vln_hamt_amd.optim import AdamW, clip_grad_norm_
from vln_hamt_amd.optim.misc import NO_DECAY
from test_gpu_model import build, _two_rank_schedule, _two_rank_batch
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cfg = tiny_cfg()
seq, shapes, hyp = _two_rank_schedule(True)
res = []
all_snaps = []
for run in range(runs):
    sd = make_state_dict(pretrain_param_shapes(cfg), seed=5)
    m = build(cfg, sd, "bf16", train=True)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    named = list(m.named_parameters())
    o = AdamW([{'params': [p for n, p in named if not any(nd in n for nd in NO_DECAY)], 'weight_decay': 0.01},
               {'params': [p for n, p in named if any(nd in n for nd in NO_DECAY)], 'weight_decay': 0.0}], betas=(0.9, 0.98), **hyp)
    o.materialize()
    bs = [{t: _two_rank_batch(t, r, cfg, shapes) for t in set(seq)} for r in range(2)]
    losses = []
    snaps = []
    for t in seq:
        l_ = m(bs[0][t], t, True).mean(); losses.append(float(l_)); l_.backward()
        o._pack_grads(); g0 = o._flat_g.clone(); o.zero_grad()
        l_ = m(bs[1][t], t, True).mean(); losses.append(float(l_)); l_.backward()
        o._pack_grads()
        o._flat_g.copy_(((g0 * 0.5).to(torch.bfloat16) + (o._flat_g * 0.5).to(torch.bfloat16)).float())
        snaps.append(("grad", o._flat_g.detach().clone()))
        clip_grad_norm_(m.parameters(), 5.0, optimizer=o)
        o.step(); o.zero_grad()
        snaps.append(("param", o._flat_p.detach().clone()))
    torch.cuda.synchronize()
    all_snaps.append(snaps)
    if run and not all(torch.equal(a[1], b[1]) for a, b in zip(snaps, all_snaps[0])):
        k = next(i for i, (a, b) in enumerate(zip(snaps, all_snaps[0])) if not torch.equal(a[1], b[1]))
        kind, cur = snaps[k]
        d = (cur - all_snaps[0][k][1]).abs()
        offs = [o._offs[o._index_of[id(p_)]] for _, p_ in named]
        rows = sorted(((int((d[off:off + p_.numel()] > 0).sum()), float(d[off:off + p_.numel()].max()), float(all_snaps[0][k][1][off:off + p_.numel()].abs().max()), n)
                       for (n, p_), off in zip(named, offs)), reverse=True)
        print(f"  first difference: step {k // 2} ({seq[k // 2]}), the exchanged {kind} arena; parameters affected:")
        for cnt, mx, sc, n in rows[:10]:
            if cnt:
                print(f"      {cnt:7d} elements, max |d| {mx:.3e} (scale {sc:.3e}): {n}")
    res.append((o._flat_p.detach().clone(), o._flat_m.detach().clone(), losses))
    if run:
        d = (res[run][0] - res[0][0]).abs()
        dm = (res[run][1] - res[0][1]).abs()
        first = next((i for i, (a, b) in enumerate(zip(res[run][2], res[0][2])) if a != b), None)
        print(f"run {run} vs run 0: parameters differ in {int((d > 0).sum())} elements (max {float(d.max()):.2e}), exp_avg max difference "
              f"{float(dm.max()) / float(res[0][1].abs().max()):.2e} of scale, first differing loss index {first}", flush=True)
