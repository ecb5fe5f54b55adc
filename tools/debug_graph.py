import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from _util import tiny_cfg
from oracle.hamt_oracle import make_state_dict, pretrain_param_shapes
from test_gpu_model import build
from vln_hamt_amd.graph import GraphedTrainStep
from vln_hamt_amd.optim import AdamW, clip_grad_norm_
from vln_hamt_amd.optim.misc import NO_DECAY
from vln_hamt_amd.synth import make_batch, make_itm_rng
DEV = "cuda"
cfg = tiny_cfg()
sd = make_state_dict(pretrain_param_shapes(cfg), seed=7)

def make():
    m = build(cfg, sd, "bf16", train=True)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    named = list(m.named_parameters())
    groups = [{'params': [p for n, p in named if not any(nd in n for nd in NO_DECAY)], 'weight_decay': 0.01},
              {'params': [p for n, p in named if any(nd in n for nd in NO_DECAY)], 'weight_decay': 0.0}]
    return m, AdamW(groups, lr=1e-3, betas=(0.9, 0.98), eps=1.0)

seq = ["sap", "mlm", "itm", "sap", "mlm", "itm", "mrc", "sap", "mrc"]
batches = {}
for t in set(seq):
    b = make_batch(t, 4, cfg, seed=sum(map(ord, t)), txt_len=20, hist_len=4, ragged=True, device=DEV)
    if t == "itm":
        r = make_itm_rng(b, seed=3)
        b["itm_neg_idxs"], b["itm_shuffled_pos_ids"] = r["neg_idxs"], r["shuffled_pos_ids"]
    batches[t] = b

def eager_run():
    m, o = make()
    snaps = []
    for t in seq:
        m(batches[t], t, True).mean().backward()
        clip_grad_norm_(m.parameters(), 5.0, optimizer=o)
        o.step(); o.zero_grad()
        snaps.append({k: p.detach().clone() for k, p in m.named_parameters()})
    return snaps

def graph_run():
    m, o = make()
    gs = GraphedTrainStep(m, o, 5.0)
    snaps = []
    for t in seq:
        gs.step(t, batches[t], t)
        torch.cuda.synchronize()
        snaps.append({k: p.detach().clone() for k, p in m.named_parameters()})
    return snaps

a, b, c = eager_run(), eager_run(), graph_run()
for i, t in enumerate(seq):
    d_ee = max(float((a[i][k] - b[i][k]).abs().max()) for k in a[i])
    worst = max(((float((a[i][k] - c[i][k]).abs().max()), k) for k in a[i]))
    print(f"step {i} {t}: eager-vs-eager {d_ee:.2e}   eager-vs-graph {worst[0]:.2e} at {worst[1]}")
