#!/usr/bin/env python3
"""How many captured training steps does one process survive?  Captures steps of a tiny model under ever new keys (batch shapes) and
prints progress; `fresh` > 0 rebuilds the GraphedTrainStep (dropping its graphs) every `fresh` captures.
usage: graph_stress.py [captures=1500] [fresh=0] [overlap=0]"""
import gc, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vln_hamt_amd.graph import GraphedTrainStep
from vln_hamt_amd.model.pretrain_cmt import MultiStepNavCMTPreTraining
from vln_hamt_amd.modeling import HamtConfig
from vln_hamt_amd.optim import AdamW
from vln_hamt_amd.synth import make_batch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
fresh = int(sys.argv[2]) if len(sys.argv) > 2 else 0
mode = sys.argv[3] if len(sys.argv) > 3 else "0"      # 1: update at the head of the next replay from the start; 2: for the last 50 captures only
overlap = mode == "1"
dev = torch.device("cuda")
cfg = HamtConfig(hamt_precision="bf16", pretrain_tasks={"mlm", "sap", "sar", "sprel", "mrc", "itm"}, hidden_size=128, num_attention_heads=2,
                 intermediate_size=256, image_feat_size=64, num_l_layers=2, num_x_layers=1, num_h_pano_layers=1)
torch.manual_seed(0)
model = MultiStepNavCMTPreTraining(cfg).to(dev)
model.train()
opt = AdamW([{"params": list(model.parameters()), "weight_decay": 0.0}], lr=1e-4, betas=(0.9, 0.98))
gs = GraphedTrainStep(model, opt, 5.0, overlap_update=overlap)
for i in range(n):
    if mode == "2" and i == n - 50:
        del gs
        gc.collect(); torch.cuda.synchronize(); torch.cuda.empty_cache()
        overlap = True
        gs = GraphedTrainStep(model, opt, 5.0, overlap_update=True)
    if fresh and i and i % fresh == 0:
        if overlap:
            gs.finish()
        del gs
        gc.collect(); torch.cuda.synchronize(); torch.cuda.empty_cache()
        gs = GraphedTrainStep(model, opt, 5.0, overlap_update=overlap)
    task = ("sap", "sar", "mrc")[i % 3]
    b = make_batch(task, 2 + i % 5, cfg, seed=i, txt_len=8 + (i // 5) % 40, hist_len=1 + (i // 200) % 4, device=dev)
    key = GraphedTrainStep.key_for(task, b) + (i,)
    loss = float(gs.step(key, b, task))
    loss = float(gs.step(key, b, task))
    if i % 50 == 0:
        print(f"capture {i}: loss {loss:.4f}, graphs alive {len(gs.graphs)}, allocated {torch.cuda.memory_allocated() / 2**20:.0f} MiB", flush=True)
print("survived", n, "captures")
