import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from _util import tiny_cfg
from oracle.hamt_oracle import make_state_dict, pretrain_param_shapes
from test_gpu_model import build
from vln_hamt_amd.optim import AdamW
from vln_hamt_amd.synth import make_batch
DEV = "cuda"
cfg = tiny_cfg()
sd = make_state_dict(pretrain_param_shapes(cfg), seed=7)
task = sys.argv[1] if len(sys.argv) > 1 else "sap"
m = build(cfg, sd, "bf16", train=True)
for mod in m.modules():
    if isinstance(mod, torch.nn.Dropout):
        mod.p = 0.0
opt = AdamW(m.parameters(), lr=1e-3)
opt.materialize()
b = make_batch(task, 4, cfg, seed=5, txt_len=20, hist_len=4, ragged=True, device=DEV)

def run():
    loss = m(b, task, True).mean()
    loss.backward()
    return loss

loss_e = run().detach()
ge = {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}
m.zero_grad(set_to_none=True)
# poison the allocator's free blocks so that stale reads show up
junk = [torch.full((1 << 20,), float("nan"), device=DEV) for _ in range(64)]
del junk
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    run().detach(); m.zero_grad(set_to_none=True)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    loss_g = run().detach()
gg_t = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
for rep in range(3):
    # poison pool-adjacent memory between replays
    g.replay()
    torch.cuda.synchronize()
    worst = (0.0, None)
    nan = []
    for k in ge:
        d = (gg_t[k] - ge[k]).abs().max().item()
        if d != d:
            nan.append(k)
        elif d > worst[0]:
            worst = (d, k)
    print(f"replay {rep}: loss eager {loss_e.item():.6f} graph {loss_g.item():.6f}  worst grad diff {worst[0]:.3e} at {worst[1]}  nan params: {len(nan)} {nan[:3]}")
