#!/usr/bin/env python3
"""Which torch (aten) ops still launch kernels inside a training step, and from which lines of this repository: one eager step per
task under a TorchDispatchMode that records every aten call touching a CUDA tensor with the innermost repository frame.
usage: aten_sources.py [batch] [task ...]"""
import collections, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from bench import build_model
from vln_hamt_amd.optim import AdamW, clip_grad_norm_
from vln_hamt_amd.synth import make_batch, make_itm_rng
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VIEW = ("view", "reshape", "expand", "slice", "select", "aten.t.default", "transpose", "unsqueeze", "squeeze", "detach", "alias", "as_strided",
        "_unsafe_view", "permute", "empty", "_local_scalar", "is_", "size", "stride", "record_stream", "narrow", "unbind", "split", "chunk", "lift", "set_")


class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.agg = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        flat = [a for a in list(args) + list((kwargs or {}).values()) if torch.is_tensor(a)]
        for a in args:
            if isinstance(a, (list, tuple)):
                flat += [t for t in a if torch.is_tensor(t)]
        outs = [out] if torch.is_tensor(out) else [t for t in (out if isinstance(out, (list, tuple)) else []) if torch.is_tensor(t)]
        if any(t.is_cuda for t in flat + outs) and not any(v in name for v in VIEW):
            fr = [f for f in traceback.extract_stack() if ("vln-hamt_amd" in f.filename or "vln_hamt_amd" in f.filename) and "aten_sources" not in f.filename]
            where = f"{os.path.relpath(fr[-1].filename, ROOT)}:{fr[-1].lineno} {fr[-1].name}" if fr else "(autograd engine: gradient accumulation / materialisation)"
            self.agg[(name.replace("aten.", ""), where)] += 1
        return out


dev = torch.device("cuda", 0)
model, cfg = build_model("bf16", dev)
opt = AdamW([{"params": list(model.parameters()), "weight_decay": 0.01}], lr=5e-5).materialize()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
for task in (sys.argv[2:] or ["sap", "mlm", "itm"]):
    b = make_batch(task, B, cfg, seed=1, txt_len=80, hist_len=5, mlm_exact=12 if task == "mlm" else None, device=dev)
    if task == "itm":
        r = make_itm_rng(b, seed=1)
        b["itm_neg_idxs"], b["itm_shuffled_pos_ids"] = r["neg_idxs"], r["shuffled_pos_ids"]

    def run():
        model(b, task, True).mean().backward()
        clip_grad_norm_(model.parameters(), 5.0, optimizer=opt)
        opt.step()
        opt.zero_grad()
    run()
    torch.cuda.synchronize()
    with Log() as lg:
        run()
    torch.cuda.synchronize()
    print(f"== {task} (B={B}): {sum(lg.agg.values())} aten calls on CUDA tensors")
    for (name, where), n in lg.agg.most_common(45):
        print(f"{n:4d}  {name:32s} {where}")
