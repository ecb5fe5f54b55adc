#!/usr/bin/env python3
"""Row N4 measurement: host collation + PCIe transport of R2R pretrain batches (B per GPU, 768-d features, T <= 5, 36
views, L <= 80), and the PCIe-INCLUSIVE training step.

  1. reference-style: the host loops of the reference's collate (zero fill + one slice copy per sample and field,
     data/common.py:5-20, restated below with torch ops) -> pin every tensor -> one H2D copy per tensor
  2. this repo: pack ragged rows into one buffer -> pin -> ONE H2D copy -> hamt_unpack_padded / hamt_seq_masks
  3. SAP training steps (hipGraph replay) fed by PrefetchLoader: every step's batch is collated on the host, crosses
     PCIe and is unpacked straight into the graph's static inputs, overlapped with the previous step
usage: collate_bench.py [batch=64] [steps=40]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vln_hamt_amd.data import COLLATE, PrefetchLoader, move_to_cuda
from vln_hamt_amd.synth import make_samples

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device("cuda", 0)
torch.set_num_threads(8)
pools = {t: make_samples(t, 4 * B, seed=9, max_hist=5) for t in ("sap", "mlm")}
draw = lambda t, i: [pools[t][(i * B + j) % (4 * B)] for j in range(B)]

def timeit(fn, n=12):
    fn(0); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n): fn(i + 1)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n

def _pad(ts, lens=None, pad=0):
    lens = [t.shape[0] for t in ts] if lens is None else lens
    out = torch.full((len(ts), max(lens)) + tuple(ts[0].shape[1:]), pad, dtype=ts[0].dtype)
    for i, (t, l) in enumerate(zip(ts, lens)):
        out[i, :l] = t
    return out

def ref_collate(task, inputs):
    """what the reference's *_collate functions do on the host, for the tensor fields (timing leg only)"""
    b = {k: [x[k] for x in inputs] for k in inputs[0]}
    out = {"txt_ids": _pad(b["txt_ids"]), "txt_masks": torch.arange(max(b["txt_lens"]))[None] < torch.tensor(b["txt_lens"])[:, None]}
    if "txt_labels" in b:
        out["txt_labels"] = _pad(b["txt_labels"], pad=-1)
    for k in ("hist_img_fts", "hist_ang_fts", "hist_pano_img_fts", "hist_pano_ang_fts"):
        out[k] = _pad(b[k], b["hist_lens"])
    hl = torch.tensor(b["hist_lens"]) + 1
    out["hist_masks"] = torch.arange(int(hl.max()))[None] < hl[:, None]
    if "ob_img_fts" in b:
        for k in ("ob_img_fts", "ob_ang_fts"):
            out[k] = _pad(b[k], b["ob_lens"])
        out["ob_nav_types"] = _pad(b["ob_nav_types"])
        out["ob_masks"] = torch.arange(max(b["ob_lens"]))[None] < torch.tensor(b["ob_lens"])[:, None]
    return out

ORACLE = {t: (lambda inputs, t=t: {k: v.numpy() for k, v in ref_collate(t, inputs).items()}) for t in ("sap", "mlm")}
for task in ("sap", "mlm"):
    def ref_style(i):
        b = ORACLE[task](draw(task, i))
        return {k: (torch.from_numpy(np.ascontiguousarray(v)).pin_memory().to(dev, non_blocking=True) if isinstance(v, np.ndarray) else v) for k, v in b.items()}
    def ref_host_only(i):
        return ORACLE[task](draw(task, i))
    def packed(i):
        return COLLATE[task](draw(task, i)).pin_memory().to_device(dev)
    def packed_host_only(i):
        return COLLATE[task](draw(task, i))
    pb = COLLATE[task](draw(task, 0))
    padded = sum(v.nbytes for v in ORACLE[task](draw(task, 0)).values() if isinstance(v, np.ndarray))
    a, b_, c, d = timeit(ref_host_only), timeit(ref_style), timeit(packed_host_only), timeit(packed)
    print(f"{task} B={B}: padded batch {padded/2**20:.1f} MiB, packed buffer {pb.nbytes/2**20:.1f} MiB")
    print(f"  reference-style  host collate {a*1e3:6.2f} ms, + pin + H2D per tensor {b_*1e3:6.2f} ms  = {B/b_:8.0f} samples/s")
    print(f"  packed transport host pack    {c*1e3:6.2f} ms, + pin + one H2D + device unpack {d*1e3:6.2f} ms  = {B/d:8.0f} samples/s")

# ---- PCIe-inclusive SAP step
from vln_hamt_amd import ops
from vln_hamt_amd.graph import GraphedTrainStep
from vln_hamt_amd.model.pretrain_cmt import MultiStepNavCMTPreTraining
from vln_hamt_amd.modeling import HamtConfig
from vln_hamt_amd.optim import AdamW
from vln_hamt_amd.optim.misc import NO_DECAY
ops.manual_seed(3, dev)
cfg = HamtConfig(hamt_precision="bf16", pretrain_tasks={"mlm", "sap", "sar", "sprel", "mrc", "itm"})
torch.manual_seed(0)
model = MultiStepNavCMTPreTraining(cfg).to(dev).train()
named = list(model.named_parameters())
opt = AdamW([{"params": [p for n, p in named if not any(nd in n for nd in NO_DECAY)], "weight_decay": 0.01},
             {"params": [p for n, p in named if any(nd in n for nd in NO_DECAY)], "weight_decay": 0.0}], lr=5e-5, betas=(0.9, 0.98))
opt.materialize()
# fixed shapes for the captured graph: every sample full length (the bench's shape); the DATA changes every step
full = make_samples("sap", 4 * B, seed=10, max_hist=5)
rng = np.random.Generator(np.random.PCG64(1))
def full_len(s):
    s = dict(s)
    L, T, V = 80, 5, 37
    s["txt_ids"], s["txt_lens"] = torch.from_numpy(rng.integers(1000, 29000, size=L)), L
    for k, shp in (("ob_img_fts", (V, 768)), ("ob_ang_fts", (V, 4)), ("hist_img_fts", (T, 768)), ("hist_ang_fts", (T, 4)),
                   ("hist_pano_img_fts", (T, 36, 768)), ("hist_pano_ang_fts", (T, 36, 4))):
        s[k] = torch.from_numpy(rng.standard_normal(shp, dtype=np.float32))
    s["ob_nav_types"] = torch.from_numpy(np.concatenate([rng.integers(0, 2, size=V - 1), [2]]))
    s["ob_nav_types"][0] = 1
    s["ob_lens"], s["hist_lens"], s["ob_action_viewindex"] = V, T, 0
    return s
full = [full_len(s) for s in full]

class Feed(torch.utils.data.Dataset):
    def __len__(self): return (steps + 16) * B
    def __getitem__(self, i): return full[i % len(full)]
static = None
graphed = GraphedTrainStep(model, opt, max_grad_norm=5.0)

class StaticPrefetch(PrefetchLoader):          # unpack straight into the graph's static inputs
    def preload(self, it):
        try:
            pb = next(it)
        except StopIteration:
            self.batch = None
            return
        with torch.cuda.stream(self.stream):
            self.stream.wait_stream(torch.cuda.current_stream())       # the previous replay has consumed the static inputs
            self.batch = move_to_cuda(pb, self.device, out=static)

for workers in (0, 4):
    dl = torch.utils.data.DataLoader(Feed(), batch_size=B, shuffle=False, collate_fn=COLLATE["sap"], pin_memory=True, num_workers=workers,
                                     persistent_workers=workers > 0, prefetch_factor=4 if workers else None)
    it = iter(StaticPrefetch(dl, dev))
    t0 = None
    for s in range(steps + 8):
        if s == 8:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        batch = next(it)
        if static is None:
            static = {k: v for k, v in batch.items() if torch.is_tensor(v)}
        graphed.step("sap", batch, "sap")
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"PCIe-inclusive SAP step, B={B}, DataLoader workers={workers}: {dt*1e3:.2f} ms/step = {B/dt:.0f} panorama-steps/s")
# resident-input reference point
b0 = {k: v for k, v in static.items()}
torch.cuda.synchronize(); t0 = time.perf_counter()
for s in range(steps):
    graphed.step("sap", b0, "sap")
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"inputs resident in HBM (bench.py's convention), same graph: {dt*1e3:.2f} ms/step = {B/dt:.0f} panorama-steps/s")
