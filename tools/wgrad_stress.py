#!/usr/bin/env python3
"""Stress of the grouped weight-gradient launch on the REAL problem list of a backward pass (B = 32 by default, task DBG_TASK): the
launcher's own tile choice (256-square two-phase tiles) against the 128-row tiles, REPS times, with the LDS of every CU filled with
bf16 NaNs in front of every launch (POISON=0 to skip) and GEMMs running on a second stream; ACCUM=1 for accumulate semantics."""
import os, sys
sys.path.insert(0, "/root/repo")
import torch
import bench
from vln_hamt_amd import _lib as L, ops, wgrad
from vln_hamt_amd.synth import make_batch, make_itm_rng
dev = torch.device("cuda")
ops.manual_seed(1, dev)
model, cfg = bench.build_model("bf16", dev)
B = int(os.environ.get("DBG_B", "32"))
task = os.environ.get("DBG_TASK", "sar")
b = make_batch(task, B, cfg, seed=106, txt_len=80, hist_len=5, mlm_exact=12 if task == "mlm" else None, device=dev)
items = []
wgrad.set_handler(items.extend)
model(b, task, True).mean().backward()
wgrad.set_handler(None)
torch.cuda.synchronize()
print("problems", len(items), "K values", sorted({it[2].shape[0] for it in items}))
lib = L.load()
def run(tile=None):
    if tile: os.environ["HAMT_WGRAD_TILE"] = tile
    else: os.environ.pop("HAMT_WGRAD_TILE", None)
    n = len(items)
    descs = (L.WgradDesc * n)()
    outs = []
    for i, (w, bb, dy16, x16) in enumerate(items):
        acc = os.environ.get("ACCUM") == "1"
        dw = torch.full(w.shape, 0.25 if acc else float("nan"), dtype=torch.float32, device=dev)
        db = torch.full((w.shape[0],), 0.5 if acc else float("nan"), dtype=torch.float32, device=dev) if bb is not None else None
        outs.append((dw, db))
        d = descs[i]
        d.dy, d.x, d.dw, d.db = dy16.data_ptr(), x16.data_ptr(), dw.data_ptr(), (db.data_ptr() if db is not None else None)
        d.M, d.N, d.K, d.ldy, d.ldx, d.ldw, d.accum_dw, d.accum_db = w.shape[0], w.shape[1], dy16.shape[0], dy16.stride(0), x16.stride(0), w.shape[1], int(acc), int(acc)
        d.K_valid = wgrad.valid_rows(dy16)
    tab = torch.empty(max(1, wgrad.table_entries(descs, n)) * L.WGRAD_TABLE_ENTRY, dtype=torch.uint8, device=dev)
    L.check(lib.hamt_wgrad_grouped(n, descs, tab.data_ptr(), tab.numel(), ops._stream()), "wgrad")
    return outs
ref = run("128")
torch.cuda.synchronize()
side = torch.cuda.Stream()
xa = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
nbad = 0
for rep in range(int(os.environ.get("REPS", "60"))):
    with torch.cuda.stream(side):
        for _ in range(3):
            xa @ xa
    if os.environ.get('POISON', '1') == '1':
        L.check(lib.hamt_debug_fill_lds(0x7FC07FC0, ops._stream()), 'fill')
    out = run()
    torch.cuda.synchronize()
    for i, ((dw, db), (rw, rb)) in enumerate(zip(out, ref)):
        w = items[i][0]
        bad = not torch.isfinite(dw).all() or float((dw - rw).abs().max()) > 1e-3 * max(1e-20, float(rw.abs().max()))
        badb = db is not None and (not torch.isfinite(db).all() or float((db - rb).abs().max()) > 1e-3 * max(1e-20, float(rb.abs().max())))
        if bad or badb:
            nbad += 1
            if nbad <= 12:
                nanw = int((~torch.isfinite(dw)).sum())
                print(f"rep {rep} problem {i}: W {tuple(w.shape)} K={items[i][2].shape[0]} ldy={items[i][2].stride(0)} ldx={items[i][3].stride(0)} bad_dw={bad} bad_db={badb} "
                      f"nonfinite={nanw} maxdiff={float((torch.nan_to_num(dw) - rw).abs().max()):.3e} ref max {float(rw.abs().max()):.3e}")
print("mismatching problem-launches:", nbad)
