"""Live per-kernel accounting of one optimisation step for bench.py's `roofline` object.

rocprofv3 --kernel-trace tells which kernel (template instantiation) owns the largest share of a step; this module measures
the same thing inside the bench process, with HIP events on the launch stream, so that the bench line can name the dominant
kernel and its achieved rate without a profiler attached:

  * every hamt_gemm call of one forward + backward pass per task is recorded together with the name of the kernel the launcher
    picked for it (hamt_last_kernel), grouped by that name, and each group is re-issued as a captured hipGraph (same operands,
    same epilogues; eager re-issue would measure the host's launch rate for the small ones);
  * the grouped weight-gradient launch is re-issued from the problem list the pass queued;
  * the optimizer's update kernel is timed on the real arenas.

Per-step time of a kernel = sum over tasks of (task frequency in the 5:1:1:1:2:2 mix) x (time of its launches in that task's
step).  Algorithmic work: 2 M N K per GEMM launch; 30 bytes per parameter (+ 4 where the gradient slot is zeroed) for the AdamW table kernel (SURVEY 8d / DESIGN 4).
"""
from __future__ import annotations

import collections
import ctypes as C
import json
import os

import torch

PEAK_BF16_TFLOPS = 2500.0
PEAK_HBM_GBS = 8000.0
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _graph_time(fn, stream, reps=3):
    """seconds per call of fn() when replayed from a captured graph on `stream`"""
    stream.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(stream):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            fn()
        g.replay()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(stream)
        for _ in range(reps):
            g.replay()
        e.record(stream)
        torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e-3 / reps


def record_step(model, batch, task):
    """One forward + backward of `task`: ([(kernel, a, b, out, kwargs, flops)], [wgrad items])."""
    from vln_hamt_amd import _lib as Lb, blocks, blocks_preln, ops, wgrad
    calls, items = [], []
    orig = ops.gemm

    def rec(a, bb, out, **kw):
        r = orig(a, bb, out, **kw)
        if a.dtype == torch.bfloat16 and bb.dtype == torch.bfloat16:
            K = kw.get("k_red") or (a.shape[0] if kw.get("a_kmajor") else a.shape[1])
            calls.append((Lb.last_kernel(), a, bb, out, dict(kw), 2.0 * out.shape[0] * out.shape[1] * K))
        return r

    mods = [m for m in (ops, blocks, blocks_preln) if getattr(m, "gemm", None) is orig]
    prev = wgrad.get_handler()
    for m in mods:
        m.gemm = rec
    wgrad.set_handler(items.extend)
    try:
        model(batch, task, True).mean().backward()
    finally:
        for m in mods:
            m.gemm = orig
        wgrad.set_handler(prev)
    for p_ in model.parameters():
        p_.grad = None
    torch.cuda.synchronize()
    return calls, items


def _wgrad_launcher(items, device):
    from vln_hamt_amd import _lib as Lb, ops, wgrad
    lib = Lb.load()
    items = [it for it in items if it[2].stride(0) >= 256 and it[3].stride(0) >= 256 and it[2].shape[0] >= 128]   # the 256-square-tile class
    n = len(items)
    descs = (Lb.WgradDesc * max(1, n))()
    keep, flops = [], 0.0
    for i, (w, bb, dy16, x16) in enumerate(items):
        dw = torch.empty(w.shape, dtype=torch.float32, device=device)
        db = torch.empty(w.shape[0], dtype=torch.float32, device=device)
        keep += [dw, db, dy16, x16]
        d = descs[i]
        d.dy, d.x, d.dw, d.db = dy16.data_ptr(), x16.data_ptr(), dw.data_ptr(), (db.data_ptr() if bb is not None else None)
        d.M, d.N, d.K, d.ldy, d.ldx, d.ldw, d.accum_dw, d.accum_db = w.shape[0], w.shape[1], dy16.shape[0], dy16.stride(0), x16.stride(0), w.shape[1], 0, 0
        d.K_valid = wgrad.valid_rows(dy16)
        flops += 2.0 * w.shape[0] * w.shape[1] * dy16.shape[0]
    tab = torch.empty(max(1, wgrad.table_entries(descs, n)) * Lb.WGRAD_TABLE_ENTRY, dtype=torch.uint8, device=device)
    keep.append(tab)

    def run():
        if n:
            Lb.check(lib.hamt_wgrad_grouped(n, descs, tab.data_ptr(), tab.numel(), ops._stream()), "hamt_wgrad_grouped")
    return run, flops, n, keep


def kernel_table(model, opt, cycle, device, live=True):
    """[{kernel, bound, per_step_ms, launches_per_step, avg_launch_us, achieved, unit, frac}] sorted by per-step time.
    `cycle` = [(task, batch)] of one task-mix cycle (the batches of the timed region).  `live`: time the grouped weight-gradient
    kernel and the update kernel inside one eager step per task (single-process runs only: with a gradient exchange installed a
    step on ONE rank would start collectives nobody answers); else their back-to-back figures stand in."""
    from vln_hamt_amd import ops
    st = torch.cuda.Stream()
    freq = collections.Counter(t for t, _ in cycle)
    ncyc = float(len(cycle))
    acc = collections.defaultdict(lambda: {"s": 0.0, "launches": 0.0, "work": 0.0})
    adam = {"s": 0.0, "bytes": 0.0}
    seen = set()
    for task, b in cycle:
        if task in seen:
            continue
        seen.add(task)
        wt = freq[task] / ncyc
        calls, items = record_step(model, b, task)
        groups = collections.defaultdict(list)
        for c in calls:
            groups[c[0]].append(c)
        for name, cs in groups.items():
            def run(cs=cs):
                for _, a, bb, out, kw, _f in cs:
                    ops.gemm(a, bb, out, **kw)
            dt = _graph_time(run, st)
            a_ = acc[name]
            a_["s"] += wt * dt
            a_["launches"] += wt * len(cs)
            a_["work"] += wt * sum(c[5] for c in cs)
        run, flops, n, keep = _wgrad_launcher(items, device)
        if n:
            # per KERNEL launch (rocprofv3's unit: a call with more than 48 table entries is two launches), HIP events on the launch
            # stream around each one (hamt_debug_wgrad_timing); eager, back to back, the table writes outside the brackets
            import ctypes
            from vln_hamt_amd import _lib as Lb
            lib, reps = Lb.load(), 3
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                run()
                torch.cuda.synchronize()
                lib.hamt_debug_wgrad_timing(1)
                for _ in range(reps):
                    run()
                torch.cuda.synchronize()
            us, trows, fl = (ctypes.c_float * 64)(), (ctypes.c_int * 64)(), (ctypes.c_double * 64)()
            k = lib.hamt_debug_wgrad_times(us, trows, fl, 64)
            lib.hamt_debug_wgrad_timing(0)
            assert 0 < k <= 64, k
            big = [us[i] for i in range(k) if trows[i] == 256]
            big_fl = sum(fl[i] for i in range(k) if trows[i] == 256) / reps
            # ... and the same kernel where it runs: one eager forward + backward of the task with the pass's own flush (gradient
            # arena targets, tile sums of squares, the pass's launch groups), bracketed the same way.  THIS is the row's time --
            # the launches rocprofv3 sees in the step; the back-to-back figure above is kept beside it.
            a_ = acc["wgrad_grouped_p8_kernel" if os.environ.get("HAMT_WGRAD_P8", "1") != "0" else "wgrad_grouped_kernel<256, 256, 2, 4>"]
            if not live:
                a_["s"] += wt * sum(big) * 1e-6 / reps
                a_["launches"] += wt * len(big) / reps
                a_["work"] += wt * big_fl
                del calls, items, groups, keep
                continue
            lib.hamt_debug_wgrad_timing(1)
            model(b, task, True).mean().backward()
            torch.cuda.synchronize()
            k2 = lib.hamt_debug_wgrad_times(us, trows, fl, 64)
            lib.hamt_debug_wgrad_timing(0)
            # ... and the update that follows it in the step (this task's gradients: its own active set, the gradient arena as
            # the weight-gradient launch left it), torch events on the launch stream around the one kernel of launch_step
            from vln_hamt_amd.optim import clip_grad_norm_
            clip_grad_norm_(model.parameters(), 5.0, optimizer=opt)
            if not opt._packed:
                opt._pack_grads()
            opt._ensure_table()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            nbytes_live = opt.update_bytes(opt._active)
            e0.record()
            opt.launch_step()
            e1.record()
            opt.mark_updated()
            torch.cuda.synchronize()
            adam["s"] += wt * e0.elapsed_time(e1) * 1e-3
            adam["bytes"] += wt * nbytes_live
            for p_ in model.parameters():
                p_.grad = None
            assert 0 < k2 <= 64, k2
            live = [us[i] for i in range(k2) if trows[i] == 256]
            a_ = acc["wgrad_grouped_p8_kernel" if os.environ.get("HAMT_WGRAD_P8", "1") != "0" else "wgrad_grouped_kernel<256, 256, 2, 4>"]
            a_["s"] += wt * sum(live) * 1e-6
            a_["launches"] += wt * len(live)
            a_["work"] += wt * sum(fl[i] for i in range(k2) if trows[i] == 256)     # what these launches multiplied (the library's own count)
            a_["b2b"] = a_.get("b2b", 0.0) + wt * sum(big) * 1e-6 / reps
            a_["b2b_work"] = a_.get("b2b_work", 0.0) + wt * big_fl
        del calls, items, groups, keep
    rows = []
    for name, a_ in acc.items():
        tf = a_["work"] / a_["s"] / 1e12
        rows.append({"kernel": name, "bound": "mfma", "per_step_ms": round(a_["s"] * 1e3, 3), "launches_per_step": round(a_["launches"], 1),
                     "avg_launch_us": round(a_["s"] * 1e6 / a_["launches"], 2), "achieved": round(tf, 1), "unit": "TFLOP/s",
                     "peak": PEAK_BF16_TFLOPS, "frac": round(tf / PEAK_BF16_TFLOPS, 4), "work_per_launch": a_["work"] / a_["launches"]})
        if "b2b" in a_:       # the grouped weight-gradient kernel: in-step (above) and re-issued back to back on fresh targets
            rows[-1]["back_to_back_ms_per_step"] = round(a_["b2b"] * 1e3, 3)
            rows[-1]["back_to_back_achieved"] = round(a_["b2b_work"] / a_["b2b"] / 1e12, 1)
    # the optimizer's update kernel: in-step above (each task's own update, weighted by the mix); beside it the same kernel
    # re-issued back to back from a captured graph with every parameter active (no gradient left in the Infinity Cache by
    # the weight-gradient launch in front of it: the upper bound)
    opt._packed = True
    opt.prepare_step([True] * len(opt._params))
    dt = _graph_time(lambda: opt.launch_step(), st)
    nbytes = opt.update_bytes()       # 30 B per element (+ 4 where the gradient slot is zeroed too)
    if adam["s"] > 0:
        dl, bl = adam["s"], adam["bytes"]
        rows.append({"kernel": "adamw_table_kernel<4>", "bound": "hbm", "per_step_ms": round(dl * 1e3, 3), "launches_per_step": 1.0,
                     "avg_launch_us": round(dl * 1e6, 2), "achieved": round(bl / dl / 1e9, 1), "unit": "GB/s", "peak": PEAK_HBM_GBS,
                     "frac": round(bl / dl / 1e9 / PEAK_HBM_GBS, 4), "work_per_launch": bl,
                     "back_to_back_ms_per_step": round(dt * 1e3, 3), "back_to_back_achieved": round(nbytes / dt / 1e9, 1)})
    else:
        rows.append({"kernel": "adamw_table_kernel<4>", "bound": "hbm", "per_step_ms": round(dt * 1e3, 3), "launches_per_step": 1.0,
                     "avg_launch_us": round(dt * 1e6, 2), "achieved": round(nbytes / dt / 1e9, 1), "unit": "GB/s", "peak": PEAK_HBM_GBS,
                     "frac": round(nbytes / dt / 1e9 / PEAK_HBM_GBS, 4), "work_per_launch": nbytes})
    opt._packed = False
    rows.sort(key=lambda r: -r["per_step_ms"])
    return rows


def traffic_of(kernel: str, batch: int):
    """HBM bytes per launch of `kernel` from the committed PMC summary (profiles/rNN_traffic_b<batch>.json -- the newest round's --, written by
    tools/hbm_rates.py --json from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes; FETCH_SIZE x 2 per the gfx950
    correction of MI355X_MICROARCH.md), or None when that kernel / batch was not collected."""
    path = next((q for q in (os.path.join(ROOT, "profiles", f"r{r:02d}_traffic_b{batch}.json") for r in (9, 8, 7, 6, 5, 4, 3, 2)) if os.path.exists(q)), None)
    if path is None:
        return None, None
    tab = json.load(open(path))
    key = kernel.replace(" ", "")
    for k, v in tab.items():
        if k.replace(" ", "") == key:
            return int(v["read_bytes"] + v["write_bytes"]), os.path.relpath(path, ROOT)
    return None, os.path.relpath(path, ROOT)


def subblock_xattn(model, batch, device, L_txt=80, n_vis=43):
    """The graded cross-modal attention SUB-BLOCK: LayerNorm'd inputs -> Q / K,V projections -> masked
    softmax attention -> output projection -> dropout + residual + LayerNorm, forward AND backward (incl. its weight gradients)
    of blocks.CrossAttnBlockFn with the first x-layer's weights at the x-layers' own shapes, both directions.
    Algorithmic FLOPs = 3 x (4 Sq H^2 + 4 Sk H^2 + 4 Sq Sk H) per direction per sample."""
    from vln_hamt_amd import blocks
    xl = model.bert.encoder.x_layers[0]
    att, out = xl.visual_attention.att, xl.visual_attention.output
    Hd = att.query.weight.shape[0]
    st = torch.cuda.Stream()
    res, tot_s, tot_f = {}, 0.0, 0.0
    for name, Sq, Sk in (("text<-vision", L_txt, n_vis), ("vision<-text", n_vis, L_txt)):
        x = torch.randn(batch, Sq, Hd, device=device, requires_grad=True)
        c = torch.randn(batch, Sk, Hd, device=device, requires_grad=True)
        mask = torch.zeros(batch, 1, 1, Sk, device=device)
        dy = torch.randn(batch, Sq, Hd, device=device)

        def step():
            y = blocks.cross_attn_block(x, c, mask, att, out, True)
            y.backward(dy)
            x.grad = None
            c.grad = None
        step()
        for p_ in model.parameters():
            p_.grad = None
        dt = _graph_time(step, st)
        for p_ in model.parameters():
            p_.grad = None
        fl = 3.0 * batch * (4.0 * Sq * Hd * Hd + 4.0 * Sk * Hd * Hd + 4.0 * Sq * Sk * Hd)
        res[name] = {"Sq": Sq, "Sk": Sk, "fwd_bwd_us": round(dt * 1e6, 1), "tflops": round(fl / dt / 1e12, 1)}
        tot_s += dt
        tot_f += fl
    tf = tot_f / tot_s / 1e12
    return {"block": f"CrossAttnBlockFn forward + backward (LN-ed inputs, Q / KV projections, masked softmax attention, output projection, dropout + "
                     f"residual + LayerNorm, all weight gradients), both directions, B={batch}, 12 heads x 64, bf16, dropout 0.1",
            "flops_formula": "3 * (4 Sq H^2 + 4 Sk H^2 + 4 Sq Sk H) per direction per sample (SURVEY 8d)", "bound": "mfma",
            "achieved": round(tf, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / PEAK_BF16_TFLOPS, 4), "cases": res}
