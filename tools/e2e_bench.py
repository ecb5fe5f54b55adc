#!/usr/bin/env python3
"""Image-input pretrain step (BASELINE config 4 "end-to-end pretrain", one GPU): raw 224x224 views -> ViT-B/16 backbone
(T*36 panorama views no-grad, T history views + 36 observation views with gradient) -> HAMT trunk -> loss -> backward
-> clip 5.0 -> flat AdamW over all 261 M parameters.  The reference runs this at train_batch_size 1, max_txt_len 60
(pretrain_r2r_e2e.json) with the 5:1:1:1:2:2 task mix; its Ralamb+Lookahead optimiser is outside the hot-path scope,
AdamW stands in.  usage: e2e_bench.py [batch=2] [steps=12] [graph]   (graph: whole-step hipGraph replay per task)
bench.py imports `run` for its `e2e_image_step` key."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

T, V, L = 5, 36, 60
H, FFN = 768, 3072
VIT_FWD_GF = 35.1            # ViT-B/16 forward per 224 x 224 view


def _trunk_fwd(task):
    layer = lambda S: 24 * S * H * H + 4 * S * S * H
    xl = lambda Lq, Vn: 32 * (Lq + Vn) * H * H + 8 * Lq * Vn * H + 4 * Lq * Lq * H + 4 * Vn * Vn * H
    ob = task in ("sap", "sar", "sprel")
    f = 9 * layer(L) + T * (2 * layer(V) + 2 * V * H * H) + T * 2 * H * H + (ob * (V + 1) * 2 * H * H)
    return f + (5 if task == "itm" else 1) * 4 * xl(L, T + 1 + (V + 1 if ob else 0))


def run(B=1, steps=12, use_graph=True, dev=None, verbose=False):
    from vln_hamt_amd import ops
    from vln_hamt_amd.model.image_pretrain import MultiStepNavImagePreTraining
    from vln_hamt_amd.modeling import HamtConfig
    from vln_hamt_amd.optim import AdamW, clip_grad_norm_
    from vln_hamt_amd.optim.misc import NO_DECAY
    from vln_hamt_amd.parallel import TaskSchedule
    from vln_hamt_amd.synth import make_batch, make_itm_rng
    dev = dev or torch.device("cuda", 0)
    ops.manual_seed(7, dev)
    cfg = HamtConfig(hamt_precision="bf16", pretrain_tasks={"mlm", "sap", "sar", "sprel", "mrc", "itm"})
    model = MultiStepNavImagePreTraining(cfg).to(dev).train()
    named = list(model.named_parameters())
    n_par = sum(p.numel() for _, p in named)
    opt = AdamW([{"params": [p for n, p in named if not any(nd in n for nd in NO_DECAY)], "weight_decay": 0.01},
                 {"params": [p for n, p in named if any(nd in n for nd in NO_DECAY)], "weight_decay": 0.0}], lr=5e-5, betas=(0.9, 0.98))
    opt.materialize()
    sched = TaskSchedule(cyclic=True)
    g = torch.Generator(device=dev); g.manual_seed(3)
    img = lambda *s: torch.randn(*s, device=dev, generator=g)
    batches = {}

    def get(step):
        task = sched.task_at(step)
        if task not in batches:
            b = make_batch(task, B, cfg, seed=50 + step, txt_len=L, hist_len=T, mlm_exact=9 if task == "mlm" else None, device=dev)
            n = b["txt_ids"].shape[0]
            for k in ("hist_img_fts", "hist_pano_img_fts", "ob_img_fts"):
                b.pop(k, None)
            b["hist_images"], b["hist_pano_images"] = img(n, T, 3, 224, 224), img(n, T, V, 3, 224, 224)
            if task in ("sap", "sar", "sprel"):
                b["ob_images"], b["ob_v_exists"] = img(n, V, 3, 224, 224), torch.ones(n, V, dtype=torch.bool, device=dev)
            if task == "itm":
                r = make_itm_rng(b, seed=step)
                b["itm_neg_idxs"], b["itm_shuffled_pos_ids"] = r["neg_idxs"], r["shuffled_pos_ids"]
            batches[task] = b
        return task, batches[task]

    graphed = None
    if use_graph:
        from vln_hamt_amd.graph import GraphedTrainStep
        graphed = GraphedTrainStep(model, opt, max_grad_norm=5.0)

    def step(s):
        task, b = get(s)
        if graphed is not None:
            graphed.step(task, b, task)
            return task, b["txt_ids"].shape[0]
        loss = model(b, task, True).mean()
        loss.backward()
        clip_grad_norm_(model.parameters(), 5.0, optimizer=opt)
        opt.step(); opt.zero_grad(); ops.advance_rng_epoch(dev)
        return task, b["txt_ids"].shape[0]

    for s in range(12):
        step(s)
    torch.cuda.synchronize()
    per, n_pano, flops = {}, 0, 0.0
    t0 = time.perf_counter()
    for s in range(steps):
        t1 = time.perf_counter()
        task, n = step(12 + s)
        if verbose:
            torch.cuda.synchronize()
            per.setdefault(task, []).append(time.perf_counter() - t1)
        n_pano += n
        n_grad = n * T + (n * V if task in ("sap", "sar", "sprel") else 0)
        flops += (n * T * V + 3 * n_grad) * VIT_FWD_GF * 1e9 + 3.0 * n * _trunk_fwd(task)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    finite = bool(torch.isfinite(opt._flat_p).all())
    out = {"workload": "BASELINE config 4: image-input pretrain step -- ViT-B/16 over raw 224x224 views (T*36 panorama views no-grad, T history + 36 "
                       "observation views with gradient) + the feature-input trunk, six-task mix, fwd + bwd + clip + AdamW over all parameters, dropout 0.1",
           "per_gpu_batch": B, "txt_len": L, "hist_len": T, "views": V, "parameters_M": round(n_par / 1e6, 1), "steps": steps,
           "ms_per_step": round(dt / steps * 1e3, 2), "value": round(n_pano / dt, 2), "unit": "panorama-steps/s",
           "no_grad_views_per_s": round(n_pano * T * V / dt, 0), "launch": "hipGraph replay" if use_graph else "eager",
           "roofline": {"bound": "mfma", "achieved": round(flops / dt / 1e12, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(flops / dt / 1e12 / 2500.0, 4),
                        "work": "35.1 GFLOP per view forward (x1 no-grad panorama views, x3 views with gradient) + 3 x trunk forward (SURVEY 8a formula at L = 60)"},
           "state_finite": finite, "hbm_peak_gb": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 1)}
    if verbose:
        out["per_task_ms"] = {k: round(sum(v) / len(v) * 1e3, 1) for k, v in per.items()}
    del graphed, opt, model, batches
    torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    use_graph = len(sys.argv) > 3 and sys.argv[3] == "graph"
    import json
    print(json.dumps(run(B, steps, use_graph, verbose=True), indent=1))
