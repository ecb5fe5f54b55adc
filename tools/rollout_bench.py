#!/usr/bin/env python3
"""BASELINE config 5 shape (RxR long-horizon rollout): NavCMT `language` once, then per step `history` (one panorama)
+ `visual` (text x {history so far, 37 observation tokens}), B=8, L=160, up to 20 history steps, image_feat 512,
`no_lang_ca`.  Forward latency per rollout step (hipGraph replay per step, and eager launches, bf16 path), and with grad (IL loss
on the action logits of every step + one backward over the whole rollout).  A measurement of the 'next' row N2, not the headline
bench; bench.py imports `run` for its `rollout_step` key."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

H = 768


def _visual_fwd_flops(L, n_hist, V1):
    """forward FLOPs of one `visual` call per episode with no_lang_ca (vilmodel_cmt.py:701-709: the text side only serves as keys /
    values -- projected once per episode here): 4 x-layers over Vn = n_hist + V1 visual tokens = cross attention of the visual queries over
    L text keys (q proj, scores + context, out proj), visual self attention and FFN; + the observation embedder."""
    Vn = n_hist + V1
    per_layer = (2 * Vn * H * H + 4 * Vn * L * H + 2 * Vn * H * H) + (8 * Vn * H * H + 4 * Vn * Vn * H) + 16 * Vn * H * H
    return 4 * per_layer


def run(batch=8, txt=160, steps=20, feat=512, reps=5, dev=None, with_train=True):
    from vln_hamt_amd.modeling import HamtConfig
    from vln_hamt_amd.models.vilmodel_cmt import NavCMT
    from vln_hamt_amd import ops
    from vln_hamt_amd.graph import GraphedInference
    dev = dev or torch.device("cuda")
    cfg = HamtConfig(hamt_precision="bf16", image_feat_size=feat, hist_enc_pano=True, num_h_pano_layers=2, no_lang_ca=True,
                     act_pred_token="ob_txt", fix_lang_embedding=False, fix_hist_embedding=False, fix_obs_embedding=False,
                     update_lang_bert=True, vocab_size=250002 // 8 * 8)
    torch.manual_seed(0)
    model = NavCMT(cfg).to(dev)
    B, L, T, V, D = batch, txt, steps, 36, feat
    g = torch.Generator(device="cpu").manual_seed(1)
    txt_ids = torch.randint(5, 30000, (B, L), generator=g).to(dev); txt_masks = torch.ones(B, L, dtype=torch.bool, device=dev)
    pano = torch.randn(T, B, V, D, generator=g).to(dev); pang = torch.randn(T, B, V, 4, generator=g).to(dev)
    img = torch.randn(T, B, D, generator=g).to(dev); ang = torch.randn(T, B, 4, generator=g).to(dev)
    ob_img = torch.randn(T, B, V + 1, D, generator=g).to(dev); ob_ang = torch.randn(T, B, V + 1, 4, generator=g).to(dev)
    nav = torch.zeros(B, V + 1, dtype=torch.long, device=dev); nav[:, :4] = 1; nav[:, V] = 2
    ob_masks = torch.ones(B, V + 1, dtype=torch.bool, device=dev)
    target = torch.randint(0, 4, (B,), generator=g).to(dev)

    def rollout(train):
        model.train(train)
        lang = model("language", txt_ids=txt_ids, txt_masks=txt_masks)
        hs = [model("history").expand(B, -1)]
        loss = 0.0
        for t in range(T):
            hist = torch.stack(hs, 1)
            hist_masks = torch.ones(B, len(hs), dtype=torch.bool, device=dev)
            out = model("visual", txt_embeds=lang, hist_embeds=hist, txt_masks=txt_masks, hist_masks=hist_masks,
                        ob_img_feats=ob_img[t], ob_ang_feats=ob_ang[t], ob_nav_types=nav, ob_masks=ob_masks)
            if train:
                loss = loss + ops.cross_entropy(out[0], target).mean()
            hs.append(model("history", hist_img_feats=img[t], hist_ang_feats=ang[t], ob_step_ids=torch.tensor([t], device=dev),
                            hist_pano_img_feats=pano[t], hist_pano_ang_feats=pang[t]))
        if train:
            loss.backward()
            model.zero_grad(set_to_none=True)

    model.eval()
    with torch.no_grad():
        lang_static = model("language", txt_ids=txt_ids, txt_masks=txt_masks)
        cls_h = model("history").expand(B, -1).contiguous()
    gv = GraphedInference(lambda hist, hm, oi, oa: model("visual", txt_embeds=lang_static, hist_embeds=hist, txt_masks=txt_masks, hist_masks=hm,
                                                          ob_img_feats=oi, ob_ang_feats=oa, ob_nav_types=nav, ob_masks=ob_masks))
    gh = GraphedInference(lambda i_, a_, sid, p_, pa_: model("history", hist_img_feats=i_, hist_ang_feats=a_, ob_step_ids=sid,
                                                              hist_pano_img_feats=p_, hist_pano_ang_feats=pa_))

    def rollout_graphed():
        """inference rollout with one captured graph per history length (`visual`) and one for `history`"""
        hs = [cls_h]
        for t in range(T):
            hist = torch.stack(hs, 1)
            hist_masks = torch.ones(B, len(hs), dtype=torch.bool, device=dev)
            gv(("visual", len(hs)), hist, hist_masks, ob_img[t], ob_ang[t])
            hs.append(gh("history", img[t], ang[t], torch.tensor([t], device=dev), pano[t], pang[t]).clone())

    rollout_graphed(); torch.cuda.synchronize()          # captures
    t0 = time.perf_counter()
    for _ in range(reps):
        rollout_graphed()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    pano_layer = lambda S: 24 * S * H * H + 4 * S * S * H
    flops = sum(B * (_visual_fwd_flops(L, n + 1, V + 1) + 2 * pano_layer(V) + 2 * V * D * H + 2 * (V + 1) * D * H) for n in range(T))
    out = {"workload": "BASELINE config 5 shape: RxR rollout forward through NavCMT -- per agent step one `visual` call (4 cross-modal layers over "
                       "history-so-far + 37 observation tokens against the 160-token instruction, no_lang_ca, text keys / values projected once per "
                       "episode) and one `history` call (2-layer panorama encoder over 36 views); the `language` pass runs once per episode and is excluded",
           "batch": B, "txt_len": L, "rollout_steps": T, "image_feat": D, "launch": "hipGraph replay per step",
           "ms_per_step": round(dt / T * 1e3, 3), "value": round(B * T / dt, 1), "unit": "agent-steps/s",
           "roofline": {"bound": "mfma", "achieved": round(flops / dt / 1e12, 2), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(flops / dt / 1e12 / 2500.0, 5),
                        "work": "forward FLOPs of the visual + history calls of the rollout (launch-latency bound at B = 8: ~100 kernels of a few us per agent step)"}}
    for train in ((False, True) if with_train else (False,)):
        ctx = torch.enable_grad() if train else torch.no_grad()
        with ctx:
            rollout(train); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(max(1, reps // 2)):
                rollout(train)
            torch.cuda.synchronize()
        d2 = (time.perf_counter() - t0) / max(1, reps // 2)
        out["eager_train_ms_per_step" if train else "eager_inference_ms_per_step"] = round(d2 / T * 1e3, 3)
    del gv, gh, model
    torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8); ap.add_argument("--txt", type=int, default=160)
    ap.add_argument("--steps", type=int, default=20); ap.add_argument("--feat", type=int, default=512)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    import json
    print(json.dumps(run(a.batch, a.txt, a.steps, a.feat, a.reps), indent=1))
