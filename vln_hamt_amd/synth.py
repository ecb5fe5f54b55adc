"""Synthetic proxy-task batches that reproduce the reference's collate conventions.

The real loaders (pretrain_src/data/r2r_tasks.py) need the Matterport/R2R datasets, which are not
available; this generator restates only the *shape / dtype / padding* contract the model sees:

* zero-padded features, bool masks built from lengths, history mask has +1 for the cls slot
  (r2r_tasks.py:373-374), ``hist_*`` = None when every sample is at step 0 (r2r_tasks.py:360-366);
* observation = 36 views + a zero STOP row with nav_type 2 at the last index
  (r2r_data.py:205-211); nav_type 1 for navigable views;
* ``txt_ids`` = [CLS]=101 ... [SEP]=102, body uniform in the reference's vocab range
  (r2r_tasks.py:60-62), ``txt_labels`` = -1 for "ignore" (r2r_tasks.py:46), at least one masked
  token per sample (r2r_tasks.py:48-51);
* the ITM loader halves the batch (loader.py:130).

Everything is drawn from a numpy PCG64 stream so the same call gives the same batch on any host.
"""
from __future__ import annotations

import numpy as np
import torch

TASKS = ("mlm", "sap", "sar", "sprel", "mrc", "itm")
MIX_RATIO = {"mlm": 5, "sap": 1, "sar": 1, "sprel": 1, "mrc": 2, "itm": 2}   # pretrain_r2r.json:43-58


def _angles(rng, shape):
    h = rng.uniform(-np.pi, np.pi, size=shape).astype(np.float32)
    e = rng.uniform(-np.pi / 6, np.pi / 6, size=shape).astype(np.float32)
    return np.stack([np.sin(h), np.cos(h), np.sin(e), np.cos(e)], -1).astype(np.float32)  # r2r_data.py:14-17


def text_pack_plan(lens, L: int, bucket: int = 128):
    """(pack_idx int64 [M], cu_seqlens int32 [n + 1], unpack_idx int64 [B L]) for a padded text batch [B, L] with `lens` real tokens per
    row, or None when (almost) nothing is padding.  pack_idx = the flat positions b L + i of the real tokens, row by row, then filler
    positions up to a multiple of `bucket` rows (a static length per bucket for captured graphs); cu_seqlens = the B sequences' row
    ranges followed by ceil(bucket / L) filler slots -- sequences of at most L rows each that cover the filler rows, the rest empty -- so
    that every packed row belongs to a sequence and `cu` has ONE length per (B, L, bucket); unpack_idx sends each
    padded position to a packed row -- its own for a real token, its sequence's first row for padding (any finite value serves there).
    Consumed by model.vilmodel.NavPreTrainedModel._text."""
    lens = np.asarray(lens).astype(np.int64)
    B = lens.shape[0]
    M = int(lens.sum())
    Mb = (M + bucket - 1) // bucket * bucket
    if Mb >= B * L:
        return None
    starts = np.concatenate([[0], np.cumsum(lens)])
    valid = np.arange(L)[None, :] < lens[:, None]
    flat = np.flatnonzero(valid.reshape(-1))
    pack = np.concatenate([flat, np.full(Mb - M, flat[0], dtype=np.int64)])
    cu = list(starts)
    while cu[-1] < Mb:
        cu.append(min(Mb, cu[-1] + L))
    # a FIXED number of filler slots for this (B, L, bucket) -- the unused ones are empty sequences, which the kernels skip -- so that
    # the shape of `cu` (part of a captured step's key, graph.GraphedTrainStep.key_for) depends on the packed row count's bucket only
    cu += [Mb] * (B + 1 + (bucket + L - 1) // L - len(cu))
    unpack = np.repeat(starts[:-1], L)                       # padding -> the sequence's first row
    unpack[flat] = np.arange(M)
    return torch.from_numpy(pack), torch.tensor(cu, dtype=torch.int32), torch.from_numpy(unpack)


def make_batch(task: str, batch_size: int, cfg, seed: int = 0, txt_len: int = 80, hist_len: int = 5,
               num_views: int = 36, ragged: bool = False, device="cpu", mlm_exact: int | None = None, txt_pack: bool = True) -> dict:
    """Build one collated batch for `task` (one of TASKS).

    ragged=True draws per-sample text lengths in [txt_len//4, txt_len] and history lengths in
    [0, hist_len]; otherwise every sample is full length.  `mlm_exact`, when given, fixes the number
    of masked tokens per sample (used by the fixed-shape bench); else ~15 % as in r2r_tasks.py:23.
    """
    task = task.split("_")[0]
    assert task in TASKS, task
    rng = np.random.Generator(np.random.PCG64(seed))
    B = batch_size // 2 if task == "itm" else batch_size          # loader.py:130
    B = max(B, 1)
    D, A = cfg.image_feat_size, cfg.angle_feat_size
    L, T, V = txt_len, hist_len, num_views
    out: dict = {}

    # ---- text
    lens = rng.integers(max(4, L // 4), L + 1, size=B) if ragged else np.full(B, L)
    lens[rng.integers(0, B)] = L                                  # batch max length == L
    lo, hi = (1996, 29611) if cfg.vocab_size >= 29611 else (5, cfg.vocab_size)
    ids = np.zeros((B, L), dtype=np.int64)
    for b in range(B):
        ids[b, :lens[b]] = rng.integers(lo, hi, size=lens[b])
        ids[b, 0], ids[b, lens[b] - 1] = 101 % cfg.vocab_size, 102 % cfg.vocab_size
    out["txt_masks"] = torch.from_numpy(np.arange(L)[None] < lens[:, None])
    if task == "mlm":
        labels = np.full((B, L), -1, dtype=np.int64)
        for b in range(B):
            body = np.arange(1, lens[b] - 1)
            if mlm_exact is not None:
                pick = rng.choice(body, size=min(mlm_exact, len(body)), replace=False)
            else:
                pick = body[rng.random(len(body)) < 0.15]
                if len(pick) == 0:
                    pick = body[:1]
            labels[b, pick] = ids[b, pick]
            ids[b, pick] = 103 % cfg.vocab_size
        out["txt_labels"] = torch.from_numpy(labels)
        out["txt_label_idx"] = torch.from_numpy(np.flatnonzero(labels.reshape(-1) != -1).astype(np.int64))
    out["txt_ids"] = torch.from_numpy(ids)
    if ragged and txt_pack:
        plan = text_pack_plan(lens, L)
        if plan is not None:
            out["txt_pack_idx"], out["txt_cu"], out["txt_unpack_idx"] = plan

    # ---- history
    hl = rng.integers(0, T + 1, size=B) if ragged else np.full(B, T)
    if T > 0 and ragged:
        hl[rng.integers(0, B)] = T
    if task in ("mrc", "itm"):
        hl = np.maximum(hl, min(T, 2))                            # these tasks need a real trajectory
    if T == 0 or hl.max() == 0:
        for k in ("hist_img_fts", "hist_ang_fts", "hist_pano_img_fts", "hist_pano_ang_fts"):
            out[k] = None
        hl = np.zeros(B, dtype=np.int64)
        Tm = 0
    else:
        Tm = int(hl.max())
        valid = (np.arange(Tm)[None] < hl[:, None]).astype(np.float32)
        out["hist_img_fts"] = torch.from_numpy(rng.standard_normal((B, Tm, D), dtype=np.float32) * valid[..., None])
        out["hist_ang_fts"] = torch.from_numpy(_angles(rng, (B, Tm))[..., :A] * valid[..., None])
        out["hist_pano_img_fts"] = torch.from_numpy(
            rng.standard_normal((B, Tm, V, D), dtype=np.float32) * valid[..., None, None])
        out["hist_pano_ang_fts"] = torch.from_numpy(_angles(rng, (B, Tm, V))[..., :A] * valid[..., None, None])
    out["hist_masks"] = torch.from_numpy(np.arange(Tm + 1)[None] < (hl + 1)[:, None])

    # ---- observation (sap / sar / sprel)
    if task in ("sap", "sar", "sprel"):
        img = rng.standard_normal((B, V + 1, D), dtype=np.float32)
        ang = _angles(rng, (B, V + 1))[..., :A]
        img[:, V], ang[:, V] = 0.0, 0.0                           # STOP row
        nav = np.zeros((B, V + 1), dtype=np.int64)
        nav[:, V] = 2
        for b in range(B):
            nav[b, rng.choice(V, size=4, replace=False)] = 1
        out.update(ob_img_fts=torch.from_numpy(img), ob_ang_fts=torch.from_numpy(ang),
                   ob_nav_types=torch.from_numpy(nav), ob_masks=torch.ones(B, V + 1, dtype=torch.bool))
        if task == "sap":
            act = np.array([rng.choice(np.nonzero(nav[b])[0]) for b in range(B)], dtype=np.int64)
            out["ob_action_viewindex"] = torch.from_numpy(act)
        elif task == "sar":
            out["ob_action_angles"] = torch.from_numpy(rng.uniform(-np.pi, np.pi, size=(B, 2)).astype(np.float32))
            out["ob_progress"] = torch.from_numpy(rng.uniform(0, 1, size=(B,)).astype(np.float32))
        else:
            out["sp_anchor_idxs"] = torch.from_numpy(rng.integers(0, V, size=B).astype(np.int64))
            out["sp_targets"] = torch.from_numpy(rng.uniform(-np.pi, np.pi, size=(B, V, 2)).astype(np.float32))

    # ---- MRC targets
    if task == "mrc":
        m = (rng.random((B, Tm)) < 0.15) & (np.arange(Tm)[None] < hl[:, None])
        for b in range(B):
            if not m[b].any():
                m[b, rng.integers(0, hl[b])] = True               # r2r_tasks.py: at least one region
        logits = rng.standard_normal((B, Tm, cfg.image_prob_size), dtype=np.float32) * 2.0
        probs = np.exp(logits - logits.max(-1, keepdims=True))
        probs = (probs / probs.sum(-1, keepdims=True)).astype(np.float32)
        img = out["hist_img_fts"].numpy()
        img[m] = 0.0                                              # masked step features are zeroed
        out["hist_img_fts"] = torch.from_numpy(img)
        out["hist_mrc_masks"] = torch.from_numpy(m)
        out["hist_mrc_idx"] = torch.from_numpy(np.flatnonzero(m.reshape(-1)).astype(np.int64))
        out["hist_img_probs"] = torch.from_numpy(probs)

    if device != "cpu":
        out = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in out.items()}
    return out


def make_itm_rng(batch: dict, seed: int = 0, num_neg_trajs: int = 4) -> dict:
    """Draw the ITM negatives the reference samples inside forward (vilmodel.py:681-704) as explicit,
    injectable indices: `neg_idxs` (B,K) in-batch negatives (never i itself) and K position tables
    whose first hist_len entries are a permutation and whose tail is the identity."""
    rng = np.random.Generator(np.random.PCG64(seed))
    masks = batch["hist_masks"].cpu().numpy()
    B, T = masks.shape[0], masks.shape[1] - 1
    K = num_neg_trajs // 2
    neg = None
    if B > 1:
        neg = np.stack([rng.choice([j for j in range(B) if j != i], size=K) for i in range(B)], 0).astype(np.int64)
    else:
        K = num_neg_trajs
    lens = masks.sum(1) - 1
    tabs = []
    for _ in range(K):
        tab = np.tile(np.arange(T, dtype=np.int64), (B, 1))
        for i in range(B):
            tab[i, :lens[i]] = rng.permutation(lens[i])
        tabs.append(torch.from_numpy(tab))
    dev = batch["hist_masks"].device
    return {"neg_idxs": None if neg is None else torch.from_numpy(neg).to(dev),
            "shuffled_pos_ids": [t.to(dev) for t in tabs]}


def make_samples(task: str, n: int, seed: int = 0, feat: int = 768, ang: int = 4, prob: int = 1000, max_txt: int = 80,
                 max_hist: int = 5, views: int = 36, first_step: bool = False) -> list:
    """`n` per-sample dicts as the reference's Dataset.__getitem__ returns them (r2r_tasks.py:69-93, 168-200, 243-266,
    305-341, 398-437, 521-551): ragged torch tensors + python scalars, the input of the *_collate functions.
    first_step=True: every history is empty (the `hist_img_fts = None` branch of the SAP/SAR/SPREL collates)."""
    task = task.split("_")[0]
    rng = np.random.Generator(np.random.PCG64(seed))
    out = []
    for _ in range(n):
        L = int(rng.integers(max(4, max_txt // 4), max_txt + 1))
        T = 0 if first_step else int(rng.integers(0 if task in ("sap", "sar", "sprel") else 1, max_hist + 1))
        s = {"txt_ids": torch.from_numpy(rng.integers(1000, 29000, size=L)), "txt_lens": L}
        if task == "mlm":
            lab = np.full(L, -1, dtype=np.int64)
            m = rng.random(L) < 0.15
            lab[m] = s["txt_ids"].numpy()[m]
            s["txt_labels"] = torch.from_numpy(lab)
        if task in ("sap", "sar", "sprel"):
            V = int(rng.integers(views - 4, views + 1)) + 1              # candidates + views (+ stop)
            s["ob_img_fts"] = torch.from_numpy(rng.standard_normal((V, feat), dtype=np.float32))
            s["ob_ang_fts"] = torch.from_numpy(_angles(rng, (V,)))
            s["ob_nav_types"] = torch.from_numpy(rng.integers(0, 3, size=V))
            s["ob_lens"] = V
        s["hist_img_fts"] = torch.from_numpy(rng.standard_normal((T, feat), dtype=np.float32))
        s["hist_ang_fts"] = torch.from_numpy(_angles(rng, (T,)))
        s["hist_pano_img_fts"] = torch.from_numpy(rng.standard_normal((T, views, feat), dtype=np.float32))
        s["hist_pano_ang_fts"] = torch.from_numpy(_angles(rng, (T, views)))
        s["hist_lens"] = T
        if task == "mrc":
            s["hist_img_probs"] = torch.from_numpy(rng.random((T, prob), dtype=np.float32))
            s["hist_mrc_masks"] = torch.from_numpy(rng.random(T) < 0.3)
        if task == "sap":
            s["ob_action_viewindex"] = int(rng.integers(0, s["ob_lens"]))
        if task == "sar":
            s["ob_action_angles"] = rng.uniform(-np.pi, np.pi, size=2)
            s["ob_progress"] = float(rng.random())
        if task == "sprel":
            s["sp_anchor_idxs"] = int(rng.integers(0, views))
            s["sp_targets"] = rng.uniform(-np.pi, np.pi, size=(views, 2)).astype(np.float32)
        out.append(s)
    return out
