"""Shared helpers for the tests (golden loading, batch reconstruction)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_npz(name):
    with np.load(os.path.join(GOLDEN, name), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def sub(store, prefix):
    """Entries of `store` under `prefix` with the prefix stripped."""
    return {k[len(prefix):]: v for k, v in store.items() if k.startswith(prefix)}


def batch_from(store, tag, device="cpu"):
    """Rebuild the input batch (and injected ITM indices) of a golden case."""
    b = {k: torch.from_numpy(v).to(device) for k, v in sub(store, f"{tag}/in/").items()}
    rng = sub(store, f"{tag}/rng/")
    itm = None
    if rng:
        tabs = [torch.from_numpy(rng[k]).to(device) for k in sorted(k for k in rng if k.startswith("shuffled_pos_ids."))]
        itm = {"neg_idxs": torch.from_numpy(rng["neg_idxs"]).to(device) if "neg_idxs" in rng else None,
               "shuffled_pos_ids": tabs}
    return b, itm


def tiny_cfg(**kw):
    from oracle.hamt_oracle import OracleConfig
    return OracleConfig.tiny(hidden_size=128, num_attention_heads=2, intermediate_size=256, image_feat_size=64, **kw)


def grad_probe(g, n=257):
    """the generator's probe of a gradient: n evenly strided elements, zero padded (oracle/gen_goldens.py::grad_probe)"""
    f = g.detach().flatten().cpu()
    pr = f[:: max(1, f.numel() // n)][:n]
    out = np.zeros(n, dtype=np.float32)
    out[:pr.numel()] = pr.float().numpy()
    return out


def canon_batch(store, task, cfg):
    from vln_hamt_amd.synth import make_batch
    batch = make_batch(task, 2 if task != "itm" else 4, cfg, seed=int(store[f"{task}/seed"]), txt_len=80, hist_len=5)
    rng = sub(store, f"{task}/rng/")
    itm = None
    if rng:
        itm = {"neg_idxs": torch.from_numpy(rng["neg_idxs"]),
               "shuffled_pos_ids": [torch.from_numpy(rng[k]) for k in sorted(rng) if k.startswith("shuffled")]}
    return batch, itm
