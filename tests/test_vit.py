"""Row N3: ViT backbone.  CPU: the oracle restatement against goldens produced by the reference's own
VisionTransformer class (oracle/gen_goldens.py gen_vit, through the timm stand-ins of oracle/ref_shim.py).
GPU: the HIP module against the same goldens (forward at 1e-3 fp32 / 1e-2 bf16, gradients on the tiny config)."""
import math

import numpy as np
import pytest
import torch

from _util import load_npz

TOL = {"fp32": 1e-3, "bf16": 1e-2}


def _rel(a, b):
    a, b = a.detach().double().cpu(), torch.as_tensor(b).double()
    return float((a - b).abs().max()) / max(1.0, float(b.abs().max()))


def _inputs(store, tag, c):
    rng = np.random.Generator(np.random.PCG64(5))
    n_img = store[f"{tag}/probe"].shape[0]
    imgs = torch.from_numpy(rng.standard_normal((n_img, c.in_chans, c.img_size, c.img_size), dtype=np.float32))
    probe = torch.from_numpy(rng.standard_normal((n_img, c.embed_dim), dtype=np.float32))
    assert np.array_equal(probe.numpy(), store[f"{tag}/probe"])                     # the recipe regenerates the fixture's inputs
    crop = store[f"{tag}/images"]
    assert np.array_equal(imgs[:, :, :crop.shape[2], :crop.shape[3]].numpy(), crop)
    return imgs, probe


@pytest.mark.parametrize("tag", ["tiny", "b16"])
def test_vit_oracle_matches_reference_goldens(tag):
    from oracle.hamt_oracle import VitConfig, make_vit_state_dict, vit_forward_features
    store = load_npz("vit.npz")
    c = VitConfig.tiny() if tag == "tiny" else VitConfig()
    if tag == "b16":
        torch.set_num_threads(8)
    sd = {k: v.clone().requires_grad_(tag == "tiny") for k, v in make_vit_state_dict(c, seed=21).items()}
    imgs, probe = _inputs(store, tag, c)
    feats = vit_forward_features(sd, c, imgs)
    assert _rel(feats, store[f"{tag}/feats"]) <= 1e-6
    if tag == "tiny":
        (feats * probe).sum().backward()
        for k, v in sd.items():
            assert _rel(v.grad, store[f"{tag}/grad/{k}"]) <= 1e-6, k


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("tag", ["tiny", "b16"])
def test_vit_hip_matches_reference_goldens(tag, prec):
    from oracle.hamt_oracle import VitConfig, make_vit_state_dict
    from vln_hamt_amd.model.vision_transformer import VisionTransformer
    store = load_npz("vit.npz")
    c = VitConfig.tiny() if tag == "tiny" else VitConfig()
    model = VisionTransformer(c.img_size, c.patch_size, c.in_chans, c.embed_dim, c.depth, c.num_heads, c.mlp_ratio, hamt_precision=prec)
    model.load_state_dict(make_vit_state_dict(c, seed=21), strict=True)                # the reference's parameter names
    model = model.cuda().train(tag == "tiny")
    imgs, probe = _inputs(store, tag, c)
    if tag == "b16":
        with torch.no_grad():                                                           # the no-grad panorama pass
            feats = model.forward_features(imgs.cuda())
    else:
        feats = model.forward_features(imgs.cuda())
    e = _rel(feats, store[f"{tag}/feats"])
    msg = f"[vit {tag} {prec}] features err {e:.2e}"
    assert e <= TOL[prec], msg
    if tag == "tiny":
        (feats * probe.cuda()).sum().backward()
        gmax = max(float(np.linalg.norm(store[k])) for k in store if k.startswith("tiny/grad/"))
        dot = n1 = n2 = 0.0
        worst = 0.0
        for k, p in model.named_parameters():
            r = torch.from_numpy(store[f"tiny/grad/{k}"]).double()
            g = p.grad.detach().double().cpu()
            worst = max(worst, abs(float(g.norm()) - float(r.norm())) / max(float(r.norm()), 5e-2 * gmax))
            dot += float((g * r).sum()); n1 += float((g * g).sum()); n2 += float((r * r).sum())
        cos = dot / math.sqrt(n1 * n2)
        msg += f"; grad cosine {cos:.6f}, worst per-parameter norm error {worst:.2e}"
        assert cos >= (0.99999 if prec == "fp32" else 0.995) and worst <= (2e-3 if prec == "fp32" else 6e-2), msg
    print(msg)
