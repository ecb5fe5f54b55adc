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


def _b16_grad_check(named, store, what, cos_min, norm_tol):
    """gradients of sum(features * probe) at ViT-B/16, 224 x 224 against the reference's autograd (vit.npz b16/grad_*)"""
    from _util import grad_probe
    names = [str(n) for n in store["b16/grad_names"]]
    norms, probes = store["b16/grad_norms"], store["b16/grad_probes"].astype(np.float64)
    assert set(names) == set(named), set(names) ^ set(named)
    got = np.stack([grad_probe(named[k].grad, probes.shape[1]) for k in names]).astype(np.float64)
    gn = np.array([float(named[k].grad.double().norm()) for k in names])
    pcos = float((got * probes).sum() / np.sqrt((got ** 2).sum() * (probes ** 2).sum()))
    nerr = float(np.max(np.abs(gn - norms) / np.maximum(norms, 5e-2 * float(norms.max()))))
    print(f"[vit b16 backward {what}] probe cosine {pcos:.6f}, worst per-parameter norm error {nerr:.2e}")
    assert pcos >= cos_min and nerr <= norm_tol, (what, pcos, nerr)


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
        if "b16/grad_names" in store:       # BASELINE config 4's backbone, backward at full size (history / observation views train it)
            model.train(True)
            for m_ in model.modules():
                if isinstance(m_, torch.nn.Dropout):
                    m_.p = 0.0
            f2 = model.forward_features(imgs.cuda())
            assert _rel(f2, store["b16/feats"]) <= TOL[prec]
            (f2 * probe.cuda()).sum().backward()
            _b16_grad_check(dict(model.named_parameters()), store, prec, 0.99999 if prec == "fp32" else 0.995, 2e-3 if prec == "fp32" else 6e-2)
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


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_image_input_model_vs_oracle_composition(prec):
    """Config-4 model (images -> ViT backbone -> the feature-input model): SAP and MRC losses + gradients against the
    composition of the two pinned oracles (vit_forward_features, HamtOracle) on tiny configs, dropout off."""
    sys_path_oracle = None
    from _util import tiny_cfg
    from oracle.hamt_oracle import HamtOracle, VitConfig, make_state_dict, make_vit_state_dict, pretrain_param_shapes, vit_forward_features
    from vln_hamt_amd.model.image_pretrain import MultiStepNavImagePreTraining
    from vln_hamt_amd.modeling import HamtConfig
    from vln_hamt_amd.synth import make_batch
    vc = VitConfig.tiny()                              # 32x32 images, 128-d features
    ocfg = tiny_cfg()
    ocfg.image_feat_size = vc.embed_dim
    for k in ("hidden_dropout_prob", "attention_probs_dropout_prob", "pred_head_dropout_prob"):
        setattr(ocfg, k, 0.0)
    sd = make_state_dict(pretrain_param_shapes(ocfg), seed=3)
    vsd = make_vit_state_dict(vc, seed=4)
    kw = dict(vars(ocfg))
    model = MultiStepNavImagePreTraining(HamtConfig(hamt_precision=prec, **kw),
                                         vit_kwargs=dict(img_size=vc.img_size, patch_size=vc.patch_size, depth=vc.depth, num_heads=vc.num_heads, mlp_ratio=vc.mlp_ratio))
    full = {k: v.clone() for k, v in sd.items()}
    full.update({"bert.vision_backbone." + k: v.clone() for k, v in vsd.items()})
    model.load_state_dict(full, strict=True)
    model = model.cuda().train()
    B, T, V = 3, 2, 36
    rng = np.random.Generator(np.random.PCG64(17))
    img = lambda *s: torch.from_numpy(rng.standard_normal(s, dtype=np.float32))
    for task in ("sap", "mrc"):
        b = make_batch(task, B, ocfg, seed=40, txt_len=12, hist_len=T, ragged=False)
        ib = {k: v for k, v in b.items() if k not in ("hist_img_fts", "hist_pano_img_fts", "ob_img_fts")}
        ib["hist_images"], ib["hist_pano_images"] = img(B, T, 3, 32, 32), img(B, T, V, 3, 32, 32)
        if task == "sap":
            ib["ob_images"] = img(B, V, 3, 32, 32)
            ib["ob_v_exists"] = torch.ones(B, V, dtype=torch.bool)
            ib["ob_v_exists"][0, 5] = False
        # ---- oracle composition (CPU fp32 autograd)
        osd = {k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in sd.items()}
        ovsd = {k: v.clone().requires_grad_(True) for k, v in vsd.items()}
        fb = dict(b)
        hf = vit_forward_features(ovsd, vc, ib["hist_images"].reshape(-1, 3, 32, 32)).reshape(B, T, -1)
        with torch.no_grad():
            pf = vit_forward_features(ovsd, vc, ib["hist_pano_images"].reshape(-1, 3, 32, 32)).reshape(B, T, V, -1)
        if task == "mrc":
            hf = hf.masked_fill(b["hist_mrc_masks"].unsqueeze(-1), 0)
            pf = pf.masked_fill(b["hist_mrc_masks"].unsqueeze(-1).unsqueeze(-1), 0)
        fb["hist_img_fts"], fb["hist_pano_img_fts"] = hf, pf
        if task == "sap":
            of = vit_forward_features(ovsd, vc, ib["ob_images"].reshape(-1, 3, 32, 32)).reshape(B, V, -1)
            of = of.masked_fill(ib["ob_v_exists"].logical_not().unsqueeze(-1), 0)
            fb["ob_img_fts"] = torch.cat([of, of.new_zeros(B, 1, of.shape[2])], 1)
        oloss = HamtOracle(osd, ocfg, training=True).forward(fb, task, True)
        oloss.mean().backward()
        # ---- HIP
        model.zero_grad(set_to_none=True)
        loss = model({k: (v.cuda() if torch.is_tensor(v) else v) for k, v in ib.items()}, task, True)
        loss.mean().backward()
        e = _rel(loss, oloss.detach())
        ref = {k: v.grad for k, v in osd.items() if v.grad is not None}
        ref.update({"bert.vision_backbone." + k: v.grad for k, v in ovsd.items() if v.grad is not None})
        got = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
        dot = n1 = n2 = 0.0
        for k, r in ref.items():
            if k not in got:
                assert float(r.abs().max()) == 0.0, k
                continue
            g, r = got[k].detach().double().cpu(), r.double()
            dot += float((g * r).sum()); n1 += float((g * g).sum()); n2 += float((r * r).sum())
        cos = dot / math.sqrt(n1 * n2)
        print(f"[image model {task} {prec}] loss err {e:.2e}, global grad cosine {cos:.6f}")
        assert e <= TOL[prec] and cos >= (0.99999 if prec == "fp32" else 0.99), (task, e, cos)


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K,out_dt", [(197 * 3, 256, 128, torch.float32), (1000, 768, 3072, torch.float32), (333, 512, 64, torch.bfloat16),
                                          (197 * 16, 3072, 768, torch.bfloat16)])
def test_gemm_dropout_epilogue(M, N, K, out_dt):
    """HAMT_EPI_DROPOUT: the keep mask of element (m, n) is the one hamt_cast_pad_bf16_dropout re-creates (extracted by
    casting ones), values are exactly {0, 1/(1-p)}, the keep rate is 1-p, dropout comes BEFORE the residual add, and with
    GELU_GRAD both outputs carry the mask.  Covers the 4- and 8-wide epilogues and every tile height the dispatcher picks."""
    from vln_hamt_amd import _lib as L, ops
    dev = torch.device("cuda")
    ops.manual_seed(11, dev)
    g = torch.Generator(device=dev); g.manual_seed(2)
    a = torch.randn(M, K, device=dev, generator=g).bfloat16()
    w = (torch.randn(N, K, device=dev, generator=g) / math.sqrt(K)).bfloat16()
    bias = torch.randn(N, device=dev, generator=g)
    res = torch.randn(M, N, device=dev, generator=g)
    p, cid = 0.1, ops.next_call_id()
    mask = ops.cast_pad16_dropout(torch.ones(M, N, device=dev), p, cid)[:M].float()
    vals = torch.unique(mask)
    assert vals.numel() == 2 and float(vals[0]) == 0.0 and abs(float(vals[1]) - 1 / 0.9) < 5e-3
    keep = float((mask > 0).float().mean())
    assert abs(keep - 0.9) < 4 * math.sqrt(0.09 / (M * N)) + 1e-3, keep
    assert abs(float((mask[:, ::4] > 0).float().mean()) - 0.9) < 5e-3 and abs(float((mask[::3] > 0).float().mean()) - 0.9) < 5e-3
    keepf = (mask > 0).float() / 0.9
    base = a.float() @ w.float().t() + bias
    # residual form (proj / fc2)
    out = torch.empty(M, N, dtype=out_dt, device=dev)
    ops.gemm(a, w, out, bias=bias, epilogue=L.EPI_ADD_AUX, aux=res, drop=(p, cid))
    ref = res + keepf * base
    tol = 2e-2 if out_dt == torch.bfloat16 else 2e-3
    assert float((out.float() - ref).abs().max()) <= tol * max(1.0, float(ref.abs().max()))
    # activation form (fc1): gelu and gelu' both masked
    out2 = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    pre = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    ops.gemm(a, w, out2, bias=bias, epilogue=L.EPI_GELU_GRAD, aux=pre, drop=(p, cid))
    xr = base.clone().requires_grad_(True)
    gl = torch.nn.functional.gelu(xr)
    gl.sum().backward()
    assert float((out2.float() - keepf * gl.detach()).abs().max()) <= 2e-2 * max(1.0, float(gl.abs().max()))
    assert float((pre.float() - keepf * xr.grad).abs().max()) <= 2e-2
    # another call id / another epoch -> another mask
    m2 = ops.cast_pad16_dropout(torch.ones(M, N, device=dev), p, cid + 1)[:M].float()
    assert float(((m2 > 0) != (mask > 0)).float().mean()) > 0.1
    ops.advance_rng_epoch(dev)
    m3 = ops.cast_pad16_dropout(torch.ones(M, N, device=dev), p, cid)[:M].float()
    assert float(((m3 > 0) != (mask > 0)).float().mean()) > 0.1


@pytest.mark.gpu
def test_preln_blocks_with_branch_dropout_vs_torch():
    """The fused pre-LN half blocks with proj_drop / Mlp.drop on: forward and every gradient against torch fp32 autograd
    given the SAME masks (extracted through hamt_cast_pad_bf16_dropout with the call ids the Functions draw)."""
    from vln_hamt_amd import blocks_preln, ops
    dev = torch.device("cuda")
    ops.manual_seed(5, dev)
    g = torch.Generator(device=dev); g.manual_seed(9)
    B, S, D, H, I, p = 6, 197, 256, 4, 1024, 0.1
    M = B * S
    rn = lambda *s, sc=1.0: (torch.randn(*s, device=dev, generator=g) * sc).requires_grad_(True)
    x = rn(B, S, D)
    gam, bet = rn(D), rn(D, sc=0.1)
    with torch.no_grad():
        gam.add_(1.0)
    wq, bq, wp, bp = rn(3 * D, D, sc=D ** -0.5), rn(3 * D, sc=0.1), rn(D, D, sc=D ** -0.5), rn(D, sc=0.1)
    w1, b1, w2, b2 = rn(I, D, sc=D ** -0.5), rn(I, sc=0.1), rn(D, I, sc=I ** -0.5), rn(D, sc=0.1)
    probe = torch.randn(B, S, D, device=dev, generator=g)
    leaves = [x, gam, bet, wq, bq, wp, bp, w1, b1, w2, b2]
    names = "x gamma beta wqkv bqkv wproj bproj w1 b1 w2 b2".split()

    def masks(cid, n):
        return (ops.cast_pad16_dropout(torch.ones(M, n, device=dev), p, cid)[:M].float() > 0).float() / (1 - p)

    def check(tag, out, ref, used):
        e = float((out - ref).abs().max()) / max(1.0, float(ref.abs().max()))
        assert e <= 1e-2, (tag, e)
        for u in used:
            u.grad = None
        (out * probe).sum().backward()                      # (.backward(): weight gradients come from the deferred queue)
        go = [u.grad.clone() for u in used]
        gr = torch.autograd.grad((ref * probe).sum(), used, retain_graph=True)
        for t, a, b in zip([names[[id(l) for l in leaves].index(id(u))] for u in used], go, gr):
            ea = float((a - b).abs().max()) / max(1e-3, float(b.abs().max()))
            assert ea <= 3e-2, (tag, t, ea)
        print(f"[pre-LN {tag} with dropout] forward err {e:.2e}, gradients ok")

    # attention half: x + drop(proj(attn(qkv(LN x))))
    c0 = ops._call_counter[0]
    out = blocks_preln.PreLnAttnFn.apply(x, H, 0.0, p, 1e-6, gam, bet, wq, bq, wp, bp)
    mp = masks(c0 + 2, D)                                   # c0+1: the attention kernel's id, c0+2: proj_drop
    ln = torch.nn.functional.layer_norm(x, (D,), gam, bet, 1e-6)
    qkv = (ln @ wq.t() + bq).view(B, S, 3, H, D // H).permute(2, 0, 3, 1, 4)
    att = torch.softmax(qkv[0] @ qkv[1].transpose(-1, -2) * (D // H) ** -0.5, -1)
    cx = (att @ qkv[2]).transpose(1, 2).reshape(B, S, D)
    ref = x + mp.view(B, S, D) * (cx @ wp.t() + bp)
    check("attention", out, ref, [x, gam, bet, wq, bq, wp, bp])
    # MLP half: x + drop(fc2(drop(gelu(fc1(LN x)))))
    c0 = ops._call_counter[0]
    out = blocks_preln.PreLnMlpFn.apply(x, p, 1e-6, gam, bet, w1, b1, w2, b2)
    m1, m2 = masks(c0 + 1, I), masks(c0 + 2, D)
    h = m1.view(B, S, I) * torch.nn.functional.gelu(ln @ w1.t() + b1)
    ref = x + m2.view(B, S, D) * (h @ w2.t() + b2)
    check("mlp", out, ref, [x, gam, bet, w1, b1, w2, b2])


@pytest.mark.gpu
def test_vit_uninitialised_memory_never_reaches_a_result():
    """ViT-B/16 at full width, 3 images (591 token rows: ragged against every tile size), forward + backward in bf16 mode with every
    scratch / output buffer of the package pre-filled with NaN: same features, same finite gradients (see test_gpu_model)."""
    from oracle.hamt_oracle import VitConfig, make_vit_state_dict
    from test_gpu_model import _NanScratch
    from vln_hamt_amd.model.vision_transformer import VisionTransformer
    c = VitConfig()
    model = VisionTransformer(c.img_size, c.patch_size, c.in_chans, c.embed_dim, c.depth, c.num_heads, c.mlp_ratio, hamt_precision="bf16")
    model.load_state_dict(make_vit_state_dict(c, seed=21), strict=True)
    model = model.cuda().train(False)
    g = torch.Generator().manual_seed(5)
    imgs = torch.randn(3, 3, c.img_size, c.img_size, generator=g).cuda()
    probe = torch.randn(3, c.embed_dim, generator=g).cuda()
    res = []
    for poisoned in (False, True):
        with _NanScratch(poisoned):
            for p in model.parameters():
                p.grad = None
            feats = model.forward_features(imgs)
            (feats * probe).sum().backward()
            torch.cuda.synchronize()
            res.append((feats.detach().clone(), {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}))
    (f0, g0), (f1, g1) = res
    assert bool(torch.isfinite(f1).all()) and torch.equal(f0, f1)
    for k in g0:
        assert bool(torch.isfinite(g1[k]).all()), k
        if not k.endswith("qkv.bias"):
            assert float((g0[k] - g1[k]).abs().max()) <= 1e-4 * max(float(g0[k].abs().max()), 1e-6), k
