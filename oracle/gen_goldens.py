#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REAL reference.

TEST INFRASTRUCTURE; runs only in the build container (needs /root/reference).  Usage:

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_goldens.py

Outputs (all small .npz; inputs + expected outputs, no reference source):
  tiny_pretrain.npz   tiny config, every proxy task (ragged batch, hist=None SAP, B=1 ITM): inputs,
                      logits, losses, trunk embeddings, per-parameter gradient norms + small grads
  itm_rng.npz         the negatives the reference drew inside forward_itm under fixed seeds
  canon_pretrain.npz  R2R-canon full config (B=2, L=80, T=5): per-task logits/losses + probes
  canon_multi.npz     R2R-canon over two more weight seeds and at per-GPU batch 16: losses, trunk probes, gradient norms / probes,
                      single-candidate-logit gradients for ITM
  canon_multi_sar.npz the SAR gradient of the canon_multi draws term by term (single regression outputs)
  canon_b64.npz / canon_b64_sar.npz   the same two for ONE draw at the BENCHMARKED per-GPU batch 64 (`canon_b64`, `canon_b64_sar`; ~25 min of CPU, ~25 GB)
  canon_ragged.npz    R2R-canon on RAGGED batches (L ~ U[20, 80], T ~ U[0, 7], B = 16): what the packed text path is compared with
  optim_tiny.npz      3 steps of clip(5.0) + reference AdamW + warmup schedule on the tiny model
  tiny_finetune.npz   NavCMT language / history / visual modes (incl. no_lang_ca)
  a2c.npz             the agent's A2C loss block (agent_cmt.py:473-515), its own statements run on scripted rollout lists
  vit.npz             ViT backbone features / gradients from the reference's VisionTransformer class
  collate.npz         outputs of the reference's six *_collate functions on seeded ragged samples
  r2r_tiny/ + r2r_data.npz   a tiny R2R-style dataset (data files) and what the reference's MultiStepNavData reads / builds from it
  r2r_tasks.npz       items of the reference's six task datasets under fixed python / numpy seeds
  loader.npz          (task, batch) sequences of the reference's MetaLoader and build_dataloader's loader attributes
Weights always come from oracle.hamt_oracle.make_state_dict (numpy PCG64), never from random init.
"""
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle import ref_shim                                   # noqa: E402
from oracle.hamt_oracle import (HamtOracle, OracleConfig, make_state_dict, navcmt_param_shapes,  # noqa: E402
                                pretrain_param_shapes, decays, lr_at)
from vln_hamt_amd.synth import make_batch                     # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
torch.set_grad_enabled(True)
torch.manual_seed(0)


def sd_hash(sd):
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(sd[k].detach().cpu().numpy().tobytes())
    return h.hexdigest()


def to_np(d, prefix):
    out = {}
    for k, v in d.items():
        if v is None:
            continue
        if isinstance(v, (list, tuple)):
            for i, t in enumerate(v):
                out[f"{prefix}{k}.{i}"] = t.detach().cpu().numpy()
        else:
            out[f"{prefix}{k}"] = v.detach().cpu().numpy()
    return out


def build_ref_pretrain(cfg, sd):
    vil, cmt = ref_shim.import_pretrain()
    model = cmt.MultiStepNavCMTPreTraining(ref_shim.make_config(cfg))
    missing = model.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
    model.eval()
    return model, vil


class RngRecorder:
    """Record np.random.choice / torch.randperm draws made inside forward_itm (vilmodel.py:684, :698)."""

    def __init__(self, vil):
        self.vil, self.choices, self.perms = vil, [], []

    def __enter__(self):
        self._c, self._p = np.random.choice, torch.randperm

        def choice(a, size=None, *args, **kw):
            r = self._c(a, size, *args, **kw)
            self.choices.append(np.array(r))
            return r

        def randperm(n, *args, **kw):
            r = self._p(int(n), *args, **kw)
            self.perms.append(r.clone())
            return r
        np.random.choice, torch.randperm = choice, randperm
        return self

    def __exit__(self, *a):
        np.random.choice, torch.randperm = self._c, self._p


def itm_rng_from_record(rec, batch):
    B = batch["txt_ids"].shape[0]
    T = batch["hist_masks"].shape[1] - 1
    lens = (batch["hist_masks"].sum(1) - 1).tolist()
    neg = torch.from_numpy(np.stack(rec.choices, 0).astype(np.int64)) if rec.choices else None
    K = len(rec.perms) // B
    tabs = []
    for k in range(K):
        tab = torch.arange(T).repeat(B, 1)
        for i in range(B):
            tab[i, :lens[i]] = rec.perms[k * B + i]
        tabs.append(tab)
    return {"neg_idxs": neg, "shuffled_pos_ids": tabs}


def grads_summary(named_params, prefix):
    out = {}
    for k, p in named_params:
        if p.grad is None:
            continue
        g = p.grad.detach()
        out[f"{prefix}gnorm/{k}"] = np.float64(g.double().norm().item())
        if g.numel() <= 4096:
            out[f"{prefix}grad/{k}"] = g.numpy().copy()
        else:
            out[f"{prefix}gprobe/{k}"] = g.flatten()[:: max(1, g.numel() // 257)][:257].numpy().copy()
    return out


def grad_probe(g, n=257):
    """n evenly strided elements of a gradient (zero padded for tensors with fewer): what tests/ compare element-wise"""
    f = g.detach().flatten()
    pr = f[:: max(1, f.numel() // n)][:n]
    out = np.zeros(n, dtype=np.float32)
    out[:pr.numel()] = pr.numpy()
    return out


def grads_packed(named_params, prefix):
    """the same information as grads_summary in three arrays (the canon model has 416 parameters x 6 tasks)"""
    names, norms, probes = [], [], []
    for k, p in named_params:
        if p.grad is None:
            continue
        names.append(k)
        norms.append(p.grad.detach().double().norm().item())
        probes.append(grad_probe(p.grad))
    return {f"{prefix}grad_names": np.array(names), f"{prefix}grad_norms": np.array(norms, dtype=np.float64),
            f"{prefix}grad_probes": np.stack(probes)}


def run_task(model, vil, oracle_sd, cfg, batch, task, tag, store, check):
    # ---- reference
    model.zero_grad(set_to_none=True)
    rec = None
    if task == "itm":
        np.random.seed(1234)
        torch.manual_seed(1234)
        with RngRecorder(vil) as rec:
            loss = model(batch, task, True)
        itm_rng = itm_rng_from_record(rec, batch)
        np.random.seed(1234)
        torch.manual_seed(1234)
        logits = model(batch, task, False)
    else:
        itm_rng = None
        loss = model(batch, task, True)
        logits = model(batch, task, False)
    loss.mean().backward()
    pre = f"{tag}/"
    store.update(to_np({k: v for k, v in batch.items()}, pre + "in/"))
    if itm_rng is not None:
        store.update(to_np(itm_rng, pre + "rng/"))
    store[pre + "loss"] = loss.detach().numpy()
    if isinstance(logits, tuple):
        store[pre + "logits"] = logits[0].detach().numpy()
        store[pre + "targets"] = logits[1].detach().numpy()
    else:
        store[pre + "logits"] = logits.detach().numpy()
    store.update(grads_summary(model.named_parameters(), pre))
    # trunk embeddings (eval) for non-itm tasks
    if task != "itm":
        g = lambda k: batch.get(k)
        with torch.no_grad():
            t, h, o = model.bert(g("txt_ids"), g("txt_masks"), g("hist_img_fts"), g("hist_ang_fts"),
                                 g("hist_pano_img_fts"), g("hist_pano_ang_fts"), g("hist_masks"),
                                 g("ob_img_fts"), g("ob_ang_fts"), g("ob_nav_types"), g("ob_masks"))
        store[pre + "txt_embeds"] = t.numpy()
        store[pre + "hist_embeds"] = h.numpy()
        if o is not None:
            store[pre + "ob_embeds"] = o.numpy()
    # ---- oracle cross-check at generation time
    if check:
        osd = {k: v.clone().requires_grad_(True) for k, v in oracle_sd.items() if k != "mlm_head.predictions.decoder.weight"}
        orc = HamtOracle(osd, cfg, training=False)
        ol = orc.forward(batch, task, True, itm_rng)
        d = (ol - loss.detach()).abs().max().item()
        ol.mean().backward()
        gd = 0.0
        for k, p in model.named_parameters():
            if p.grad is not None and osd[k].grad is not None:
                gd = max(gd, (osd[k].grad - p.grad).abs().max().item())
        print(f"  [{tag}] oracle-vs-reference: loss max|d|={d:.3e}  grad max|d|={gd:.3e}")


def gen_tiny():
    cfg = OracleConfig.tiny(hidden_size=128, num_attention_heads=2, intermediate_size=256, image_feat_size=64)
    sd = make_state_dict(pretrain_param_shapes(cfg), seed=7)
    model, vil = build_ref_pretrain(cfg, sd)
    store = {"meta/sd_sha256": np.array(sd_hash(sd)), "meta/sd_seed": np.array(7)}
    cases = [("mlm", dict(batch_size=3, ragged=True, seed=11)),
             ("sap", dict(batch_size=3, ragged=True, seed=12)),
             ("sap_nohist", dict(batch_size=2, hist_len=0, seed=13)),
             ("sar", dict(batch_size=3, ragged=True, seed=14)),
             ("sprel", dict(batch_size=3, ragged=True, seed=15)),
             ("mrc", dict(batch_size=3, ragged=True, seed=16)),
             ("itm", dict(batch_size=6, ragged=True, seed=17)),
             ("itm_b1", dict(batch_size=2, ragged=False, seed=18))]
    for tag, kw in cases:
        task = tag.split("_")[0]
        batch = make_batch(task, cfg=cfg, txt_len=20, hist_len=kw.pop("hist_len", 4), **kw)
        run_task(model, vil, sd, cfg, batch, task, tag, store, check=True)
    np.savez_compressed(os.path.join(OUT, "tiny_pretrain.npz"), **store)
    print("tiny_pretrain.npz:", len(store), "arrays")


def gen_canon():
    cfg = OracleConfig()
    sd = make_state_dict(pretrain_param_shapes(cfg), seed=2024)
    model, vil = build_ref_pretrain(cfg, sd)
    store = {"meta/sd_sha256": np.array(sd_hash(sd)), "meta/sd_seed": np.array(2024)}
    for i, task in enumerate(("mlm", "sap", "sar", "sprel", "mrc", "itm")):
        batch = make_batch(task, 2 if task != "itm" else 4, cfg, seed=100 + i, txt_len=80, hist_len=5)
        pre = f"{task}/"
        itm_rng = None
        with torch.no_grad():
            if task == "itm":
                np.random.seed(4321)
                torch.manual_seed(4321)
                with RngRecorder(vil) as rec:
                    loss = model(batch, task, True)
                itm_rng = itm_rng_from_record(rec, batch)
                np.random.seed(4321)
                torch.manual_seed(4321)
                logits = model(batch, task, False)
            else:
                loss = model(batch, task, True)
                logits = model(batch, task, False)
        store[pre + "seed"] = np.array(100 + i)
        if itm_rng is not None:
            store.update(to_np(itm_rng, pre + "rng/"))
        store[pre + "loss"] = loss.numpy()
        lg = logits[0] if isinstance(logits, tuple) else logits
        store[pre + "logits"] = lg.numpy() if lg.numel() <= 65536 else lg[:, :: lg.shape[1] // 509][:, :509].numpy().copy()
        if task != "itm":
            g = lambda k: batch.get(k)
            with torch.no_grad():
                t, h, o = model.bert(g("txt_ids"), g("txt_masks"), g("hist_img_fts"), g("hist_ang_fts"),
                                     g("hist_pano_img_fts"), g("hist_pano_ang_fts"), g("hist_masks"),
                                     g("ob_img_fts"), g("ob_ang_fts"), g("ob_nav_types"), g("ob_masks"))
            store[pre + "txt_probe"] = t[:, :4, :32].numpy().copy()
            store[pre + "txt_norm"] = t.norm(dim=-1).numpy()
            store[pre + "hist_embeds"] = h.numpy()
            if o is not None:
                store[pre + "ob_probe"] = o[:, :, :16].numpy().copy()
        # ---- backward at the benchmarked model size (main_r2r.py:237-246: loss.mean().backward()): per-parameter gradient
        # norms and 257-point probes of every gradient from the REFERENCE's autograd (dropout off: the model is in eval mode)
        model.zero_grad(set_to_none=True)
        if task == "itm":
            np.random.seed(4321)
            torch.manual_seed(4321)
        model(batch, task, True).mean().backward()
        store.update(grads_packed(model.named_parameters(), pre))
        osd = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k != "mlm_head.predictions.decoder.weight"}
        orc = HamtOracle(osd, cfg, training=False)
        ol = orc.forward(batch, task, True, itm_rng)
        ol.mean().backward()
        gd, gmax = 0.0, 0.0
        for k, p_ in model.named_parameters():
            if p_.grad is not None:
                assert osd[k].grad is not None, k
                gd = max(gd, (osd[k].grad - p_.grad).abs().max().item())
                gmax = max(gmax, p_.grad.abs().max().item())
            else:
                assert osd[k].grad is None or float(osd[k].grad.abs().max()) == 0.0, k
        model.zero_grad(set_to_none=True)
        print(f"  [canon {task}] oracle-vs-reference loss max|d|={(ol.detach() - loss).abs().max().item():.3e}  grad max|d|={gd:.3e} (max |g| {gmax:.3e})")
    np.savez_compressed(os.path.join(OUT, "canon_pretrain.npz"), **store)
    print("canon_pretrain.npz:", len(store), "arrays")


def gen_optim():
    """Reference AdamW (optim/adamw.py) + grouping (optim/misc.py:12-22) + schedule (optim/sched.py) +
    clip (main_r2r.py:271) for 3 steps of SAP on the tiny model, dropout off."""
    import importlib
    sys.path.insert(0, os.path.join(ref_shim.REF, "pretrain_src"))
    for k in [k for k in sys.modules if k == "optim" or k.startswith("optim.")]:
        del sys.modules[k]
    try:
        ref_adamw = importlib.import_module("optim.adamw")
        ref_sched = importlib.import_module("optim.sched")
    finally:
        sys.path.pop(0)
    cfg = OracleConfig.tiny(hidden_size=128, num_attention_heads=2, intermediate_size=256, image_feat_size=64)
    sd = make_state_dict(pretrain_param_shapes(cfg), seed=7)
    model, _ = build_ref_pretrain(cfg, sd)
    named = list(model.named_parameters())
    no_decay = ['bias', 'LayerNorm.bias', 'LayerNorm.weight']
    groups = [{'params': [p for n, p in named if not any(nd in n for nd in no_decay)], 'weight_decay': 0.01},
              {'params': [p for n, p in named if any(nd in n for nd in no_decay)], 'weight_decay': 0.0}]
    opt = ref_adamw.AdamW(groups, lr=5e-3, betas=(0.9, 0.98))

    class O:  # opts namespace for get_lr_sched
        learning_rate, warmup_steps, num_train_steps = 5e-3, 2, 10
    store = {"meta/decay_names": np.array([n for n, _ in named if not any(nd in n for nd in no_decay)])}
    opt.zero_grad()
    opt.step()
    probe_keys = ["bert.encoder.x_layers.0.visual_attention.att.query.weight", "bert.embeddings.LayerNorm.weight",
                  "bert.img_embeddings.layer_norm.weight", "next_action.net.2.weight", "next_action.net.4.bias",
                  "bert.hist_embeddings.cls_token", "bert.encoder.layer.1.output.dense.bias"]
    for step in range(1, 4):
        batch = make_batch("sap", 3, cfg, seed=50 + step, txt_len=20, hist_len=4, ragged=True)
        loss = model(batch, "sap", True).mean()
        loss.backward()
        lr = ref_sched.get_lr_sched(step, O)
        assert abs(lr - lr_at(step, 5e-3, 2, 10)) < 1e-15
        for g in opt.param_groups:
            g['lr'] = lr
        gn = torch.nn.utils.clip_grad_norm_(model.parameters(), 5.0)
        opt.step()
        opt.zero_grad()
        store[f"step{step}/loss"] = np.float64(loss.item())
        store[f"step{step}/grad_norm"] = np.float64(gn.item())
        store[f"step{step}/lr"] = np.float64(lr)
        for k in probe_keys:
            store[f"step{step}/param/{k}"] = dict(named)[k].detach().numpy().copy()
        store[f"step{step}/pnorm_total"] = np.float64(
            torch.sqrt(sum(p.detach().double().pow(2).sum() for _, p in named)).item())
    np.savez_compressed(os.path.join(OUT, "optim_tiny.npz"), **store)
    print("optim_tiny.npz:", len(store), "arrays")


def gen_finetune():
    ft = ref_shim.import_finetune()
    store = {}
    for tag, extra in (("ca", dict(no_lang_ca=False, act_pred_token="ob_txt")),
                       ("nolangca", dict(no_lang_ca=True, act_pred_token="ob")),
                       ("obhist", dict(no_lang_ca=False, act_pred_token="ob_hist")),            # vilmodel_cmt.py:722-723
                       ("obtxthist", dict(no_lang_ca=False, act_pred_token="ob_txt_hist"))):    # vilmodel_cmt.py:724-725
        cfg = OracleConfig.tiny(hidden_size=128, num_attention_heads=2, intermediate_size=256, image_feat_size=64, **extra)
        sd = make_state_dict(navcmt_param_shapes(cfg), seed=9)
        rcfg = ref_shim.make_config(cfg, output_attentions=True)
        model = ft.NavCMT(rcfg)
        model.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
        model.eval()
        b = make_batch("sap", 4, cfg, seed=31, txt_len=24, hist_len=3, ragged=True)
        orc = HamtOracle(sd, cfg, training=False)
        with torch.no_grad():
            lang = model("language", txt_ids=b["txt_ids"], txt_masks=b["txt_masks"])
            olang = orc.ft_forward("language", txt_ids=b["txt_ids"], txt_masks=b["txt_masks"])
            h_cls = model("history")
            hs = [h_cls.expand(4, -1)]
            ohs = [orc.ft_forward("history").expand(4, -1)]
            for t in range(3):
                kw = dict(hist_img_feats=b["hist_img_fts"][:, t], hist_ang_feats=b["hist_ang_fts"][:, t],
                          ob_step_ids=torch.LongTensor([t]), hist_pano_img_feats=b["hist_pano_img_fts"][:, t],
                          hist_pano_ang_feats=b["hist_pano_ang_fts"][:, t])
                hs.append(model("history", **kw))
                ohs.append(orc.ft_forward("history", **kw))
            hist = torch.stack(hs, 1)
            vkw = dict(txt_masks=b["txt_masks"], hist_masks=b["hist_masks"], ob_img_feats=b["ob_img_fts"],
                       ob_ang_feats=b["ob_ang_fts"], ob_nav_types=b["ob_nav_types"], ob_masks=b["ob_masks"])
            out = model("visual", txt_embeds=lang, hist_embeds=hist, **vkw)
            oout = orc.ft_forward("visual", txt_embeds=olang, hist_embeds=torch.stack(ohs, 1), **vkw)
        pre = f"{tag}/"
        store.update(to_np(b, pre + "in/"))
        if isinstance(lang, list):
            for i, t in enumerate(lang):
                store[pre + f"lang.{i}"] = t.numpy()
        else:
            store[pre + "lang"] = lang.numpy()
        store[pre + "hist"] = hist.numpy()
        for n, t in zip(("act_logits", "txt", "hist_out", "ob_out"), out):
            store[pre + n] = t.numpy()
        d = max((a - c).abs()[torch.isfinite(a)].max().item() for a, c in zip(out, oout))
        print(f"  [finetune {tag}] oracle-vs-reference visual max|d|={d:.3e}")
    gen_agent_models(store)
    np.savez_compressed(os.path.join(OUT, "tiny_finetune.npz"), **store)
    print("tiny_finetune.npz:", len(store), "arrays")


def gen_agent_models(store):
    """Row A23: the reference's own VLNBertCMT.forward (model_HAMT.py:20-65) and Critic (:258-269).  VLNBertCMT is built
    around an already constructed reference NavCMT (its __init__ would fetch bert-base-uncased's config from the hub and go
    through HF 4.12's from_pretrained, neither of which exists here), then driven exactly as the agent drives it
    (agent_cmt.py:270-397): language once, history cls, one history step per time step with `ob_step`, visual with the LIST
    of history embeddings, ragged `hist_lens` and return_states=True.  eval mode: drop_env is the identity."""
    import types
    ft, mh = ref_shim.import_finetune_agent_models()
    for tag, no_lang_ca in (("agent_ca", False), ("agent_nolangca", True)):
        cfg = OracleConfig.tiny(hidden_size=128, num_attention_heads=2, intermediate_size=256, image_feat_size=64,
                                no_lang_ca=no_lang_ca, act_pred_token="ob" if no_lang_ca else "ob_txt")
        sd = make_state_dict(navcmt_param_shapes(cfg), seed=9)
        nav = ft.NavCMT(ref_shim.make_config(cfg, output_attentions=True))
        nav.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
        agent = mh.VLNBertCMT.__new__(mh.VLNBertCMT)
        torch.nn.Module.__init__(agent)
        agent.args = types.SimpleNamespace(no_lang_ca=no_lang_ca, feat_dropout=0.4)
        agent.vln_bert = nav
        agent.drop_env = torch.nn.Dropout(p=0.4)
        agent.eval()
        b = make_batch("sap", 4, cfg, seed=33, txt_len=24, hist_len=3, ragged=False)
        hist_lens = [4, 2, 3, 1]                       # valid history tokens per sample (cls + steps taken), agent_cmt.py:312
        with torch.no_grad(), ref_shim.cuda_is_identity():
            lang = agent("language", txt_ids=b["txt_ids"], txt_masks=b["txt_masks"])
            hs = [agent("history").expand(4, -1)]
            for t in range(3):
                hs.append(agent("history", hist_img_feats=b["hist_img_fts"][:, t], hist_ang_feats=b["hist_ang_fts"][:, t],
                                hist_pano_img_feats=b["hist_pano_img_fts"][:, t], hist_pano_ang_feats=b["hist_pano_ang_fts"][:, t], ob_step=t))
            logits, states = agent("visual", txt_embeds=lang, txt_masks=b["txt_masks"], hist_embeds=hs, hist_lens=hist_lens,
                                   ob_img_feats=b["ob_img_fts"], ob_ang_feats=b["ob_ang_fts"], ob_nav_types=b["ob_nav_types"],
                                   ob_masks=b["ob_masks"], return_states=True)
            (logits_only,) = agent("visual", txt_embeds=lang, txt_masks=b["txt_masks"], hist_embeds=hs, hist_lens=hist_lens,
                                   ob_img_feats=b["ob_img_fts"], ob_ang_feats=b["ob_ang_fts"], ob_nav_types=b["ob_nav_types"], ob_masks=b["ob_masks"])
            mask = mh.length2mask(hist_lens, size=4)
        assert torch.equal(logits, logits_only)
        pre = f"{tag}/"
        store.update(to_np(b, pre + "in/"))
        store[pre + "hist_lens"] = np.array(hist_lens)
        store[pre + "length2mask"] = mask.numpy()
        store[pre + "hist"] = torch.stack(hs, 1).numpy()
        store[pre + "act_logits"] = logits.numpy()
        store[pre + "states"] = states.numpy()
        print(f"  [finetune {tag}] VLNBertCMT.forward: logits {tuple(logits.shape)}, states {tuple(states.shape)}")
    critic = mh.Critic(types.SimpleNamespace(dropout=0.5))
    csd = make_state_dict({"state2value.0.weight": (512, 768), "state2value.0.bias": (512,), "state2value.3.weight": (1, 512), "state2value.3.bias": (1,)}, seed=12)
    critic.load_state_dict(csd, strict=True)
    critic.eval()
    g = torch.Generator().manual_seed(5)
    st = torch.randn(8, 768, generator=g)
    with torch.no_grad():
        val = critic(st)
    store["critic/state"] = st.numpy()
    store["critic/value"] = val.numpy()
    store["critic/sd_seed"] = np.array(12)


def _reference_a2c_block():
    """The A2C statements of the agent's rollout (finetune_src/r2r/agent_cmt.py, from `rl_loss = 0.` to `self.loss += rl_loss`) as a
    code object compiled FROM THE REFERENCE FILE at generation time -- the reference's own statements, not a restatement.  They sit
    inline in a 280-line method that needs the Matterport simulator for everything in front of them; the block itself only reads the
    rollout's per-step lists, which are scripted here."""
    import textwrap
    path = os.path.join(ref_shim.REF, "finetune_src", "r2r", "agent_cmt.py")
    lines = open(path).read().split("\n")
    lo = next(i for i, ln in enumerate(lines) if ln.strip() == "rl_loss = 0.")
    hi = next(i for i, ln in enumerate(lines) if i > lo and ln.strip() == "self.loss += rl_loss")
    src = textwrap.dedent("\n".join(lines[lo:hi + 1]))
    assert "last_value__ = self.critic(last_h_).detach()" in src and "normalize_loss" in src, "agent_cmt.py changed"
    return compile(src, f"{path}:{lo + 1}-{hi + 1}", "exec"), (lo + 1, hi + 1)


def gen_a2c():
    """The agent's A2C loss (agent_cmt.py:473-515) pinned: the reference's own statements (compiled from its file, `_reference_a2c_block`)
    run on scripted rollout lists -- per-step policy log-probabilities, hidden states (through the reference's Critic), rewards, masks,
    entropies, a last state, `ended` -- for the three normalisations and both feedback modes; stored: the inputs, rl_loss, the logged
    sums and the gradients w.r.t. the log-probabilities, the hidden states, the entropies and every critic parameter."""
    import types
    from collections import defaultdict
    _, mh = ref_shim.import_finetune_agent_models()
    code, span = _reference_a2c_block()
    store = {"meta/span": np.asarray(span)}
    T, B, Hs = 6, 5, 768
    critic = mh.Critic(types.SimpleNamespace(dropout=0.5))
    csd = make_state_dict({"state2value.0.weight": (512, 768), "state2value.0.bias": (512,), "state2value.3.weight": (1, 512), "state2value.3.bias": (1,)}, seed=12)
    critic.load_state_dict(csd, strict=True)
    critic.eval()
    store["meta/critic_seed"] = np.array(12)
    rng = np.random.Generator(np.random.PCG64(77))
    ended_at = np.array([3, 6, 8, 5, 9])                              # episodes 2 and 4 run past the rollout: not ended
    masks = (np.arange(T)[:, None] < ended_at[None]).astype(np.float32)
    rewards = (rng.standard_normal((T, B)).astype(np.float32) * 2.0) * masks
    ended = ended_at <= T
    base = {"logp": -rng.random((T, B), dtype=np.float32) * 3.0, "hidden": rng.standard_normal((T, B, Hs), dtype=np.float32) * 0.5,
            "ent": rng.random((T, B), dtype=np.float32), "last_h": rng.standard_normal((B, Hs), dtype=np.float32) * 0.5}
    store.update({"in/rewards": rewards, "in/masks": masks, "in/ended": ended, **{f"in/{k}": v for k, v in base.items()}})
    for normalize in ("total", "batch", "none"):
        for feedback in ("sample", "teacher"):
            logp = torch.from_numpy(base["logp"]).requires_grad_(True)
            hidden = torch.from_numpy(base["hidden"]).requires_grad_(True)
            ent = torch.from_numpy(base["ent"]).requires_grad_(True)
            critic.zero_grad(set_to_none=True)
            me = types.SimpleNamespace(critic=critic, feedback=feedback, logs=defaultdict(list), loss=0.0,
                                       args=types.SimpleNamespace(gamma=0.9, entropy_loss_weight=0.01, normalize_loss=normalize))
            ns = {"self": me, "np": np, "torch": torch, "last_h_": torch.from_numpy(base["last_h"]), "batch_size": B, "ended": ended.copy(),
                  "rewards": [rewards[t] for t in range(T)], "masks": [masks[t] for t in range(T)],
                  "policy_log_probs": [logp[t] for t in range(T)], "hidden_states": [hidden[t] for t in range(T)],
                  "entropys": [ent[t] for t in range(T)]}
            with ref_shim.cuda_is_identity():
                exec(code, ns)
            loss = me.loss
            loss.backward()
            pre = f"{normalize}_{feedback}/"
            store[pre + "rl_loss"] = np.float64(loss.item())
            store[pre + "policy_sum"] = np.float64(sum(me.logs["policy_loss"]))
            store[pre + "critic_sum"] = np.float64(sum(me.logs["critic_loss"]))
            store[pre + "total"] = np.float64(me.logs["total"][0])
            store[pre + "d_logp"] = logp.grad.numpy().copy()
            store[pre + "d_hidden"] = hidden.grad.numpy().copy()
            store[pre + "d_ent"] = ent.grad.numpy().copy() if ent.grad is not None else np.zeros((T, B), np.float32)
            for k, p_ in critic.named_parameters():
                if p_.numel() <= 4096:
                    store[pre + "d_critic/" + k] = p_.grad.numpy().copy()
                else:      # (the 512 x 768 weight: norm + 257-point probe, as for the model's large gradients)
                    store[pre + "d_critic_norm/" + k] = np.float64(p_.grad.double().norm().item())
                    store[pre + "d_critic_probe/" + k] = grad_probe(p_.grad)
            print(f"  [a2c {normalize} {feedback}] rl_loss {loss.item():.6f}")
    np.savez_compressed(os.path.join(OUT, "a2c.npz"), **store)
    print(f"a2c.npz: {len(store)} arrays from agent_cmt.py:{span[0]}-{span[1]}")


def gen_vit():
    """ViT backbone (row N3): the reference's own VisionTransformer class (imported through ref_shim.import_vit) on a
    tiny config (features + every parameter gradient of sum(feats * probe)) and on ViT-B/16 at 224x224 (features only)."""
    from oracle.hamt_oracle import VitConfig, make_vit_state_dict, vit_forward_features
    vt = ref_shim.import_vit()
    store = {}
    for tag, c, n_img in (("tiny", VitConfig.tiny(), 3), ("b16", VitConfig(), 2)):
        sd = make_vit_state_dict(c, seed=21)
        ref = vt.VisionTransformer(img_size=c.img_size, patch_size=c.patch_size, in_chans=c.in_chans, num_classes=0,
                                   embed_dim=c.embed_dim, depth=c.depth, num_heads=c.num_heads, mlp_ratio=c.mlp_ratio, qkv_bias=True)
        ref.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
        ref.eval()
        rng = np.random.Generator(np.random.PCG64(5))
        imgs = torch.from_numpy(rng.standard_normal((n_img, c.in_chans, c.img_size, c.img_size), dtype=np.float32))
        probe = torch.from_numpy(rng.standard_normal((n_img, c.embed_dim), dtype=np.float32))
        feats = ref.forward_features(imgs)
        osd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        ofeats = vit_forward_features(osd, c, imgs)
        print(f"  [vit {tag}] oracle-vs-reference features max|d|={(feats - ofeats).abs().max().item():.3e}")
        store[f"{tag}/images"] = imgs.numpy() if tag == "tiny" else imgs[:, :, :8, :8].numpy()      # b16 inputs are regenerated
        store[f"{tag}/probe"] = probe.numpy()
        store[f"{tag}/feats"] = feats.detach().numpy()
        (feats * probe).sum().backward()
        (ofeats * probe).sum().backward()
        worst = 0.0
        for k, p_ in ref.named_parameters():
            if tag == "tiny":
                store[f"{tag}/grad/{k}"] = p_.grad.numpy()
            worst = max(worst, (p_.grad - osd[k].grad).abs().max().item())
        if tag == "b16":      # config 4's backbone at full size: per-parameter gradient norms + 257-point probes of the reference's autograd
            store.update(grads_packed(ref.named_parameters(), "b16/"))
        print(f"  [vit {tag}] oracle-vs-reference gradients max|d|={worst:.3e}")
    np.savez_compressed(os.path.join(OUT, "vit.npz"), **store)
    print("vit.npz:", len(store), "arrays")


COLLATE_CASES = [("mlm", 5, 11, False), ("mrc", 4, 12, False), ("itm", 3, 13, False), ("sap", 5, 14, False), ("sap", 3, 15, True),
                 ("sar", 4, 16, False), ("sprel", 4, 17, False), ("sprel", 2, 18, True)]
COLLATE_DIMS = dict(feat=8, ang=4, prob=10, max_txt=12, max_hist=4, views=36)


def gen_collate():
    """Row N4 (batch collation): the reference's own six *_collate functions (r2r_tasks.py) on seeded ragged samples from
    vln_hamt_amd.synth.make_samples (small feature widths; incl. the all-first-step `hist = None` branch).  The fixture
    holds the expected outputs; the inputs are regenerated from the recipe (task, n, seed, COLLATE_DIMS)."""
    from vln_hamt_amd.synth import make_samples
    from oracle.collate_oracle import COLLATE
    ref = ref_shim.import_collate()
    store = {}
    for task, n, seed, first in COLLATE_CASES:
        tag = f"{task}{seed}"
        exp = getattr(ref, f"{task}_collate")(make_samples(task, n, seed, first_step=first, **COLLATE_DIMS))
        got = COLLATE[task](make_samples(task, n, seed, first_step=first, **COLLATE_DIMS))
        assert set(exp) == set(got), (set(exp) ^ set(got))
        for k, v in exp.items():
            if v is None:
                assert got[k] is None, k
                store[f"{tag}/{k}/none"] = np.zeros(0)
            elif torch.is_tensor(v):
                a = v.numpy()
                assert a.dtype == got[k].dtype and np.array_equal(a, got[k]), (tag, k, a.dtype, got[k].dtype)   # oracle == reference, bit exact
                store[f"{tag}/{k}"] = a
            else:                                   # keys the collate leaves as python lists
                store[f"{tag}/{k}/list"] = np.asarray(len(v))
    np.savez_compressed(os.path.join(OUT, "collate.npz"), **store)
    print("collate.npz:", len(store), "arrays; oracle restatement bit-exact on", len(COLLATE_CASES), "cases")


# (weights seed, batch seed base, per-GPU batch): the second and third weight draws at the golden batch size, and the reference's own
# per-GPU batch (16, pretrain_r2r.json) on two of them -- bf16 margins measured on ONE seed at B = 2 say little
CANON_MULTI = [(7, 300, 2), (99, 500, 2), (2024, 700, 16), (7, 900, 16)]
MULTI_PROBE = 65
# (round 6) the BENCHMARKED per-GPU batch from the reference itself: the reference's goldens stopped at its own batch 16, the B = 64 model-level
# comparison ran against the pinned oracle only.  Same content as canon_multi / canon_multi_sar, one draw, files canon_b64*.npz
# (the history embeddings as a 64-column probe: the full tensor would be 1.2 MB per task).
CANON_B64 = [(5, 6400, 64)]


def grads_packed_n(named_params, prefix, n):
    names, norms, probes = [], [], []
    for k, p in named_params:
        if p.grad is None:
            continue
        names.append(k)
        norms.append(p.grad.detach().double().norm().item())
        probes.append(grad_probe(p.grad, n))
    return {f"{prefix}grad_names": np.array(names), f"{prefix}grad_norms": np.array(norms, dtype=np.float64), f"{prefix}grad_probes": np.stack(probes)}


def gen_canon_multi(cases=None, out="canon_multi.npz", hist_probe=False):
    """R2R-canon again over CANON_MULTI (or `cases`): per task the loss, trunk probes, per-parameter gradient norms and 65-point gradient
    probes from the REFERENCE; for ITM additionally the gradients of single candidate logits (sum_b logits[b, k], k = 0 positive and
    k = 3 a shuffled negative): the loss gradient is a difference of five nearly equal such terms, so its cosine measures
    cancellation noise -- the terms themselves are gated like every other task."""
    torch.set_num_threads(8)
    cases = CANON_MULTI if cases is None else cases
    store = {"meta/cases": np.asarray(cases)}
    cfg = OracleConfig()
    models = {}
    for ci, (wseed, bseed, B) in enumerate(cases):
        if wseed not in models:
            sd = make_state_dict(pretrain_param_shapes(cfg), seed=wseed)
            models[wseed] = build_ref_pretrain(cfg, sd)
        model, vil = models[wseed]
        for i, task in enumerate(("mlm", "sap", "sar", "sprel", "mrc", "itm")):
            batch = make_batch(task, B if task != "itm" else 2 * B, cfg, seed=bseed + i, txt_len=80, hist_len=5)
            pre = f"c{ci}/{task}/"
            itm_rng = None
            with torch.no_grad():
                if task == "itm":
                    np.random.seed(4321 + ci)
                    torch.manual_seed(4321 + ci)
                    with RngRecorder(vil) as rec:
                        loss = model(batch, task, True)
                    itm_rng = itm_rng_from_record(rec, batch)
                    store.update(to_np(itm_rng, pre + "rng/"))
                else:
                    loss = model(batch, task, True)
                    g = lambda k: batch.get(k)
                    t, h, o = model.bert(g("txt_ids"), g("txt_masks"), g("hist_img_fts"), g("hist_ang_fts"), g("hist_pano_img_fts"),
                                         g("hist_pano_ang_fts"), g("hist_masks"), g("ob_img_fts"), g("ob_ang_fts"), g("ob_nav_types"), g("ob_masks"))
                    store[pre + "txt_probe"] = t[:, :4, :32].numpy().copy()
                    if hist_probe:
                        store[pre + "hist_probe"] = h[:, :, :64].numpy().copy()
                    else:
                        store[pre + "hist_embeds"] = h.numpy()
                    if o is not None:
                        store[pre + "ob_probe"] = o[:, :, :16].numpy().copy()
            store[pre + "loss"] = loss.numpy()
            model.zero_grad(set_to_none=True)
            if task == "itm":
                np.random.seed(4321 + ci)
                torch.manual_seed(4321 + ci)
            model(batch, task, True).mean().backward()
            store.update(grads_packed_n(model.named_parameters(), pre, MULTI_PROBE))
            if task == "itm":
                for k in (0, 3):
                    model.zero_grad(set_to_none=True)
                    np.random.seed(4321 + ci)
                    torch.manual_seed(4321 + ci)
                    lg = model(batch, task, False)
                    lg = lg[0] if isinstance(lg, tuple) else lg
                    store[pre + "logits"] = lg.detach().numpy()
                    (lg[:, k].sum() / lg.shape[0]).backward()
                    store.update(grads_packed_n(model.named_parameters(), pre + f"logit{k}/", MULTI_PROBE))
            model.zero_grad(set_to_none=True)
            print(f"  [canon multi c{ci} w{wseed} B{B} {task}] loss {loss.mean().item():.4f}", flush=True)
    np.savez_compressed(os.path.join(OUT, out), **store)
    print(out + ":", len(store), "arrays")


def gen_canon_multi_sar(cases=None, out="canon_multi_sar.npz"):
    """SAR on the CANON_MULTI draws (or `cases`), term by term: the loss gradient is sum_{b,k} (2 r_bk / 3B) d pred_bk, a sum over samples and the
    three regression outputs whose residuals r have both signs -- on one B = 16 draw it cancels far enough that bf16 rounding of the
    TERMS shows in the cosine of the SUM.  Stored: the reference's predictions, targets and the gradients of the three single-output
    means d(mean_b pred[b, k]) (65-point probes + norms), which tests/ gate like every other gradient."""
    torch.set_num_threads(8)
    cases = CANON_MULTI if cases is None else cases
    store = {"meta/cases": np.asarray(cases)}
    cfg = OracleConfig()
    models = {}
    for ci, (wseed, bseed, B) in enumerate(cases):
        if wseed not in models:
            models[wseed] = build_ref_pretrain(cfg, make_state_dict(pretrain_param_shapes(cfg), seed=wseed))
        model, vil = models[wseed]
        batch = make_batch("sar", B, cfg, seed=bseed + 2, txt_len=80, hist_len=5)       # (gen_canon_multi: seed = bseed + task index)
        pre = f"c{ci}/sar/"
        for k in range(3):
            model.zero_grad(set_to_none=True)
            pred = model(batch, "sar", False)
            store[pre + "logits"] = pred.detach().numpy()
            (pred[:, k].sum() / pred.shape[0]).backward()
            store.update(grads_packed_n(model.named_parameters(), pre + f"out{k}/", MULTI_PROBE))
        store[pre + "targets"] = torch.cat([batch["ob_action_angles"], batch["ob_progress"].unsqueeze(1)], 1).numpy()
        model.zero_grad(set_to_none=True)
        print(f"  [canon multi sar terms c{ci} w{wseed} B{B}]", flush=True)
    np.savez_compressed(os.path.join(OUT, out), **store)
    print(out + ":", len(store), "arrays")


# ------------------------------------------------------------------------------------------------ ragged batches (the packed text path)
# (weight seed, batch seed, per-GPU batch): SURVEY 8d's ragged variant -- L ~ U[20, 80], T ~ U[0, 7] -- at the reference's per-GPU batch.
# make_batch(ragged=True) also emits the text packing plan (txt_pack_idx / txt_cu / txt_unpack_idx); the reference ignores those keys,
# the HIP model under test consumes them, so tests/ compare the PACKED computation with the reference directly.
CANON_RAGGED = [(2024, 1100, 16), (7, 1300, 16)]
RAGGED_L, RAGGED_T = 80, 7


def gen_canon_ragged():
    """R2R-canon on ragged batches: per task the loss, the text embeddings (every 24th channel of every token + the per-token norms; tests
    compare them at the REAL positions), history / observation probes, per-parameter gradient norms + 65-point probes from the REFERENCE's
    forward and autograd, the ITM draws made inside forward_itm and the gradients of single candidate logits."""
    torch.set_num_threads(8)
    store = {"meta/cases": np.asarray(CANON_RAGGED), "meta/txt_len": np.array(RAGGED_L), "meta/hist_len": np.array(RAGGED_T)}
    cfg = OracleConfig()
    for ci, (wseed, bseed, B) in enumerate(CANON_RAGGED):
        sd = make_state_dict(pretrain_param_shapes(cfg), seed=wseed)
        model, vil = build_ref_pretrain(cfg, sd)
        for i, task in enumerate(("mlm", "sap", "sar", "sprel", "mrc", "itm")):
            batch = make_batch(task, B if task != "itm" else 2 * B, cfg, seed=bseed + i, txt_len=RAGGED_L, hist_len=RAGGED_T, ragged=True)
            assert "txt_pack_idx" in batch, "this draw has (almost) no padding: pick another seed"
            pre = f"c{ci}/{task}/"
            store[pre + "txt_lens"] = batch["txt_masks"].sum(1).numpy()
            store[pre + "hist_lens"] = (batch["hist_masks"].sum(1) - 1).numpy()
            with torch.no_grad():
                if task == "itm":
                    np.random.seed(4400 + ci)
                    torch.manual_seed(4400 + ci)
                    with RngRecorder(vil) as rec:
                        loss = model(batch, task, True)
                    store.update(to_np(itm_rng_from_record(rec, batch), pre + "rng/"))
                else:
                    loss = model(batch, task, True)
                    g = lambda k: batch.get(k)
                    t, h, o = model.bert(g("txt_ids"), g("txt_masks"), g("hist_img_fts"), g("hist_ang_fts"), g("hist_pano_img_fts"),
                                         g("hist_pano_ang_fts"), g("hist_masks"), g("ob_img_fts"), g("ob_ang_fts"), g("ob_nav_types"), g("ob_masks"))
                    store[pre + "txt_sel"] = t[:, :, ::24].numpy().copy()
                    store[pre + "txt_norm"] = t.norm(dim=-1).numpy()
                    store[pre + "hist_sel"] = h[:, :, ::4].numpy().copy()
                    if o is not None:
                        store[pre + "ob_probe"] = o[:, :, :16].numpy().copy()
            store[pre + "loss"] = loss.numpy()
            model.zero_grad(set_to_none=True)
            if task == "itm":
                np.random.seed(4400 + ci)
                torch.manual_seed(4400 + ci)
            model(batch, task, True).mean().backward()
            store.update(grads_packed_n(model.named_parameters(), pre, MULTI_PROBE))
            if task == "itm":
                for k in (0, 3):
                    model.zero_grad(set_to_none=True)
                    np.random.seed(4400 + ci)
                    torch.manual_seed(4400 + ci)
                    lg = model(batch, task, False)
                    lg = lg[0] if isinstance(lg, tuple) else lg
                    store[pre + "logits"] = lg.detach().numpy()
                    (lg[:, k].sum() / lg.shape[0]).backward()
                    store.update(grads_packed_n(model.named_parameters(), pre + f"logit{k}/", MULTI_PROBE))
            model.zero_grad(set_to_none=True)
            print(f"  [canon ragged c{ci} w{wseed} B{B} {task}] loss {loss.mean().item():.4f}  text rows {int(batch['txt_masks'].sum())} of "
                  f"{batch['txt_masks'].numel()} (packed {batch['txt_pack_idx'].numel()})", flush=True)
        del model
    np.savez_compressed(os.path.join(OUT, "canon_ragged.npz"), **store)
    print("canon_ragged.npz:", len(store), "arrays")


# ------------------------------------------------------------------------------------------------ the reference's own bf16 error
def gen_canon_autocast():
    """The yardstick for the bf16 tolerances: the REFERENCE itself under torch.autocast(bfloat16) against its own fp32 forward, on the
    CANON_MULTI and CANON_RAGGED draws -- max |bf16 - fp32| / max(1, max |fp32|) of the loss, the head outputs (logits / predictions; at the
    finite positions) and the trunk probes the other goldens hold.  Head outputs of the reference's own bf16 path reach 1.6e-2 on these
    draws (a LayerNorm + Linear on top of a 13-layer bf16 trunk amplifies), so tests/ gate the HIP model's head outputs at
    max(1e-2, this) -- "1e-2, or no worse than the reference's own bf16 path on the same draw" -- and everything else at 1e-2."""
    torch.set_num_threads(8)
    cfg = OracleConfig()
    store = {}

    def rel(a, b):
        a, b = a.float(), b.float()
        return float((a - b).abs().max()) / max(1.0, float(b.abs().max()))

    for fam, cases, seed0, kw in (("multi", CANON_MULTI, 4321, dict(txt_len=80, hist_len=5)),
                                  ("ragged", CANON_RAGGED, 4400, dict(txt_len=RAGGED_L, hist_len=RAGGED_T, ragged=True))):
        for ci, (wseed, bseed, B) in enumerate(cases):
            model, vil = build_ref_pretrain(cfg, make_state_dict(pretrain_param_shapes(cfg), seed=wseed))
            for i, task in enumerate(("mlm", "sap", "sar", "sprel", "mrc", "itm")):
                batch = make_batch(task, B if task != "itm" else 2 * B, cfg, seed=bseed + i, **kw)
                outs = {}
                for mode in ("fp32", "bf16"):
                    with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16, enabled=mode == "bf16"):
                        np.random.seed(seed0 + ci)
                        torch.manual_seed(seed0 + ci)
                        r = {"loss": model(batch, task, True)}
                        np.random.seed(seed0 + ci)
                        torch.manual_seed(seed0 + ci)
                        lg = model(batch, task, False)
                        r["logits"] = lg[0] if isinstance(lg, tuple) else lg
                        if task != "itm":
                            g = lambda k: batch.get(k)
                            t, h, o = model.bert(g("txt_ids"), g("txt_masks"), g("hist_img_fts"), g("hist_ang_fts"), g("hist_pano_img_fts"),
                                                 g("hist_pano_ang_fts"), g("hist_masks"), g("ob_img_fts"), g("ob_ang_fts"), g("ob_nav_types"), g("ob_masks"))
                            r["txt"] = t * batch["txt_masks"].unsqueeze(-1)          # (the real positions: nothing reads the others)
                            r["hist"] = h
                            if o is not None:
                                r["ob"] = o
                    outs[mode] = r
                errs = {}
                for k, b in outs["fp32"].items():
                    a = outs["bf16"][k]
                    if k == "logits":
                        fin = torch.isfinite(b)
                        a, b = a[fin], b[fin]
                    errs[k] = rel(a, b)
                    store[f"{fam}/c{ci}/{task}/{k}"] = np.float32(errs[k])
                print(f"  [reference autocast {fam} c{ci} w{wseed} B{B} {task}] " + " ".join(f"{k} {v:.2e}" for k, v in errs.items()), flush=True)
            del model
    np.savez_compressed(os.path.join(OUT, "canon_autocast.npz"), **store)
    print("canon_autocast.npz:", len(store), "arrays")


# ------------------------------------------------------------------------------------------------ N4: readers + loaders
R2R_TINY = os.path.join(OUT, "r2r_tiny")
R2R_DIMS = dict(image_feat_size=16, image_prob_size=10, angle_feat_size=4)
# get_input cases: (trajectory, instruction, t_cur, return_ob, return_hist_img_probs, return_ob_action, return_ob_progress, ob_cand_pano_view)
R2R_CASES = [(0, 0, 0, True, False, True, True, False), (0, 1, 2, True, True, True, True, False), (0, 0, 3, True, False, True, True, True),
             (1, 0, 1, False, True, False, False, None), (1, 0, 4, True, True, True, True, False), (1, 0, 4, True, False, True, True, True),
             (2, 0, 2, True, False, True, True, True), (2, 1, 3, True, True, True, True, False), (3, 0, 1, True, False, True, True, False),
             (3, 0, 0, True, False, True, False, True)]


def make_r2r_tiny():
    """Write the tiny R2R-style dataset (data, not code): two scans' connectivity files, candidate views per viewpoint, four
    trajectories (one with a `guide_path`, one longer than max_act_len, one with an over-long instruction) and the view features
    (float64 [36, feat + prob] per `scan_viewpoint`, the precompute script's dtype).  Deterministic (numpy PCG64)."""
    import json
    rng = np.random.Generator(np.random.PCG64(11))
    os.makedirs(R2R_TINY, exist_ok=True)
    scans = {"scanA": 7, "scanB": 5}
    feats, cands, conn = {}, {}, {}
    for scan, n in scans.items():
        vps = [f"{scan[-1].lower()}{k:02d}" for k in range(n)]
        xyz = rng.uniform(-6, 6, size=(n + 1, 3))
        link = np.zeros((n + 1, n + 1), dtype=bool)
        for k in range(n - 1):                       # a chain plus a few chords; the extra node is excluded from the graph
            link[k, k + 1] = link[k + 1, k] = True
        for a, b in ((0, 2), (1, 4), (2, n - 1)):
            link[a, b] = link[b, a] = True
        link[n, 0] = link[0, n] = True
        nodes = []
        for k in range(n + 1):
            pose = [0.0] * 16
            pose[3], pose[7], pose[11] = (float(v) for v in xyz[k])
            nodes.append({"image_id": vps[k] if k < n else "excluded", "pose": pose, "included": k < n, "unobstructed": [bool(b) for b in link[k]]})
        conn[scan] = nodes
        for k, vp in enumerate(vps):
            feats[f"{scan}_{vp}"] = rng.standard_normal((36, R2R_DIMS["image_feat_size"] + R2R_DIMS["image_prob_size"]))
            nb = [j for j in range(n) if link[k, j]]
            views = rng.choice(36, size=len(nb), replace=False)
            cands[f"{scan}_{vp}"] = {vps[j]: [int(v), float(rng.uniform(1, 4)), float(rng.uniform(-0.3, 0.3)), float(rng.uniform(-0.2, 0.2))]
                                     for j, v in zip(nb, views)}
    with open(os.path.join(R2R_TINY, "scans.txt"), "w") as f:
        f.write("\n".join(scans) + "\n")
    for scan, nodes in conn.items():
        with open(os.path.join(R2R_TINY, f"{scan}_connectivity.json"), "w") as f:
            json.dump(nodes, f)
    with open(os.path.join(R2R_TINY, "scanvp_cands.json"), "w") as f:
        json.dump(cands, f)
    np.savez_compressed(os.path.join(R2R_TINY, "img_fts.npz"), **feats)

    def traj(scan, idxs, n_instr, long_instr=False, guide=None):
        vps = [f"{scan[-1].lower()}{k:02d}" for k in idxs]
        act = [cands[f"{scan}_{a}"][b][0] for a, b in zip(vps[:-1], vps[1:])] + [-1]
        item = {"scan": scan, "path": vps, "path_viewindex": [int(v) for v in rng.integers(0, 36, len(vps))], "action_viewindex": act,
                "abs_pos_angles": [[float(a), float(b)] for a, b in rng.uniform(-3, 3, (len(vps), 2))],
                "rel_act_angles": [[float(a), float(b)] for a, b in rng.uniform(-1.5, 1.5, (len(vps), 2))],
                "instr_ids": [f"{scan}_{idxs[0]}_{j}" for j in range(n_instr)],
                "instr_encodings": [[101] + [int(t) for t in rng.integers(1996, 29611, (14 if long_instr else 5) + j)] + [102] for j in range(n_instr)]}
        if guide is not None:
            item["guide_path"] = [f"{scan[-1].lower()}{k:02d}" for k in guide]
        return item

    trajs = [traj("scanA", [0, 1, 2, 3], 2), traj("scanA", [0, 2, 6, 5, 4, 3], 1), traj("scanB", [0, 1, 4, 3], 2, long_instr=True),
             traj("scanB", [1, 2, 3], 1, guide=[1, 4])]
    with open(os.path.join(R2R_TINY, "traj.jsonl"), "w") as f:
        for t in trajs[:3]:
            f.write(json.dumps(t) + "\n")
    with open(os.path.join(R2R_TINY, "traj2.jsonl"), "w") as f:
        f.write(json.dumps(trajs[3]) + "\n\n")          # (a blank line at the end: skipped by both readers)


def r2r_tiny_kwargs():
    return dict(traj_files=[os.path.join(R2R_TINY, "traj.jsonl"), os.path.join(R2R_TINY, "traj2.jsonl")], img_ft_file=os.path.join(R2R_TINY, "img_fts.npz"),
                scanvp_cands_file=os.path.join(R2R_TINY, "scanvp_cands.json"), connectivity_dir=R2R_TINY, max_txt_len=12, max_act_len=6, **R2R_DIMS)


def flatten_sample(prefix, out, store):
    for k, v in out.items():
        if isinstance(v, list) and len(v) == 0:
            store[f"{prefix}/{k}/emptylist"] = np.zeros(0)
        elif isinstance(v, str):
            store[f"{prefix}/{k}/str"] = np.asarray(v)
        else:
            store[f"{prefix}/{k}"] = np.asarray(v)


def gen_r2r_data():
    """Row N4 (readers): the reference's own MultiStepNavData (r2r_data.py) on the tiny dataset -- trajectory index lists, angle
    tables, shortest distances, `get_input` over R2R_CASES (both observation layouts, history soft labels, progress, truncation
    by max_act_len / max_txt_len, the STOP step) and the `val_sample_num` subsampling under a fixed numpy seed."""
    make_r2r_tiny()
    ref = ref_shim.import_r2r_data(os.path.join(R2R_TINY, "img_fts.npz"))
    kw = r2r_tiny_kwargs()
    store = {}
    for pano in (True, False):
        db = ref.MultiStepNavData(hist_enc_pano=pano, **kw)
        tag = "pano" if pano else "nopano"
        store[f"{tag}/traj_refer"] = np.asarray(db.traj_refer)
        store[f"{tag}/traj_step_refer"] = np.asarray(db.traj_step_refer)
        for c, case in enumerate(R2R_CASES):
            i, j, t, ob, probs, act, prog, cand = case
            out = db.get_input(i, j, t, return_ob=ob, return_hist_img_probs=probs, return_ob_action=act, return_ob_progress=prog, ob_cand_pano_view=cand)
            flatten_sample(f"{tag}/case{c}", out, store)
    store["angle_features"] = np.stack(db.angle_features, 0)
    store["rel_angles"] = np.stack(db.rel_angles, 0)
    for scan, d in db.shortest_distances.items():
        vps = sorted(d)
        store[f"dist/{scan}"] = np.asarray([[d[a][b] for b in vps] for a in vps])
    np.random.seed(5)
    dbv = ref.MultiStepNavData(val_sample_num=4, **kw)
    store["val/traj_refer"], store["val/traj_step_refer"] = np.asarray(dbv.traj_refer), np.asarray(dbv.traj_step_refer)
    np.savez_compressed(os.path.join(OUT, "r2r_data.npz"), **store)
    print("r2r_data.npz:", len(store), "arrays from the reference's MultiStepNavData on tests/golden/r2r_tiny")


TASK_DS_CASES = [("mlm", [0, 2, 5]), ("mrc", [1, 3, 4]), ("itm", [0, 4]), ("sap", [0, 3, 7, 11, 14]), ("sar", [1, 2, 9, 13]), ("sprel", [0, 5, 8, 12])]


def task_datasets(mod, db):
    import types
    tok = types.SimpleNamespace(cls_token_id=101, sep_token_id=102, mask_token_id=103, pad_token_id=0)
    return {"mlm": mod.MlmDataset(db, tok), "mrc": mod.MrcDataset(db, tok, 0.5), "itm": mod.ItmDataset(db, tok),
            "sap": mod.SapDataset(db, tok, 0.3, 0.43), "sar": mod.SarDataset(db, tok, 0.3, 0.43), "sprel": mod.SprelDataset(db, tok, 0.3, 0.43)}


def gen_r2r_tasks():
    """Row N4 (task datasets): the reference's six Dataset classes (r2r_tasks.py) over its own MultiStepNavData on the tiny dataset,
    python / numpy / torch RNGs seeded per item: word masking, region masks, view / angle kills, SPREL anchors and targets."""
    import random
    make_r2r_tiny()
    rd = ref_shim.import_r2r_data(os.path.join(R2R_TINY, "img_fts.npz"))
    rt = ref_shim.import_collate()
    db = rd.MultiStepNavData(**r2r_tiny_kwargs())
    dss = task_datasets(rt, db)
    store = {}
    for task, idxs in TASK_DS_CASES:
        store[f"{task}/len"] = np.asarray(len(dss[task]))
        for i in idxs:
            random.seed(1000 + i); np.random.seed(2000 + i); torch.manual_seed(3000 + i)
            item = dss[task][i]
            for k, v in item.items():
                store[f"{task}/{i}/{k}"] = v.numpy() if torch.is_tensor(v) else np.asarray(v)
    np.savez_compressed(os.path.join(OUT, "r2r_tasks.npz"), **store)
    print("r2r_tasks.npz:", len(store), "arrays from the reference's six task datasets")


LOADER_RATIOS = {"mlm": 5, "sap": 1, "itm": 2}


def loader_sets():
    """three tiny index datasets of different lengths: a batch is the list of sample ids it holds"""
    from torch.utils.data import TensorDataset
    return {"mlm": TensorDataset(torch.arange(0, 23)), "sap": TensorDataset(torch.arange(100, 107)), "itm": TensorDataset(torch.arange(200, 210))}


def loader_opts(**kw):
    import types
    d = dict(train_batch_size=4, val_batch_size=3, local_rank=-1, n_workers=0, pin_mem=False)
    d.update(kw)
    return types.SimpleNamespace(**d)


def gen_loader():
    """Row N4 (task-mixing loader): the reference's own MetaLoader / build_dataloader (data/loader.py) under fixed torch seeds:
    the (task, batch) sequence of 60 steps for accum_steps 1 and 2 (task draws, per-task epoch restarts and reshuffles included)
    and the attributes of the DataLoaders build_dataloader returns."""
    ref = ref_shim.import_loader()
    col = lambda items: torch.stack([it[0] for it in items])
    store = {}
    for accum in (1, 2):
        torch.manual_seed(100 + accum)
        sets = loader_sets()
        loaders = {n: (ref.build_dataloader(n, sets[n], col, True, loader_opts())[0], r, (lambda e: None)) for n, r in LOADER_RATIOS.items()}
        ml = ref.MetaLoader(loaders, accum_steps=accum, distributed=False, device=None)
        names, flat, lens = [], [], []
        for step, (task, batch) in enumerate(ml):
            if step == 60:
                break
            names.append(list(LOADER_RATIOS).index(task))
            flat += batch.tolist()
            lens.append(len(batch))
        store[f"accum{accum}/task"], store[f"accum{accum}/ids"], store[f"accum{accum}/lens"] = np.asarray(names), np.asarray(flat), np.asarray(lens)
    attrs = []
    for task in ("mlm", "itm"):
        for train in (True, False):
            ld, _ = ref.build_dataloader(task, loader_sets()["mlm"], col, train, loader_opts())
            attrs.append([ld.batch_size, int(type(ld.sampler).__name__ == "RandomSampler"), int(ld.drop_last), ld.num_workers, int(ld.pin_memory)])
    store["attrs"] = np.asarray(attrs)
    np.savez_compressed(os.path.join(OUT, "loader.npz"), **store)
    print("loader.npz:", len(store), "arrays from the reference's MetaLoader / build_dataloader")


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or ["tiny", "optim", "finetune", "canon", "vit", "collate", "r2r_data", "r2r_tasks", "loader", "canon_multi", "canon_multi_sar", "canon_ragged", "a2c", "canon_autocast"]
    for w in which:
        {"tiny": gen_tiny, "canon": gen_canon, "optim": gen_optim, "finetune": gen_finetune, "vit": gen_vit, "collate": gen_collate,
         "r2r_data": gen_r2r_data, "r2r_tasks": gen_r2r_tasks, "loader": gen_loader, "canon_multi": gen_canon_multi, "canon_ragged": gen_canon_ragged, "canon_multi_sar": gen_canon_multi_sar, "a2c": gen_a2c,
         "canon_autocast": gen_canon_autocast,
         "canon_b64": lambda: gen_canon_multi(CANON_B64, "canon_b64.npz", hist_probe=True),
         "canon_b64_sar": lambda: gen_canon_multi_sar(CANON_B64, "canon_b64_sar.npz")}[w]()
