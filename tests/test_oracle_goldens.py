"""Pin the CPU oracle against vectors captured from the real reference (oracle/gen_goldens.py)."""
import hashlib

import numpy as np
import pytest
import torch

from oracle.hamt_oracle import (HamtOracle, OracleConfig, adamw_step, clip_grad_norm, decays, lr_at,
                                make_state_dict, navcmt_param_shapes, pretrain_param_shapes)
from vln_hamt_amd.synth import make_batch

from _util import batch_from, load_npz, sub, tiny_cfg

TINY_CASES = ["mlm", "sap", "sap_nohist", "sar", "sprel", "mrc", "itm", "itm_b1"]


def _sd_hash(sd):
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(sd[k].detach().cpu().numpy().tobytes())
    return h.hexdigest()


@pytest.fixture(scope="module")
def tiny():
    store = load_npz("tiny_pretrain.npz")
    cfg = tiny_cfg()
    sd = make_state_dict(pretrain_param_shapes(cfg), seed=int(store["meta/sd_seed"]))
    assert _sd_hash(sd) == str(store["meta/sd_sha256"]), "numpy weight recipe drifted"
    return store, cfg, sd


def test_param_inventory_matches_reference_count():
    shapes = pretrain_param_shapes(OracleConfig())
    assert len(shapes) == 417                                   # 417-entry state_dict
    n = sum(int(np.prod(s)) for k, s in shapes.items() if k != "mlm_head.predictions.decoder.weight")
    assert n == 174_786_089


@pytest.mark.parametrize("tag", TINY_CASES)
def test_tiny_task_matches_reference(tiny, tag):
    store, cfg, sd = tiny
    task = tag.split("_")[0]
    batch, itm = batch_from(store, tag)
    osd = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k != "mlm_head.predictions.decoder.weight"}
    orc = HamtOracle(osd, cfg)
    loss = orc.forward(batch, task, True, itm)
    np.testing.assert_allclose(loss.detach().numpy(), store[f"{tag}/loss"], rtol=1e-5, atol=1e-6)
    logits = orc.forward(batch, task, False, itm)
    lg = logits[0] if isinstance(logits, tuple) else logits
    ref = store[f"{tag}/logits"]
    fin = np.isfinite(ref)
    assert np.array_equal(np.isfinite(lg.detach().numpy()), fin)           # -inf positions exact (A16)
    np.testing.assert_allclose(lg.detach().numpy()[fin], ref[fin], rtol=1e-5, atol=1e-5)
    loss.mean().backward()
    for k, v in sub(store, f"{tag}/gnorm/").items():
        g = osd[k].grad
        assert g is not None, k
        assert abs(g.double().norm().item() - float(v)) <= 1e-5 * max(1.0, float(v)) + 1e-7, k
    for k, v in sub(store, f"{tag}/grad/").items():
        np.testing.assert_allclose(osd[k].grad.numpy(), v, rtol=1e-4, atol=1e-6, err_msg=k)
    # parameters the reference leaves without grad (unused heads etc.) have none here either
    used = set(sub(store, f"{tag}/gnorm/"))
    for k, p in osd.items():
        if k not in used:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k


@pytest.mark.parametrize("tag", ["mlm", "sap", "sap_nohist", "mrc"])
def test_tiny_trunk_embeddings(tiny, tag):
    store, cfg, sd = tiny
    batch, _ = batch_from(store, tag)
    g = batch.get
    with torch.no_grad():
        t, h, o = HamtOracle(sd, cfg).trunk(g("txt_ids"), g("txt_masks"), g("hist_img_fts"), g("hist_ang_fts"),
                                            g("hist_pano_img_fts"), g("hist_pano_ang_fts"), g("hist_masks"),
                                            g("ob_img_fts"), g("ob_ang_fts"), g("ob_nav_types"), g("ob_masks"))
    np.testing.assert_allclose(t.numpy(), store[f"{tag}/txt_embeds"], atol=2e-5)
    np.testing.assert_allclose(h.numpy(), store[f"{tag}/hist_embeds"], atol=2e-5)
    if o is not None:
        np.testing.assert_allclose(o.numpy(), store[f"{tag}/ob_embeds"], atol=2e-5)


def test_index_goldens_are_bit_exact(tiny):
    """Compaction order (A15/A19), SPREL gather (A18) and -inf fill (A16) are integer work: exact."""
    store, cfg, sd = tiny
    b, _ = batch_from(store, "mlm")
    sel = b["txt_labels"] != -1
    assert store["mlm/logits"].shape[0] == int(sel.sum())
    b, _ = batch_from(store, "sap")
    assert np.array_equal(np.isneginf(store["sap/logits"]), (b["ob_nav_types"] == 0).numpy())
    b, itm = batch_from(store, "itm")
    B = b["txt_ids"].shape[0]
    assert itm["neg_idxs"].shape == (B, 2)
    assert all(int(itm["neg_idxs"][i, k]) != i for i in range(B) for k in range(2))
    lens = (b["hist_masks"].sum(1) - 1).tolist()
    for tab in itm["shuffled_pos_ids"]:
        for i in range(B):
            assert sorted(tab[i, :lens[i]].tolist()) == list(range(lens[i]))
            assert tab[i, lens[i]:].tolist() == list(range(lens[i], tab.shape[1]))


def test_canon_full_config_matches_reference():
    store = load_npz("canon_pretrain.npz")
    cfg = OracleConfig()
    sd = make_state_dict(pretrain_param_shapes(cfg), seed=int(store["meta/sd_seed"]))
    assert _sd_hash(sd) == str(store["meta/sd_sha256"])
    orc = HamtOracle(sd, cfg)
    for task in ("mlm", "sap", "sar", "sprel", "mrc", "itm"):
        batch = make_batch(task, 2 if task != "itm" else 4, cfg, seed=int(store[f"{task}/seed"]), txt_len=80, hist_len=5)
        rng = sub(store, f"{task}/rng/")
        itm = None
        if rng:
            itm = {"neg_idxs": torch.from_numpy(rng["neg_idxs"]),
                   "shuffled_pos_ids": [torch.from_numpy(rng[k]) for k in sorted(rng) if k.startswith("shuffled")]}
        with torch.no_grad():
            loss = orc.forward(batch, task, True, itm)
        np.testing.assert_allclose(loss.numpy(), store[f"{task}/loss"], rtol=2e-5, atol=2e-5, err_msg=task)


@pytest.mark.parametrize("task", ["mlm", "sap", "sar", "sprel", "mrc", "itm"])
def test_canon_backward_matches_reference(task):
    """Backward at the BENCHMARKED model size (main_r2r.py:237-246): every per-parameter gradient norm and a
    257-point probe of every gradient, captured from the reference's own autograd on the R2R-canon model (B=2, L=80, T=5)."""
    from _util import canon_batch, grad_probe
    store = load_npz("canon_pretrain.npz")
    cfg = OracleConfig()
    sd = make_state_dict(pretrain_param_shapes(cfg), seed=int(store["meta/sd_seed"]))
    osd = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k != "mlm_head.predictions.decoder.weight"}
    batch, itm = canon_batch(store, task, cfg)
    HamtOracle(osd, cfg).forward(batch, task, True, itm).mean().backward()
    names = [str(n) for n in store[f"{task}/grad_names"]]
    norms, probes = store[f"{task}/grad_norms"], store[f"{task}/grad_probes"]
    assert {k for k, v in osd.items() if v.grad is not None and float(v.grad.abs().max()) > 0} <= set(names)
    gmax = float(norms.max())
    for i, k in enumerate(names):
        g = osd[k].grad
        assert g is not None, k
        assert abs(float(g.double().norm()) - norms[i]) <= 1e-4 * max(norms[i], 1e-3 * gmax), (k, float(g.double().norm()), norms[i])
        np.testing.assert_allclose(grad_probe(g), probes[i], rtol=0, atol=2e-5 * max(1.0, float(np.abs(probes[i]).max())), err_msg=k)


def test_oracle_matches_reference_at_the_benchmarked_batch():
    """canon_b64.npz (round 6): the REFERENCE's losses at per-GPU batch 64 -- the batch bench.py times and tests/test_gpu_model.py compares the
    HIP path with the oracle at -- reproduced by the oracle (forward of all six tasks), and its SAP backward against the reference's
    per-parameter gradient norms / 65-point probes: the oracle is pinned at B = 64 too, not only at the reference's own B = 2 / 16."""
    from _util import grad_probe
    store = load_npz("canon_b64.npz")
    wseed, bseed, B = (int(v) for v in store["meta/cases"][0])
    cfg = OracleConfig()
    sd = make_state_dict(pretrain_param_shapes(cfg), seed=wseed)
    orc = HamtOracle(sd, cfg)
    for i, task in enumerate(("mlm", "sap", "sar", "sprel", "mrc", "itm")):
        pre = f"c0/{task}/"
        batch = make_batch(task, B if task != "itm" else 2 * B, cfg, seed=bseed + i, txt_len=80, hist_len=5)
        rng = sub(store, pre + "rng/")
        itm = None
        if rng:
            itm = {"neg_idxs": torch.from_numpy(rng["neg_idxs"]),
                   "shuffled_pos_ids": [torch.from_numpy(rng[k]) for k in sorted(rng) if k.startswith("shuffled")]}
        with torch.no_grad():
            loss = orc.forward(batch, task, True, itm)
        np.testing.assert_allclose(loss.numpy(), store[pre + "loss"], rtol=2e-5, atol=2e-5, err_msg=task)
    osd = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k != "mlm_head.predictions.decoder.weight"}
    batch = make_batch("sap", B, cfg, seed=bseed + 1, txt_len=80, hist_len=5)
    HamtOracle(osd, cfg).forward(batch, "sap", True, None).mean().backward()
    names = [str(n) for n in store["c0/sap/grad_names"]]
    norms, probes = store["c0/sap/grad_norms"], store["c0/sap/grad_probes"]
    gmax = float(norms.max())
    for i, k in enumerate(names):
        g = osd[k].grad
        assert g is not None, k
        assert abs(float(g.double().norm()) - norms[i]) <= 1e-4 * max(norms[i], 1e-3 * gmax), (k, float(g.double().norm()), norms[i])
        np.testing.assert_allclose(grad_probe(g, probes.shape[1]), probes[i], rtol=0, atol=2e-5 * max(1.0, float(np.abs(probes[i]).max())), err_msg=k)


@pytest.mark.skipif(not __import__("os").path.isdir("/root/reference"), reason="needs the reference tree (build container only)")
def test_committed_tiny_goldens_match_their_generator(tmp_path, monkeypatch):
    """Generator <-> fixture drift guard: re-running oracle/gen_goldens.py's tiny set against the real reference
    must reproduce the committed file key for key and bit for bit."""
    import importlib
    gen = importlib.import_module("oracle.gen_goldens")
    monkeypatch.setattr(gen, "OUT", str(tmp_path))
    gen.gen_tiny()
    new = dict(np.load(tmp_path / "tiny_pretrain.npz", allow_pickle=False))
    old = load_npz("tiny_pretrain.npz")
    assert set(new) == set(old), sorted(set(new) ^ set(old))[:10]
    for k in new:
        assert np.array_equal(new[k], old[k], equal_nan=new[k].dtype.kind == 'f'), k


def test_optimizer_goldens():
    """3 steps of clip(5.0) + HF AdamW + warmup schedule with the name-based decay groups (A24)."""
    store = load_npz("optim_tiny.npz")
    cfg = tiny_cfg()
    sd = make_state_dict(pretrain_param_shapes(cfg), seed=7)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k != "mlm_head.predictions.decoder.weight"}
    assert sorted(k for k in params if decays(k)) == sorted(store["meta/decay_names"].tolist())
    state = {}
    for step in range(1, 4):
        batch = make_batch("sap", 3, cfg, seed=50 + step, txt_len=20, hist_len=4, ragged=True)
        loss = HamtOracle(params, cfg).forward(batch, "sap", True).mean()
        loss.backward()
        assert abs(loss.item() - float(store[f"step{step}/loss"])) < 2e-5
        lr = lr_at(step, 5e-3, 2, 10)
        assert abs(lr - float(store[f"step{step}/lr"])) < 1e-12
        grads = {k: p.grad for k, p in params.items() if p.grad is not None}
        gn = clip_grad_norm(list(grads.values()), 5.0)
        assert abs(gn.item() - float(store[f"step{step}/grad_norm"])) < 1e-4 * float(store[f"step{step}/grad_norm"])
        with torch.no_grad():
            adamw_step({k: p for k, p in params.items()}, grads, state, lr)
        for p in params.values():
            p.grad = None
        for k, v in sub(store, f"step{step}/param/").items():
            if k == "next_action.net.4.bias":
                # d(CE)/d(shared logit bias) is exactly 0 in exact arithmetic (softmax shift invariance): its
                # gradient is rounding noise which Adam's m/sqrt(v) amplifies to O(lr) -- not a parity signal.
                continue
            np.testing.assert_allclose(params[k].detach().numpy(), v, rtol=2e-4, atol=2e-5, err_msg=f"{step}:{k}")


FT_CASES = [("ca", dict(no_lang_ca=False, act_pred_token="ob_txt")), ("nolangca", dict(no_lang_ca=True, act_pred_token="ob")),
            ("obhist", dict(no_lang_ca=False, act_pred_token="ob_hist")), ("obtxthist", dict(no_lang_ca=False, act_pred_token="ob_txt_hist"))]


@pytest.mark.parametrize("tag,extra", FT_CASES)
def test_finetune_modes(tag, extra):
    store = load_npz("tiny_finetune.npz")
    cfg = tiny_cfg(**extra)
    sd = make_state_dict(navcmt_param_shapes(cfg), seed=9)
    orc = HamtOracle(sd, cfg)
    b = {k: torch.from_numpy(v) for k, v in sub(store, f"{tag}/in/").items()}
    with torch.no_grad():
        lang = orc.ft_forward("language", txt_ids=b["txt_ids"], txt_masks=b["txt_masks"])
        hs = [orc.ft_forward("history").expand(4, -1)]
        for t in range(3):
            hs.append(orc.ft_forward("history", hist_img_feats=b["hist_img_fts"][:, t], hist_ang_feats=b["hist_ang_fts"][:, t],
                                     ob_step_ids=torch.LongTensor([t]), hist_pano_img_feats=b["hist_pano_img_fts"][:, t],
                                     hist_pano_ang_feats=b["hist_pano_ang_fts"][:, t]))
        hist = torch.stack(hs, 1)
        np.testing.assert_allclose(hist.numpy(), store[f"{tag}/hist"], atol=2e-5)
        out = orc.ft_forward("visual", txt_embeds=lang, hist_embeds=hist, txt_masks=b["txt_masks"], hist_masks=b["hist_masks"],
                             ob_img_feats=b["ob_img_fts"], ob_ang_feats=b["ob_ang_fts"], ob_nav_types=b["ob_nav_types"],
                             ob_masks=b["ob_masks"])
    for n, t in zip(("act_logits", "txt", "hist_out", "ob_out"), out):
        ref = store[f"{tag}/{n}"]
        fin = np.isfinite(ref)
        assert np.array_equal(np.isfinite(t.numpy()), fin)
        np.testing.assert_allclose(t.numpy()[fin], ref[fin], atol=2e-5, err_msg=n)


def _length2mask(length, size):
    """finetune_src/utils/misc.py:12-17 (True = padding)"""
    return torch.arange(size)[None].repeat(len(length), 1) > (torch.as_tensor(length) - 1)[:, None]


@pytest.mark.parametrize("tag,no_lang_ca", [("agent_ca", False), ("agent_nolangca", True)])
def test_agent_model_forward(tag, no_lang_ca):
    """Row A23: VLNBertCMT.forward as the agent drives it (model_HAMT.py:20-65), restated over the oracle's NavCMT: stack of
    the per-step history embeddings, hist_masks = not length2mask(hist_lens), states = txt[:, 0] * hist[:, 0] (or hist[:, 0])."""
    store = load_npz("tiny_finetune.npz")
    cfg = tiny_cfg(no_lang_ca=no_lang_ca, act_pred_token="ob" if no_lang_ca else "ob_txt")
    sd = make_state_dict(navcmt_param_shapes(cfg), seed=9)
    orc = HamtOracle(sd, cfg)
    b = {k: torch.from_numpy(v) for k, v in sub(store, f"{tag}/in/").items()}
    lens = store[f"{tag}/hist_lens"].tolist()
    assert np.array_equal(_length2mask(lens, 4).numpy(), store[f"{tag}/length2mask"])
    with torch.no_grad():
        lang = orc.ft_forward("language", txt_ids=b["txt_ids"], txt_masks=b["txt_masks"])
        hs = [orc.ft_forward("history").expand(4, -1)]
        for t in range(3):
            hs.append(orc.ft_forward("history", hist_img_feats=b["hist_img_fts"][:, t], hist_ang_feats=b["hist_ang_fts"][:, t],
                                     ob_step_ids=torch.LongTensor([t]), hist_pano_img_feats=b["hist_pano_img_fts"][:, t],
                                     hist_pano_ang_feats=b["hist_pano_ang_fts"][:, t]))
        hist = torch.stack(hs, 1)
        logits, txt, hist_o, _ = orc.ft_forward("visual", txt_embeds=lang, hist_embeds=hist, txt_masks=b["txt_masks"],
                                                hist_masks=_length2mask(lens, 4).logical_not(), ob_img_feats=b["ob_img_fts"],
                                                ob_ang_feats=b["ob_ang_fts"], ob_nav_types=b["ob_nav_types"], ob_masks=b["ob_masks"])
        states = hist_o[:, 0] if no_lang_ca else txt[:, 0] * hist_o[:, 0]
    np.testing.assert_allclose(hist.numpy(), store[f"{tag}/hist"], atol=2e-5)
    ref = store[f"{tag}/act_logits"]
    fin = np.isfinite(ref)
    assert np.array_equal(np.isfinite(logits.numpy()), fin)
    np.testing.assert_allclose(logits.numpy()[fin], ref[fin], atol=2e-5)
    np.testing.assert_allclose(states.numpy(), store[f"{tag}/states"], atol=2e-5)


def test_critic_matches_reference():
    """Critic (model_HAMT.py:258-269): Linear(768, 512) -> ReLU -> Dropout -> Linear(512, 1) -> squeeze, eval mode."""
    store = load_npz("tiny_finetune.npz")
    sd = make_state_dict({"state2value.0.weight": (512, 768), "state2value.0.bias": (512,), "state2value.3.weight": (1, 512),
                          "state2value.3.bias": (1,)}, seed=int(store["critic/sd_seed"]))
    st = torch.from_numpy(store["critic/state"])
    h = torch.relu(st @ sd["state2value.0.weight"].t() + sd["state2value.0.bias"])
    val = (h @ sd["state2value.3.weight"].t() + sd["state2value.3.bias"]).squeeze()
    np.testing.assert_allclose(val.numpy(), store["critic/value"], atol=2e-5)


@pytest.mark.parametrize("normalize", ["total", "batch", "none"])
@pytest.mark.parametrize("feedback", ["sample", "teacher"])
def test_a2c_restatement_matches_the_reference_block(normalize, feedback):
    """oracle.hamt_oracle.a2c_loss_ref against tests/golden/a2c.npz -- the REFERENCE's own A2C statements (agent_cmt.py:476-517, compiled
    from its file by oracle/gen_goldens.py a2c and run on scripted rollout lists): rl_loss, the logged sums, and the gradients w.r.t. the
    policy log-probabilities, the hidden states (through the critic), the entropies and the critic's parameters.  Pins the restatement
    the HIP kernel is tested against (tests/test_gpu_ops.py)."""
    from oracle.hamt_oracle import a2c_loss_ref
    store = load_npz("a2c.npz")
    sd = make_state_dict({"state2value.0.weight": (512, 768), "state2value.0.bias": (512,), "state2value.3.weight": (1, 512),
                          "state2value.3.bias": (1,)}, seed=int(store["meta/critic_seed"]))
    sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}

    def critic(st):      # model_HAMT.py:258-269, eval mode (pinned by test_critic_matches_reference)
        h = torch.relu(st @ sd["state2value.0.weight"].t() + sd["state2value.0.bias"])
        return (h @ sd["state2value.3.weight"].t() + sd["state2value.3.bias"]).squeeze()
    T = store["in/logp"].shape[0]
    logp = torch.from_numpy(store["in/logp"]).requires_grad_(True)
    hidden = torch.from_numpy(store["in/hidden"]).requires_grad_(True)
    ent = torch.from_numpy(store["in/ent"]).requires_grad_(True)
    last_value = critic(torch.from_numpy(store["in/last_h"])).detach()
    loss, logs = a2c_loss_ref([logp[t] for t in range(T)], [critic(hidden[t]) for t in range(T)], [store["in/rewards"][t] for t in range(T)],
                              [store["in/masks"][t] for t in range(T)], last_value, store["in/ended"],
                              [ent[t] for t in range(T)] if feedback == "sample" else None, gamma=0.9, entropy_loss_weight=0.01, normalize_loss=normalize)
    loss.backward()
    pre = f"{normalize}_{feedback}/"
    assert abs(float(loss) - float(store[pre + "rl_loss"])) <= 1e-6 * max(1.0, abs(float(store[pre + "rl_loss"])))
    assert abs(sum(logs["policy_loss"]) - float(store[pre + "policy_sum"])) <= 1e-4 and abs(sum(logs["critic_loss"]) - float(store[pre + "critic_sum"])) <= 1e-4
    np.testing.assert_allclose(logp.grad.numpy(), store[pre + "d_logp"], atol=1e-6, rtol=1e-5)
    np.testing.assert_allclose(hidden.grad.numpy(), store[pre + "d_hidden"], atol=1e-6, rtol=1e-4)
    if feedback == "sample":
        np.testing.assert_allclose(ent.grad.numpy(), store[pre + "d_ent"], atol=1e-7, rtol=1e-5)
    else:
        assert ent.grad is None and not store[pre + "d_ent"].any()
    for k, v in sub(store, pre + "d_critic/").items():
        np.testing.assert_allclose(sd[k].grad.numpy(), v, atol=1e-5, rtol=1e-4)
    for k, v in sub(store, pre + "d_critic_norm/").items():
        assert abs(float(sd[k].grad.double().norm()) - float(v)) <= 1e-5 * float(v)


def test_get_vlnbert_models_checkpoint_rules(tmp_path):
    """get_vlnbert_models (vlnbert_init.py:13-70) is checkpoint-key plumbing, no arithmetic: `module.` prefixes are stripped
    (:25-26), `next_action.*` becomes `bert.next_action.*` (:29-30) so that loading a PRETRAIN checkpoint (keys `bert.<trunk>`,
    `next_action.*`) into the prefix-less finetune NavCMT drops the `bert.` again; config fields come from the agent's args
    (:42-63).  (The HF 4.12 loader the reference goes through is not in this image: parity of that step is unpinned.)"""
    import types
    from vln_hamt_amd.models.vlnbert_init import get_vlnbert_models
    args = types.SimpleNamespace(bert_ckpt_file=None, dataset="r2r", tokenizer="bert", image_feat_size=64, angle_feat_size=4, num_l_layers=1,
                                 num_h_layers=0, num_x_layers=1, hist_enc_pano=True, hist_pano_num_layers=1, fix_lang_embedding=False,
                                 fix_hist_embedding=False, fix_obs_embedding=False, no_lang_ca=False, act_pred_token="ob_txt")
    fresh = get_vlnbert_models(args)
    want = {k: torch.randn_like(v) for k, v in fresh.state_dict().items()}
    ckpt = {}
    for i, (k, v) in enumerate(want.items()):            # a pretraining checkpoint: trunk under `bert.`, the head prefix-less,
        name = k if k.startswith("next_action") else "bert." + k
        ckpt[("module." + name) if i % 2 else name] = v   # some keys still carrying DDP's `module.` (main_r2r.py saves either)
    ckpt["mlm_head.predictions.bias"] = torch.zeros(3)    # heads the finetune model does not have are ignored
    path = tmp_path / "pretrain.pt"
    torch.save(ckpt, path)
    args.bert_ckpt_file = str(path)
    model = get_vlnbert_models(args)
    got = model.state_dict()
    assert set(got) == set(want)
    for k in want:
        assert torch.equal(got[k], want[k]), k
    c = model.config
    assert (c.vocab_size, c.max_position_embeddings, c.layer_norm_eps, c.type_vocab_size) == (30522, 512, 1e-12, 2)
    assert (c.num_l_layers, c.num_x_layers, c.num_h_pano_layers, c.num_r_layers, c.max_action_steps) == (1, 1, 1, 0, 100)
    assert c.update_lang_bert and c.pred_head_dropout_prob == 0.1 and c.act_pred_token == "ob_txt" and not c.no_lang_ca
    args.dataset, args.bert_ckpt_file = "rxr", None       # RxR: XLM-R vocabulary / positions / eps (run_rxr.sh)
    c = get_vlnbert_models(args).config
    assert (c.vocab_size, c.max_position_embeddings, c.layer_norm_eps, c.type_vocab_size) == (250002, 514, 1e-5, 2)
