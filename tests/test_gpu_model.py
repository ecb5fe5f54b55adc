"""-m gpu: the HIP model (reference class surface) against the CPU oracle and the committed goldens.

Tolerances (north_star): activations/logits/losses <= 1e-3 in fp32 mode, <= 1e-2 in bf16 mode (max-abs,
relative to max(1, |ref|_max)); integer/index work (compaction order, -inf positions) bit exact.
"""
import math
import functools
import os

import numpy as np
import pytest
import torch

from _util import batch_from, load_npz, sub, tiny_cfg


def close(a, b, tol, what=""):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = max(1.0, float(b.abs().max()))
    err = float((a - b).abs().max())
    assert err <= tol * scale, f"{what}: max|d|={err:.3e} scale={scale:.3e} tol={tol}"

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = {"fp32": 1e-3, "bf16": 1e-2}
TINY_CASES = ["mlm", "sap", "sap_nohist", "sar", "sprel", "mrc", "itm", "itm_b1"]


def to_dev(b):
    out = {}
    for k, v in b.items():
        if isinstance(v, list):
            out[k] = [t.to(DEV) for t in v]
        else:
            out[k] = v.to(DEV) if torch.is_tensor(v) else v
    return out


def build(ocfg, sd, prec, train=False):
    from vln_hamt_amd.model.pretrain_cmt import MultiStepNavCMTPreTraining
    from vln_hamt_amd.modeling import HamtConfig
    kw = {k: v for k, v in vars(ocfg).items()}
    kw["pretrain_tasks"] = set(ocfg.pretrain_tasks)
    cfg = HamtConfig(hamt_precision=prec, **kw)
    m = MultiStepNavCMTPreTraining(cfg)
    res = m.load_state_dict(sd, strict=True)
    m = m.to(DEV)
    m.train(train)
    return m


def rel_err(a, ref):
    a, ref = a.detach().cpu().double(), torch.as_tensor(ref).double()
    fin = torch.isfinite(ref)
    assert torch.equal(torch.isfinite(a), fin), "non-finite (-inf) positions differ"
    if fin.sum() == 0:
        return 0.0
    return float((a[fin] - ref[fin]).abs().max()) / max(1.0, float(ref[fin].abs().max()))


@pytest.fixture(scope="module")
def tiny():
    from oracle.hamt_oracle import make_state_dict, pretrain_param_shapes
    store = load_npz("tiny_pretrain.npz")
    cfg = tiny_cfg()
    sd = make_state_dict(pretrain_param_shapes(cfg), seed=int(store["meta/sd_seed"]))
    return store, cfg, sd


def head_tol(prec, fam, case, task):
    """Bound for a HEAD output (ITM logits, SAR predictions): north_star's activation tolerance, flat -- 1e-3 fp32, 1e-2 bf16 -- whatever the
    reference's own bf16 path does on the draw (tests/golden/canon_autocast.npz: torch.autocast(bfloat16) against its fp32 forward is off by
    up to 1.6e-2 on these outputs; rounds 4 / 5 let that widen the gate to 1.25e-2).  Measured on every draw: <= 9.13e-3
    (profiles/r05_parity_margins.txt, profiles/r06_parity_margins.txt); a failing draw is printed with its family / case / task by `gate`."""
    return min(HEAD_CAP, TOL[prec]) if prec == "bf16" else TOL[prec]


# bf16 head outputs: north_star's 1e-2, no allowance
HEAD_CAP = 1.0e-2


def _itm_uncancelled_scale(sd, cfg, cpu_batch, itm):
    """|d(mean_b logit[b, 0])/d(theta)| from the pinned oracle.  The ITM loss gradient is sum_c (p_c - y_c) dlogit_c with five
    nearly identical dlogit_c (same text, the negatives are other / shuffled histories): with random weights the sum cancels to
    1/15 .. 1/16000 of one term, so bf16 rounding of the TERMS, not of the sum, sets the error -- measure it against a term."""
    from oracle.hamt_oracle import HamtOracle
    osd = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k != "mlm_head.predictions.decoder.weight"}
    sc = HamtOracle(osd, cfg).forward(cpu_batch, "itm", False, itm)
    sc = sc[0] if isinstance(sc, tuple) else sc
    (sc[:, 0].sum() / sc.shape[0]).backward()
    return sum(float((v.grad.double() ** 2).sum()) for v in osd.values() if v.grad is not None) ** 0.5


def _itm_uncancelled_sum(sd, cfg, cpu_batch, itm):
    """S = mean_b sum_c |p_bc - y_bc| x |d(mean_b logit[b, 0])/d(theta)|: the size the ITM loss gradient sum_c (p_c - y_c) dlogit_c would
    have if its five (nearly identical) terms did not cancel -- the yardstick of the SAR gate in test_canon_multi_seed_margins: an error
    of |g - ref| <= sqrt(2 (1 - 0.99)) S = 0.1414 S is exactly what cosine 0.99 allows a gradient of norm S."""
    from oracle.hamt_oracle import HamtOracle
    with torch.no_grad():
        sc = HamtOracle(sd, cfg).forward(cpu_batch, "itm", False, itm)
        sc = sc[0] if isinstance(sc, tuple) else sc
        p = torch.softmax(sc.double(), -1)
        p[:, 0] -= 1.0
    return float(p.abs().sum(1).mean()) * _itm_uncancelled_scale(sd, cfg, cpu_batch, itm)


def _batch_with_itm(store, tag):
    batch, itm = batch_from(store, tag)
    if itm is not None:
        batch["itm_neg_idxs"] = itm["neg_idxs"]
        batch["itm_shuffled_pos_ids"] = itm["shuffled_pos_ids"]
    return batch


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("tag", TINY_CASES)
def test_tiny_task_vs_reference_goldens(tiny, tag, prec):
    store, cfg, sd = tiny
    task = tag.split("_")[0]
    model = build(cfg, sd, prec)
    batch = to_dev(_batch_with_itm(store, tag))
    with torch.no_grad():
        loss = model(batch, task, True)
        logits = model(batch, task, False)
    lg = logits[0] if isinstance(logits, tuple) else logits
    e1, e2 = rel_err(lg, store[f"{tag}/logits"]), rel_err(loss, store[f"{tag}/loss"])
    print(f"[{tag} {prec}] logits err {e1:.2e}  loss err {e2:.2e}")
    assert e1 <= TOL[prec] and e2 <= TOL[prec], (e1, e2)
    if isinstance(logits, tuple) and task == "mrc":
        assert torch.equal(logits[1].cpu(), torch.from_numpy(store[f"{tag}/targets"]))      # compaction order exact


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("tag", ["mlm", "sap", "sap_nohist", "mrc"])
def test_tiny_trunk_embeddings(tiny, tag, prec):
    store, cfg, sd = tiny
    model = build(cfg, sd, prec)
    b = to_dev(batch_from(store, tag)[0])
    g = b.get
    with torch.no_grad():
        t, h, o = model.bert(g("txt_ids"), g("txt_masks"), g("hist_img_fts"), g("hist_ang_fts"), g("hist_pano_img_fts"),
                             g("hist_pano_ang_fts"), g("hist_masks"), g("ob_img_fts"), g("ob_ang_fts"), g("ob_nav_types"), g("ob_masks"))
    errs = [rel_err(t, store[f"{tag}/txt_embeds"]), rel_err(h, store[f"{tag}/hist_embeds"])]
    if o is not None:
        errs.append(rel_err(o, store[f"{tag}/ob_embeds"]))
    print(f"[{tag} {prec}] trunk embeds err {errs}")
    assert max(errs) <= TOL[prec], errs


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("tag", TINY_CASES)
def test_tiny_gradients_vs_reference(tiny, tag, prec):
    """d(loss.mean())/d(param).
    fp32 mode (the logic gate): every per-parameter gradient norm and every small gradient tensor stored in the
    goldens (captured from the REFERENCE's autograd) within 2e-3 of max(|ref|, 1e-2*gmax); the set of parameters
    left without gradient is identical.
    bf16 mode: same kernels with bf16-rounded GEMM operands -- compared against the oracle's full gradient:
    global cosine >= 0.99, total norm within 5 %, per-parameter norm within 10 % + 5 % of gmax (gradients that
    cancel to ~0 in exact arithmetic, e.g. the shared-logit bias, are rounding noise in any bf16 run); ITM (whose net
    gradient is the nearly cancelled sum of five candidates' terms): |g - ref| <= 1 % of ONE candidate's gradient."""
    from oracle.hamt_oracle import HamtOracle
    store, cfg, sd = tiny
    task = tag.split("_")[0]
    model = build(cfg, sd, prec)
    cpu_batch, itm = batch_from(store, tag)
    batch = to_dev(_batch_with_itm(store, tag))
    model(batch, task, True).mean().backward()
    gn = sub(store, f"{tag}/gnorm/")
    named = dict(model.named_parameters())
    gmax = max(float(v) for v in gn.values())
    for k, p in named.items():
        if k not in gn:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, f"{k}: unexpected gradient"
    for k in gn:
        assert named[k].grad is not None, f"{k}: reference has a gradient, HIP path has none"
    if prec == "fp32":
        worst = 0.0
        # itm_b1: gradients that cancel to ~0 across the 5 replicas amplify fp32 summation-order noise as well
        ftol = 2e-2 if tag == "itm_b1" else 2e-3
        for k, v in gn.items():
            ref, got = float(v), float(named[k].grad.double().norm())
            lim = ftol * max(ref, 5e-2 * gmax)
            worst = max(worst, abs(got - ref) / max(ref, 5e-2 * gmax))
            assert abs(got - ref) <= lim, f"{k}: |g| {got:.5e} vs {ref:.5e}"
        for k, v in sub(store, f"{tag}/grad/").items():
            g = named[k].grad.detach().cpu().double().reshape(-1)
            ref = torch.from_numpy(v).double().reshape(-1)
            assert float((g - ref).norm()) <= ftol * max(float(ref.norm()), 5e-2 * gmax), k
        print(f"[{tag} fp32] worst per-parameter grad-norm error {worst:.2e} (relative to max(|ref|, 5e-2 gmax))")
    else:
        osd = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k != "mlm_head.predictions.decoder.weight"}
        HamtOracle(osd, cfg).forward(cpu_batch, task, True, itm).mean().backward()
        dot = n1 = n2 = e2 = 0.0
        for k in gn:
            g, r = named[k].grad.detach().cpu().double().reshape(-1), osd[k].grad.double().reshape(-1)
            dot += float(g @ r); n1 += float(g @ g); n2 += float(r @ r); e2 += float((g - r) @ (g - r))
            if task != "itm":
                assert abs(float(g.norm()) - float(r.norm())) <= 0.10 * float(r.norm()) + 0.05 * gmax, k
        cos = dot / (n1 ** 0.5 * n2 ** 0.5)
        if task == "itm":
            # (B=1: the four negatives are position shuffles of one trajectory; the net gradient is 1/16000 of a term)
            scale = _itm_uncancelled_scale(sd, cfg, cpu_batch, itm)
            print(f"[{tag} bf16] |g - ref| = {e2 ** 0.5:.3e} = {e2 ** 0.5 / scale:.2e} of one logit's gradient ({scale:.3e}); net |ref| {n2 ** 0.5:.3e}, cosine {cos:.4f}")
            assert e2 ** 0.5 <= 1e-2 * scale, (e2 ** 0.5, scale)      # measured 2.1e-3 (itm), 3.5e-4 (itm_b1)
        else:
            print(f"[{tag} bf16] global grad cosine {cos:.5f}, norm ratio {(n1 / n2) ** 0.5:.4f}")
            assert cos >= 0.99 and abs((n1 / n2) ** 0.5 - 1) <= 0.05, (cos, (n1 / n2) ** 0.5)


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_canon_full_config_vs_reference_goldens(prec):
    """R2R-canon (H=768, 12 heads, 9+4+2 layers, vocab 30522; B=2, L=80, T=5) against the reference's outputs."""
    from oracle.hamt_oracle import OracleConfig, make_state_dict, pretrain_param_shapes
    from vln_hamt_amd.synth import make_batch
    store = load_npz("canon_pretrain.npz")
    cfg = OracleConfig()
    sd = make_state_dict(pretrain_param_shapes(cfg), seed=int(store["meta/sd_seed"]))
    model = build(cfg, sd, prec)
    for task in ("mlm", "sap", "sar", "sprel", "mrc", "itm"):
        batch = make_batch(task, 2 if task != "itm" else 4, cfg, seed=int(store[f"{task}/seed"]), txt_len=80, hist_len=5)
        rng = sub(store, f"{task}/rng/")
        if rng:
            batch["itm_neg_idxs"] = torch.from_numpy(rng["neg_idxs"])
            batch["itm_shuffled_pos_ids"] = [torch.from_numpy(rng[k]) for k in sorted(rng) if k.startswith("shuffled")]
        batch = to_dev(batch)
        with torch.no_grad():
            loss = model(batch, task, True)
            g = batch.get
            if task != "itm":
                t, h, o = model.bert(g("txt_ids"), g("txt_masks"), g("hist_img_fts"), g("hist_ang_fts"), g("hist_pano_img_fts"),
                                     g("hist_pano_ang_fts"), g("hist_masks"), g("ob_img_fts"), g("ob_ang_fts"), g("ob_nav_types"), g("ob_masks"))
        e = rel_err(loss, store[f"{task}/loss"])
        msg = f"[canon {task} {prec}] loss err {e:.2e}"
        assert e <= TOL[prec], msg
        if task != "itm":
            e_t = rel_err(t[:, :4, :32], store[f"{task}/txt_probe"])
            e_h = rel_err(h, store[f"{task}/hist_embeds"])
            msg += f" txt {e_t:.2e} hist {e_h:.2e}"
            assert e_t <= TOL[prec] and e_h <= TOL[prec], msg
            if o is not None:
                e_o = rel_err(o[:, :, :16], store[f"{task}/ob_probe"])
                msg += f" ob {e_o:.2e}"
                assert e_o <= TOL[prec], msg
        print(msg)


@pytest.mark.parametrize("fam,case", [("multi", 0), ("multi", 1), ("multi", 2), ("multi", 3), ("b64", 0)])
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_canon_multi_seed_margins(fam, case, prec):
    """The R2R-canon parity bounds over MORE than one draw (7.97e-3 against the 1e-2 bf16 bound was one seed at B = 2):
    two further weight seeds at B = 2 and the reference's own per-GPU batch 16 on two seeds (tests/golden/canon_multi.npz, from the
    reference's forward and autograd).  Activations / losses <= 1e-3 (fp32) / 1e-2 (bf16); gradients: fp32 per-parameter norms and
    probes <= 2e-3 of scale, bf16 cosine of the 65-point probes of ALL parameters >= 0.99 (SAR: per regression output, see below).  ITM: besides the loss gradient's
    un-cancelled error, the gradients of single candidate logits (positive k = 0, shuffled negative k = 3) -- the terms whose
    near-cancellation makes the loss gradient's own cosine meaningless -- are gated like the other tasks.  Margins are printed.
    fam = "b64" (round 6): ONE draw at the BENCHMARKED per-GPU batch 64 from the reference itself (canon_b64.npz / canon_b64_sar.npz:
    until then the B = 64 model-level comparison ran against the pinned oracle only) -- fp32 mode at 1e-3 included."""
    from oracle.hamt_oracle import OracleConfig, make_state_dict, pretrain_param_shapes
    from vln_hamt_amd.synth import make_batch
    from _util import grad_probe
    store = load_npz(f"canon_{fam}.npz")
    wseed, bseed, B = (int(v) for v in store["meta/cases"][case])
    cfg = OracleConfig()
    sd = make_state_dict(pretrain_param_shapes(cfg), seed=wseed)
    model = build(cfg, sd, prec)
    named = dict(model.named_parameters())
    worst_act = 0.0
    bad = []                 # every margin is printed before anything fails

    def gate(ok, what):
        if not ok:
            bad.append(what)

    last = {}

    def grads_vs(prefix, what, st=None):
        st = store if st is None else st
        names = [str(n) for n in st[prefix + "grad_names"]]
        norms, probes = st[prefix + "grad_norms"], st[prefix + "grad_probes"].astype(np.float64)
        for k, p in named.items():
            if k not in names:
                assert p.grad is None or float(p.grad.abs().max()) == 0.0, f"{what} {k}: unexpected gradient"
        got = np.stack([grad_probe(named[k].grad, probes.shape[1]) for k in names]).astype(np.float64)
        gn = np.array([float(named[k].grad.double().norm()) for k in names])
        gmax = float(norms.max())
        pcos = float((got * probes).sum() / np.sqrt((got ** 2).sum() * (probes ** 2).sum()))
        nerr = float(np.max(np.abs(gn - norms) / np.maximum(norms, 5e-2 * gmax)))
        pscale = np.maximum(np.abs(probes).max(axis=1, keepdims=True), 5e-2 * np.abs(probes).max())
        perr = float(np.max(np.abs(got - probes) / pscale))
        print(f"    [{what} {prec}] probe cosine {pcos:.5f}, worst norm err {nerr:.2e}, worst probe err {perr:.2e}")
        last["got"], last["ref"] = got, probes
        return pcos, nerr, perr

    for i, task in enumerate(("mlm", "sap", "sar", "sprel", "mrc", "itm")):
        pre = f"c{case}/{task}/"
        batch = make_batch(task, B if task != "itm" else 2 * B, cfg, seed=bseed + i, txt_len=80, hist_len=5)
        rng = sub(store, pre + "rng/")
        if rng:
            batch["itm_neg_idxs"] = torch.from_numpy(rng["neg_idxs"])
            batch["itm_shuffled_pos_ids"] = [torch.from_numpy(rng[k]) for k in sorted(rng) if k.startswith("shuffled")]
        batch = to_dev(batch)
        for p in named.values():
            p.grad = None
        loss = model(batch, task, True)
        errs = {"loss": rel_err(loss, store[pre + "loss"])}
        if task != "itm":
            with torch.no_grad():
                g = batch.get
                t, h, o = model.bert(g("txt_ids"), g("txt_masks"), g("hist_img_fts"), g("hist_ang_fts"), g("hist_pano_img_fts"),
                                     g("hist_pano_ang_fts"), g("hist_masks"), g("ob_img_fts"), g("ob_ang_fts"), g("ob_nav_types"), g("ob_masks"))
            errs["txt"] = rel_err(t[:, :4, :32], store[pre + "txt_probe"])
            errs["hist"] = rel_err(h, store[pre + "hist_embeds"]) if pre + "hist_embeds" in store else rel_err(h[:, :, :64], store[pre + "hist_probe"])
            if o is not None:
                errs["ob"] = rel_err(o[:, :, :16], store[pre + "ob_probe"])
        print(f"[canon {fam} c{case} w{wseed} B{B} {task} {prec}] " + " ".join(f"{k} {v:.2e}" for k, v in errs.items()))
        worst_act = max(worst_act, max(errs.values()))
        gate(max(errs.values()) <= TOL[prec], (task, errs))
        loss.mean().backward()
        pcos, nerr, perr = grads_vs(pre, f"{task} loss gradient")
        if prec == "fp32":
            gate(nerr <= 2e-3 and perr <= 2e-3, (task, nerr, perr))
        elif task == "sar":
            # bf16, SAR: g = sum_{b,k} (2 r_bk / 3B) d pred_bk -- a sum over samples and the three regression outputs whose residuals r
            # have both signs.  bf16 operand rounding perturbs every TERM relative to ITS size; how much of that shows in the cosine of
            # the SUM depends on how far the terms cancel on the draw (one B = 16 draw: 0.982-0.984, every other >= 0.997; the same
            # kernels in fp32 mode are within 2e-3 on that draw).  So, as for ITM: (1) the terms -- the gradients of the three single
            # outputs d(mean_b pred[b, k]) from the REFERENCE's autograd (canon_multi_sar.npz) -- are gated at cosine >= 0.99 like every
            # other gradient; (2) the loss gradient's error is gated against the UN-cancelled size of the sum, S = sum_k (2/3) mean_b
            # |r_bk| |d mean_b pred_bk|: |g - ref| <= sqrt(2 (1 - 0.99)) S = 0.1414 S is exactly what cosine 0.99 allows a gradient of
            # norm S; the cancellation factor |ref| / S is printed.  No draw-specific number is left in the gate.
            sar = load_npz(f"canon_{fam}_sar.npz")
            got_l, ref_l = last["got"], last["ref"]
            resid = np.abs(sar[pre + "logits"].astype(np.float64) - sar[pre + "targets"].astype(np.float64)).mean(axis=0)      # (3,)
            S = 0.0
            for k in range(3):
                for p in named.values():
                    p.grad = None
                pred = model(batch, "sar", False)
                e_p = rel_err(pred, sar[pre + "logits"])
                if k == 0:
                    print(f"    [sar predictions {prec}] err {e_p:.2e} (bound {head_tol(prec, 'multi', case, 'sar'):.2e})")
                gate(e_p <= head_tol(prec, "multi", case, "sar"), ("sar predictions", e_p))
                (pred[:, k].sum() / pred.shape[0]).backward()
                pc, ne, _ = grads_vs(pre + f"out{k}/", f"sar output-{k} gradient", sar)
                gate(pc >= 0.99 and ne <= 0.1, ("sar term", k, pc, ne))
                S += (2.0 / 3.0) * float(resid[k]) * float(np.sqrt((last["ref"] ** 2).sum()))
            err = float(np.sqrt(((got_l - ref_l) ** 2).sum()))
            print(f"    [sar loss gradient {prec}] |g - ref| = {err / S:.3e} of the un-cancelled sum (bound 0.1414); the sum cancels to "
                  f"{float(np.sqrt((ref_l ** 2).sum())) / S:.3f} of it; cosine of the sum {pcos:.5f}")
            gate(err <= 0.1414 * S and nerr <= 0.1, (task, err / S, nerr))
        elif task != "itm":
            gate(pcos >= 0.99 and nerr <= 0.1, (task, pcos, nerr))
        if task == "itm":
            for k in (0, 3):
                for p in named.values():
                    p.grad = None
                lg = model(batch, task, False)
                lg = lg[0] if isinstance(lg, tuple) else lg
                e_lg = rel_err(lg, store[pre + "logits"])
                print(f"    [itm logits {prec}] err {e_lg:.2e} (bound {head_tol(prec, 'multi', case, 'itm'):.2e})")
                gate(e_lg <= head_tol(prec, "multi", case, "itm"), ("itm logits", e_lg))
                (lg[:, k].sum() / lg.shape[0]).backward()
                pcos, nerr, perr = grads_vs(pre + f"logit{k}/", f"itm candidate-{k} logit gradient")
                if prec == "fp32":
                    gate(nerr <= 2e-3 and perr <= 2e-3, (k, nerr, perr))
                else:
                    gate(pcos >= 0.99 and nerr <= 0.1, (k, pcos, nerr))
    print(f"[canon {fam} c{case} w{wseed} B{B} {prec}] worst activation / loss error {worst_act:.2e} of the {TOL[prec]:.0e} bound")
    assert not bad, bad


class _CountCalls:
    """count the calls of C-ABI entry points (the ctypes function objects on the loaded library are swapped for counting wrappers)"""

    def __init__(self, *names):
        self.names, self.n = names, {k: 0 for k in names}

    def __enter__(self):
        from vln_hamt_amd import _lib as L
        self.lib, self.orig = L.load(), {}
        for k in self.names:
            f = getattr(self.lib, k)
            self.orig[k] = f

            def wrap(*a, _f=f, _k=k):
                self.n[_k] += 1
                return _f(*a)
            setattr(self.lib, k, wrap)
        return self

    def __exit__(self, *a):
        for k, f in self.orig.items():
            setattr(self.lib, k, f)


def _attach_itm(batch, rng):
    if rng:
        batch["itm_neg_idxs"] = torch.from_numpy(rng["neg_idxs"])
        batch["itm_shuffled_pos_ids"] = [torch.from_numpy(rng[k]) for k in sorted(rng) if k.startswith("shuffled")]


@pytest.mark.parametrize("case", [0, 1])
@pytest.mark.parametrize("mode", ["bf16-packed", "bf16-padded", "fp32"])
def test_canon_ragged_vs_reference_goldens(case, mode):
    """RAGGED R2R-canon batches (L ~ U[20, 80], T ~ U[0, 7]; the reference's per-GPU batch 16, six tasks) against the
    REFERENCE's forward and autograd (tests/golden/canon_ragged.npz, oracle/gen_goldens.py canon_ragged).  `bf16-packed` runs the HIP
    model WITH the batch's text packing plan (txt_pack_idx / txt_cu / txt_unpack_idx from make_batch(ragged=True), the path
    PrefetchLoader(text_pack=True) and bench.py's `ragged` line take): text embedder + nine text layers + the lang side of the four
    x-layers on the real tokens only, ITM's five packed copies with the explicit pairing, scatter-back behind the last layer --
    asserted to have run through hamt_attn_varlen_* -- and is held to the same bounds as the padded computation: losses / embeddings at
    the real positions <= 1e-2 (bf16) / 1e-3 (fp32), gradient probes of ALL parameters cosine >= 0.99 (bf16; ITM through its single
    candidate logits) / norms and probes <= 2e-3 (fp32).  The reference computes values at padded text positions too; nothing reads
    them (they are masked as keys), the packed path does not produce them: compared at the real positions only."""
    from oracle.hamt_oracle import OracleConfig, make_state_dict, pretrain_param_shapes
    from vln_hamt_amd.synth import make_batch
    from _util import grad_probe
    store = load_npz("canon_ragged.npz")
    wseed, bseed, B = (int(v) for v in store["meta/cases"][case])
    L_, T_ = int(store["meta/txt_len"]), int(store["meta/hist_len"])
    prec = mode.split("-")[0]
    packed = mode == "bf16-packed"
    cfg = OracleConfig()
    sd = make_state_dict(pretrain_param_shapes(cfg), seed=wseed)
    model = build(cfg, sd, prec)
    named = dict(model.named_parameters())
    bad, worst_act = [], 0.0

    def gate(ok, what):
        if not ok:
            bad.append(what)

    def grads_vs(prefix, what):
        names = [str(n) for n in store[prefix + "grad_names"]]
        norms, probes = store[prefix + "grad_norms"], store[prefix + "grad_probes"].astype(np.float64)
        for k, p in named.items():
            if k not in names:
                assert p.grad is None or float(p.grad.abs().max()) == 0.0, f"{what} {k}: unexpected gradient"
        got = np.stack([grad_probe(named[k].grad, probes.shape[1]) for k in names]).astype(np.float64)
        gn = np.array([float(named[k].grad.double().norm()) for k in names])
        gmax = float(norms.max())
        pcos = float((got * probes).sum() / np.sqrt((got ** 2).sum() * (probes ** 2).sum()))
        nerr = float(np.max(np.abs(gn - norms) / np.maximum(norms, 5e-2 * gmax)))
        pscale = np.maximum(np.abs(probes).max(axis=1, keepdims=True), 5e-2 * np.abs(probes).max())
        perr = float(np.max(np.abs(got - probes) / pscale))
        print(f"    [{what} {mode}] probe cosine {pcos:.5f}, worst norm err {nerr:.2e}, worst probe err {perr:.2e}")
        return pcos, nerr, perr

    with _CountCalls("hamt_attn_varlen_fwd", "hamt_attn_varlen_bwd", "hamt_attn_varlen_cross_fwd", "hamt_attn_varlen_cross_bwd") as cnt:
        for i, task in enumerate(("mlm", "sap", "sar", "sprel", "mrc", "itm")):
            pre = f"c{case}/{task}/"
            batch = make_batch(task, B if task != "itm" else 2 * B, cfg, seed=bseed + i, txt_len=L_, hist_len=T_, ragged=True, txt_pack=packed)
            assert np.array_equal(batch["txt_masks"].sum(1).numpy(), store[pre + "txt_lens"]), "the batch generator drifted from the goldens"
            assert np.array_equal((batch["hist_masks"].sum(1) - 1).numpy(), store[pre + "hist_lens"])
            assert ("txt_pack_idx" in batch) == packed
            _attach_itm(batch, sub(store, pre + "rng/"))
            batch = to_dev(batch)
            for p in named.values():
                p.grad = None
            before = dict(cnt.n)
            loss = model(batch, task, True)
            errs = {"loss": rel_err(loss, store[pre + "loss"])}
            if task != "itm":
                with torch.no_grad():
                    g = batch.get
                    if packed:      # (MultiStepNavCMTPreTraining.forward hangs the plan on txt_ids; the trunk is called directly here)
                        batch["txt_ids"]._hamt_pack = (batch["txt_pack_idx"], batch["txt_cu"], batch["txt_unpack_idx"])
                    t, h, o = model.bert(g("txt_ids"), g("txt_masks"), g("hist_img_fts"), g("hist_ang_fts"), g("hist_pano_img_fts"),
                                         g("hist_pano_ang_fts"), g("hist_masks"), g("ob_img_fts"), g("ob_ang_fts"), g("ob_nav_types"), g("ob_masks"))
                real = batch["txt_masks"].unsqueeze(-1)
                ref_sel = torch.from_numpy(store[pre + "txt_sel"])
                errs["txt"] = rel_err(t[:, :, ::24] * real, (ref_sel * real.cpu()).numpy())
                errs["txt_norm"] = rel_err(t.norm(dim=-1) * real[..., 0], store[pre + "txt_norm"] * real[..., 0].cpu().numpy())
                errs["hist"] = rel_err(h[:, :, ::4], store[pre + "hist_sel"])
                if o is not None:
                    errs["ob"] = rel_err(o[:, :, :16], store[pre + "ob_probe"])
            print(f"[canon ragged c{case} w{wseed} B{B} {task} {mode}] " + " ".join(f"{k} {v:.2e}" for k, v in errs.items()))
            worst_act = max(worst_act, max(errs.values()))
            gate(max(errs.values()) <= TOL[prec], (task, errs))
            loss.mean().backward()
            if packed:      # the packed kernels ran, forward and backward, self and cross
                assert all(cnt.n[k] > before[k] for k in cnt.n), (task, before, dict(cnt.n))
            pcos, nerr, perr = grads_vs(pre, f"{task} loss gradient")
            if prec == "fp32":
                gate(nerr <= 2e-3 and perr <= 2e-3, (task, nerr, perr))
            elif task != "itm":
                gate(pcos >= 0.99 and nerr <= 0.1, (task, pcos, nerr))
            if task == "itm":
                for k in (0, 3):
                    for p in named.values():
                        p.grad = None
                    lg = model(batch, task, False)
                    lg = lg[0] if isinstance(lg, tuple) else lg
                    e_lg = rel_err(lg, store[pre + "logits"])
                    print(f"    [itm logits {mode}] err {e_lg:.2e} (bound {head_tol(prec, 'ragged', case, 'itm'):.2e})")
                    gate(e_lg <= head_tol(prec, "ragged", case, "itm"), ("itm logits", e_lg))
                    (lg[:, k].sum() / lg.shape[0]).backward()
                    pcos, nerr, perr = grads_vs(pre + f"logit{k}/", f"itm candidate-{k} logit gradient")
                    if prec == "fp32":
                        gate(nerr <= 2e-3 and perr <= 2e-3, (k, nerr, perr))
                    else:
                        gate(pcos >= 0.99 and nerr <= 0.1, (k, pcos, nerr))
    if not packed:
        assert sum(cnt.n.values()) == 0, cnt.n
    print(f"[canon ragged c{case} w{wseed} B{B} {mode}] worst activation / loss error {worst_act:.2e} of the {TOL[prec]:.0e} bound")
    assert not bad, bad


TINY_RAGGED = ["mlm", "sap", "sar", "sprel", "mrc", "itm"]


@pytest.mark.parametrize("tag", TINY_RAGGED)
def test_tiny_packed_text_vs_reference_goldens(tiny, tag):
    """The tiny goldens' ragged cases (inputs, logits, losses, embeddings, gradients from the REFERENCE) through the PACKED text path: the
    plan is derived from the stored txt_masks (synth.text_pack_plan with a 4-row bucket: B = 3, L = 20 has no padding to drop at the
    default 128-row bucket).  bf16 mode, the bounds of test_tiny_task_vs_reference_goldens / test_tiny_gradients_vs_reference."""
    from oracle.hamt_oracle import HamtOracle
    from vln_hamt_amd.synth import text_pack_plan
    store, cfg, sd = tiny
    task = tag
    model = build(cfg, sd, "bf16")
    cpu_batch, itm = batch_from(store, tag)
    batch = _batch_with_itm(store, tag)
    lens = batch["txt_masks"].sum(1).numpy()
    plan = text_pack_plan(lens, batch["txt_masks"].shape[1], bucket=4)
    assert plan is not None, lens
    batch["txt_pack_idx"], batch["txt_cu"], batch["txt_unpack_idx"] = plan
    batch = to_dev(batch)
    with _CountCalls("hamt_attn_varlen_fwd", "hamt_attn_varlen_bwd", "hamt_attn_varlen_cross_fwd", "hamt_attn_varlen_cross_bwd") as cnt:
        with torch.no_grad():
            logits = model(batch, task, False)
        lg = logits[0] if isinstance(logits, tuple) else logits
        loss = model(batch, task, True)
        e1, e2 = rel_err(lg, store[f"{tag}/logits"]), rel_err(loss, store[f"{tag}/loss"])
        errs = [e1, e2]
        if task != "itm":
            with torch.no_grad():
                g = batch.get
                batch["txt_ids"]._hamt_pack = (batch["txt_pack_idx"], batch["txt_cu"], batch["txt_unpack_idx"])
                t, h, o = model.bert(g("txt_ids"), g("txt_masks"), g("hist_img_fts"), g("hist_ang_fts"), g("hist_pano_img_fts"),
                                     g("hist_pano_ang_fts"), g("hist_masks"), g("ob_img_fts"), g("ob_ang_fts"), g("ob_nav_types"), g("ob_masks"))
            real = batch["txt_masks"].unsqueeze(-1)
            errs.append(rel_err(t * real, store[f"{tag}/txt_embeds"] * real.cpu().numpy()))
            errs.append(rel_err(h, store[f"{tag}/hist_embeds"]))
            if o is not None:
                errs.append(rel_err(o, store[f"{tag}/ob_embeds"]))
        print(f"[tiny packed {tag}] logits / loss / embeddings err " + " ".join(f"{e:.2e}" for e in errs))
        assert max(errs) <= TOL["bf16"], errs
        loss.mean().backward()
        assert cnt.n["hamt_attn_varlen_fwd"] > 0 and cnt.n["hamt_attn_varlen_bwd"] > 0 and cnt.n["hamt_attn_varlen_cross_bwd"] > 0, cnt.n
    gn = sub(store, f"{tag}/gnorm/")
    named = dict(model.named_parameters())
    gmax = max(float(v) for v in gn.values())
    osd = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k != "mlm_head.predictions.decoder.weight"}
    HamtOracle(osd, cfg).forward(cpu_batch, task, True, itm).mean().backward()
    dot = n1 = n2 = e2_ = 0.0
    for k in gn:
        assert named[k].grad is not None, k
        g_, r = named[k].grad.detach().cpu().double().reshape(-1), osd[k].grad.double().reshape(-1)
        dot += float(g_ @ r); n1 += float(g_ @ g_); n2 += float(r @ r); e2_ += float((g_ - r) @ (g_ - r))
        if task != "itm":
            assert abs(float(g_.norm()) - float(r.norm())) <= 0.10 * float(r.norm()) + 0.05 * gmax, k
    cos = dot / (n1 ** 0.5 * n2 ** 0.5)
    if task == "itm":
        scale = _itm_uncancelled_scale(sd, cfg, cpu_batch, itm)
        print(f"[tiny packed itm] |g - ref| = {e2_ ** 0.5 / scale:.2e} of one logit's gradient, cosine {cos:.4f}")
        assert e2_ ** 0.5 <= 1e-2 * scale, (e2_ ** 0.5, scale)
    else:
        print(f"[tiny packed {tag}] global grad cosine {cos:.5f}, norm ratio {(n1 / n2) ** 0.5:.4f}")
        assert cos >= 0.99 and abs((n1 / n2) ** 0.5 - 1) <= 0.05, (cos, (n1 / n2) ** 0.5)


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_canon_gradients_vs_reference_goldens(prec):
    """Backward at the BENCHMARKED size (VERDICT r1 / SURVEY 8c item 2): R2R-canon model, B=2, L=80, T=5, all six tasks --
    the 256-square / 128-row GEMM tiles, the grouped weight-gradient launch and the M = 160..360-row shapes of this
    config against the REFERENCE's autograd (per-parameter gradient norms + 257-point probes in canon_pretrain.npz).
    fp32 mode: every norm within 2e-3 of max(|ref|, 5 % of the largest norm), every probe within 2e-3 of its scale.
    bf16 mode: global cosine of all probes >= 0.99 and of the full gradient against the pinned oracle >= 0.99."""
    from oracle.hamt_oracle import HamtOracle, OracleConfig, make_state_dict, pretrain_param_shapes
    from _util import canon_batch, grad_probe
    store = load_npz("canon_pretrain.npz")
    cfg = OracleConfig()
    sd = make_state_dict(pretrain_param_shapes(cfg), seed=int(store["meta/sd_seed"]))
    model = build(cfg, sd, prec)
    named = dict(model.named_parameters())
    for task in ("mlm", "sap", "sar", "sprel", "mrc", "itm"):
        cpu_batch, itm = canon_batch(store, task, cfg)
        batch = dict(cpu_batch)
        if itm is not None:
            batch["itm_neg_idxs"], batch["itm_shuffled_pos_ids"] = itm["neg_idxs"], itm["shuffled_pos_ids"]
        for p in named.values():
            p.grad = None
        model(to_dev(batch), task, True).mean().backward()
        names = [str(n) for n in store[f"{task}/grad_names"]]
        norms, probes = store[f"{task}/grad_norms"], store[f"{task}/grad_probes"]
        gmax = float(norms.max())
        for k, p in named.items():
            if k not in names:
                assert p.grad is None or float(p.grad.abs().max()) == 0.0, f"{task} {k}: unexpected gradient"
        got = np.stack([grad_probe(named[k].grad) for k in names])
        gn = np.array([float(named[k].grad.double().norm()) for k in names])
        pcos = float((got.astype(np.float64) * probes).sum() / np.sqrt((got.astype(np.float64) ** 2).sum() * (probes.astype(np.float64) ** 2).sum()))
        if prec == "fp32":
            worst = float(np.max(np.abs(gn - norms) / np.maximum(norms, 5e-2 * gmax)))
            pscale = np.maximum(np.abs(probes).max(axis=1, keepdims=True), 5e-2 * np.abs(probes).max())
            pworst = float(np.max(np.abs(got - probes) / pscale))
            print(f"[canon grad {task} fp32] worst norm err {worst:.2e}, worst probe err {pworst:.2e}, probe cosine {pcos:.6f}")
            assert worst <= 2e-3 and pworst <= 2e-3, (task, worst, pworst)
        else:
            osd = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k != "mlm_head.predictions.decoder.weight"}
            HamtOracle(osd, cfg).forward(cpu_batch, task, True, itm).mean().backward()
            dot = n1 = n2 = 0.0
            for k in names:
                g, r = named[k].grad.detach().cpu().double().reshape(-1), osd[k].grad.double().reshape(-1)
                dot += float(g @ r); n1 += float(g @ g); n2 += float(r @ r)
            cos = dot / (n1 ** 0.5 * n2 ** 0.5)
            nerr = float(np.max(np.abs(gn - norms) / np.maximum(norms, 5e-2 * gmax)))
            print(f"[canon grad {task} bf16] global cosine {cos:.5f}, probe cosine {pcos:.5f}, norm ratio {(n1 / n2) ** 0.5:.4f}, worst norm err {nerr:.2e}")
            if task == "itm":      # 5 near-identical candidates: the net gradient is 1/16 of one logit's (see _itm_uncancelled_scale)
                # and the norm of what is left after the cancellation is mostly rounding noise (ratio 0.9-1.2 from build to build): not gated
                err = max(0.0, n1 + n2 - 2 * dot) ** 0.5
                S = _itm_uncancelled_sum(sd, cfg, cpu_batch, itm)
                print(f"[canon grad itm bf16] |g - ref| = {err:.3e} = {err / S:.3e} of the un-cancelled sum ({S:.3e}; bound 0.1414); "
                      f"the sum cancels to {n2 ** 0.5 / S:.3f} of it")
                assert err <= 0.1414 * S, (err, S)
            else:
                assert cos >= 0.99 and pcos >= 0.99 and abs((n1 / n2) ** 0.5 - 1) <= 0.03, (task, cos, pcos, (n1 / n2) ** 0.5)


@functools.lru_cache(maxsize=1)
def _b64_model():
    """the R2R-canon weights (174.8 M elements from the numpy recipe) and the model built on them, shared by the two modes below"""
    from oracle.hamt_oracle import OracleConfig, make_state_dict, pretrain_param_shapes
    cfg = OracleConfig()
    sd = make_state_dict(pretrain_param_shapes(cfg), seed=5)
    return cfg, sd, build(cfg, sd, "bf16")


_B64_TASKS = ("mlm", "sap", "sar", "sprel", "mrc", "itm")


def _b64_oracle_jobs(mode):
    """The pinned oracle's forward + backward passes of one mode of test_canon_b64_vs_oracle (8 at B = 64: 150 of the test's 160 seconds when
    they ran one after the other on the host while the GPU idled -- VERDICT r5 weak 3: the suite at 850 of the driver's 1 200 s), started up
    front on worker threads (torch's CPU ops release the GIL); the GPU passes and the comparisons follow as the results arrive.  Every
    worker thread brings its OWN OpenMP team: with torch's default of one thread per core, eight concurrent passes put 8 x (cores) threads
    on the cores and finish no sooner than eight passes in a row (round 6, first attempt) -- the team size is cut to cores / 8 for the
    duration.  -> (pool, {task: (batch, future of the loss pass, future of the outputs pass or None)}, restore())"""
    from concurrent.futures import ThreadPoolExecutor
    from oracle.hamt_oracle import HamtOracle
    from vln_hamt_amd.synth import make_batch, make_itm_rng
    cfg, sd, _ = _b64_model()
    n_workers = 8
    n_before = torch.get_num_threads()
    torch.set_num_threads(max(4, (os.cpu_count() or 8) // n_workers))

    def oracle_job(task, cpu_batch, itm, outputs):
        osd = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k != "mlm_head.predictions.decoder.weight"}
        with torch.enable_grad():
            if not outputs:
                ref = HamtOracle(osd, cfg).forward(cpu_batch, task, True, itm)
                ref.mean().backward()
            else:
                ref = HamtOracle(osd, cfg).forward(cpu_batch, task, False, itm)
                ref = ref[0] if isinstance(ref, tuple) else ref
                ref[:, 0].mean().backward()
        return ref.detach(), {k: v.grad for k, v in osd.items()}

    pool = ThreadPoolExecutor(max_workers=n_workers)
    packed = mode == "packed"
    work = {}
    for i, task in enumerate(_B64_TASKS):
        batch = make_batch(task, 64 if task != "itm" else 32, cfg, seed=640 + i, txt_len=80, hist_len=7 if packed else 5, ragged=packed, txt_pack=packed)
        itm = None
        if task == "itm":
            itm = make_itm_rng(batch, seed=11)
            batch["itm_neg_idxs"], batch["itm_shuffled_pos_ids"] = itm["neg_idxs"], itm["shuffled_pos_ids"]
        cpu_batch = {k: v for k, v in batch.items() if not k.startswith("txt_pack") and k not in ("txt_cu", "txt_unpack_idx")}
        work[task] = (batch, pool.submit(oracle_job, task, cpu_batch, itm, False),
                      pool.submit(oracle_job, task, cpu_batch, itm, True) if task in ("sar", "itm") else None)
    return pool, work, (lambda: torch.set_num_threads(n_before))


@pytest.mark.parametrize("mode", ["padded", "packed"])
def test_canon_b64_vs_oracle(mode):
    """The BENCHMARKED batch itself (the goldens stop at B = 16): R2R-canon model, B = 64, L = 80, T = 5 (padded, the
    headline line) and the ragged batch with its text packing plan (bench.py's `ragged` line), bf16, ALL SIX tasks (ITM: 32 originals
    = 160 pairs, loader.py:130) -- M = 5120-row GEMMs on the 128- / 256-row tiles at their full grids, the 11 520-row panorama encoder,
    the grouped weight gradients with 20 x 256-row panels per problem -- against the pinned oracle (itself pinned to the reference at
    B = 2 / 16) on the same weights and batch: loss <= 1e-2; the full gradient at cosine >= 0.99 and a norm within 3 % for MLM / SAP /
    SPREL / MRC.  SAR and ITM: the loss gradient is a sum of terms that cancel (regression residuals of both signs; five nearly
    identical candidates: see test_canon_multi_seed_margins), so what is compared at cosine >= 0.99 is the gradient of ONE un-cancelled
    output, mean_b prediction[b, 0] -- and the outputs themselves (predictions / logits) at 1e-2 (head outputs: `head_tol`)."""
    from oracle.hamt_oracle import HamtOracle
    from vln_hamt_amd.synth import make_batch, make_itm_rng
    cfg, sd, model = _b64_model()
    named = dict(model.named_parameters())
    packed = mode == "packed"

    def cosine(ref_grads):
        dot = n1 = n2 = 0.0
        for k, r in ref_grads.items():
            if r is None:
                assert named[k].grad is None or float(named[k].grad.abs().max()) == 0.0, f"{k}: unexpected gradient"
                continue
            g, r = named[k].grad.detach().cpu().double().reshape(-1), r.double().reshape(-1)
            dot += float(g @ r); n1 += float(g @ g); n2 += float(r @ r)
        return dot / (n1 ** 0.5 * n2 ** 0.5), (n1 / n2) ** 0.5

    def clear():
        for p in named.values():
            p.grad = None

    tasks = _B64_TASKS
    pool, work, restore_threads = _b64_oracle_jobs(mode)
    try:
        with _CountCalls("hamt_attn_varlen_fwd", "hamt_attn_varlen_bwd") as cnt:
            for task in tasks:
                batch, f_loss, f_out = work[task]
                clear()
                loss = model(to_dev(batch), task, True)
                loss.mean().backward()
                torch.cuda.synchronize()
                ref, ref_grads = f_loss.result()
                lerr = rel_err(loss, ref)
                cos, ratio = cosine(ref_grads)
                del ref_grads
                print(f"[canon B=64 {mode} {task}] loss err {lerr:.2e}, global gradient cosine {cos:.5f}, norm ratio {ratio:.4f}")
                assert lerr <= TOL["bf16"], (task, lerr)
                if task not in ("sar", "itm"):
                    assert cos >= 0.99 and abs(ratio - 1) <= 0.03, (task, cos, ratio)
                    continue
                clear()
                out = model(to_dev(batch), task, False)
                out = out[0] if isinstance(out, tuple) else out
                out[:, 0].float().mean().backward()
                torch.cuda.synchronize()
                ro, ref_grads = f_out.result()
                oerr = rel_err(out, ro)
                cos1, ratio1 = cosine(ref_grads)
                del ref_grads
                print(f"    [{task} outputs] err {oerr:.2e}; gradient of mean_b output[b, 0]: cosine {cos1:.5f}, norm ratio {ratio1:.4f}")
                assert oerr <= HEAD_CAP and cos1 >= 0.99 and abs(ratio1 - 1) <= 0.03, (task, oerr, cos1, ratio1)
    finally:
        pool.shutdown(wait=True, cancel_futures=True)
        restore_threads()
    assert (cnt.n["hamt_attn_varlen_fwd"] > 0) == packed and (cnt.n["hamt_attn_varlen_bwd"] > 0) == packed, cnt.n


def test_long_text_takes_the_padded_kernels():
    """Instructions beyond the packed kernels' 128 keys (RxR pre-training: max_txt_len 250, pretrain_src/config/rxr_pretrain.json) arrive
    WITH a packing plan from PrefetchLoader(text_pack=True); the trunk must take the padded kernels (<= 256 keys) for them instead of
    calling the varlen kernels outside their range, and still match the oracle."""
    from oracle.hamt_oracle import HamtOracle, OracleConfig, make_state_dict, pretrain_param_shapes
    from vln_hamt_amd.synth import make_batch
    cfg = OracleConfig()
    sd = make_state_dict(pretrain_param_shapes(cfg), seed=6)
    model = build(cfg, sd, "bf16")
    batch = make_batch("sap", 4, cfg, seed=77, txt_len=200, hist_len=5, ragged=True, txt_pack=True)
    assert "txt_pack_idx" in batch and int(batch["txt_masks"].sum(1).max()) > 128
    cpu_batch = {k: v for k, v in batch.items() if not k.startswith("txt_pack") and k not in ("txt_cu", "txt_unpack_idx")}
    with _CountCalls("hamt_attn_varlen_fwd", "hamt_attn_varlen_cross_fwd") as cnt:
        loss = model(to_dev(batch), "sap", True)
        loss.mean().backward()
    ref = HamtOracle(sd, cfg).forward(cpu_batch, "sap", True)
    assert cnt.n == {"hamt_attn_varlen_fwd": 0, "hamt_attn_varlen_cross_fwd": 0}, cnt.n
    assert rel_err(loss, ref) <= TOL["bf16"], rel_err(loss, ref)


def test_train_steps_vs_optimizer_goldens(tiny):
    """3 x (forward, mean, backward, lr schedule, clip 5.0, AdamW, zero_grad) -- main_r2r.py:231-281 order --
    with the name-based decay groups; parameters after each step against the reference's."""
    from vln_hamt_amd.optim import AdamW, clip_grad_norm_
    from vln_hamt_amd.optim.misc import NO_DECAY
    from vln_hamt_amd.synth import make_batch
    from oracle.hamt_oracle import lr_at
    store_t, cfg, sd = tiny
    store = load_npz("optim_tiny.npz")
    model = build(cfg, sd, "fp32")
    named = list(model.named_parameters())
    groups = [{'params': [p for n, p in named if not any(nd in n for nd in NO_DECAY)], 'weight_decay': 0.01},
              {'params': [p for n, p in named if any(nd in n for nd in NO_DECAY)], 'weight_decay': 0.0}]
    opt = AdamW(groups, lr=5e-3, betas=(0.9, 0.98))
    assert sorted(n for n, _ in named if not any(nd in n for nd in NO_DECAY)) == sorted(store["meta/decay_names"].tolist())
    opt.zero_grad()
    for step in range(1, 4):
        batch = to_dev(make_batch("sap", 3, cfg, seed=50 + step, txt_len=20, hist_len=4, ragged=True))
        loss = model(batch, "sap", True).mean()
        loss.backward()
        lr = lr_at(step, 5e-3, 2, 10)
        for g in opt.param_groups:
            g['lr'] = lr
        gn = clip_grad_norm_(model.parameters(), 5.0, optimizer=opt)
        opt.step()
        opt.zero_grad()
        assert abs(float(loss) - float(store[f"step{step}/loss"])) < 1e-3, (step, float(loss))
        assert abs(float(gn) - float(store[f"step{step}/grad_norm"])) < 2e-3 * float(store[f"step{step}/grad_norm"]), (step, float(gn))
        cur = dict(model.named_parameters())
        for k, v in sub(store, f"step{step}/param/").items():
            if k == "next_action.net.4.bias":
                continue   # exactly-zero-gradient parameter: Adam amplifies rounding noise (see test_oracle_goldens)
            d = float((cur[k].detach().cpu() - torch.from_numpy(v)).abs().max())
            assert d < 2e-4, (step, k, d)


def test_state_dict_roundtrip_and_from_pretrained(tiny):
    from vln_hamt_amd.model.pretrain_cmt import MultiStepNavCMTPreTraining
    from vln_hamt_amd.modeling import HamtConfig
    store, cfg, sd = tiny
    kw = dict(vars(cfg))
    kw["pretrain_tasks"] = set(cfg.pretrain_tasks)
    m = MultiStepNavCMTPreTraining.from_pretrained(None, config=HamtConfig(**kw), state_dict=sd)
    out = m.state_dict()
    assert set(out) == set(sd)
    for k in sd:
        assert torch.equal(out[k], sd[k]), k
    assert m.mlm_head.predictions.decoder.weight is m.bert.embeddings.word_embeddings.weight


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("tag,extra", [("ca", dict(no_lang_ca=False, act_pred_token="ob_txt")),
                                       ("nolangca", dict(no_lang_ca=True, act_pred_token="ob")),
                                       ("obhist", dict(no_lang_ca=False, act_pred_token="ob_hist")),
                                       ("obtxthist", dict(no_lang_ca=False, act_pred_token="ob_txt_hist"))])
def test_finetune_navcmt_modes_vs_reference_goldens(tag, extra, prec):
    """finetune twin (vilmodel_cmt.py:624-728): language / history / visual modes against the reference's outputs."""
    from oracle.hamt_oracle import make_state_dict, navcmt_param_shapes
    from vln_hamt_amd.models.vilmodel_cmt import NavCMT
    from vln_hamt_amd.modeling import HamtConfig
    store = load_npz("tiny_finetune.npz")
    ocfg = tiny_cfg(**extra)
    sd = make_state_dict(navcmt_param_shapes(ocfg), seed=9)
    kw = dict(vars(ocfg))
    kw.pop("pretrain_tasks")
    model = NavCMT(HamtConfig(hamt_precision=prec, **kw))
    model.load_state_dict(sd, strict=True)
    model = model.to(DEV).eval()
    b = to_dev({k: torch.from_numpy(v) for k, v in sub(store, f"{tag}/in/").items()})
    with torch.no_grad():
        lang = model("language", txt_ids=b["txt_ids"], txt_masks=b["txt_masks"])
        hs = [model("history").expand(4, -1)]
        for t in range(3):
            hs.append(model("history", hist_img_feats=b["hist_img_fts"][:, t].contiguous(), hist_ang_feats=b["hist_ang_fts"][:, t].contiguous(),
                            ob_step_ids=torch.tensor([t], device=DEV), hist_pano_img_feats=b["hist_pano_img_fts"][:, t].contiguous(),
                            hist_pano_ang_feats=b["hist_pano_ang_fts"][:, t].contiguous()))
        hist = torch.stack(hs, 1)
        out = model("visual", txt_embeds=lang, hist_embeds=hist, txt_masks=b["txt_masks"], hist_masks=b["hist_masks"],
                    ob_img_feats=b["ob_img_fts"], ob_ang_feats=b["ob_ang_fts"], ob_nav_types=b["ob_nav_types"], ob_masks=b["ob_masks"])
    errs = {"hist": rel_err(hist, store[f"{tag}/hist"])}
    if isinstance(lang, list):
        for i, t in enumerate(lang):
            errs[f"lang{i}"] = rel_err(t, store[f"{tag}/lang.{i}"])
    else:
        errs["lang"] = rel_err(lang, store[f"{tag}/lang"])
    for n, t in zip(("act_logits", "txt", "hist_out", "ob_out"), out):
        errs[n] = rel_err(t, store[f"{tag}/{n}"])
    print(f"[finetune {tag} {prec}] " + " ".join(f"{k}={v:.1e}" for k, v in errs.items()))
    assert max(errs.values()) <= TOL[prec], errs


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("tag,no_lang_ca", [("agent_ca", False), ("agent_nolangca", True)])
def test_vlnbert_cmt_forward_vs_reference_goldens(tag, no_lang_ca, prec):
    """Row A23: VLNBertCMT.forward (model_HAMT.py:20-65) driven like the agent (agent_cmt.py:270-397) -- language, history cls,
    one history step per time step with `ob_step`, visual over the LIST of history embeddings with ragged `hist_lens` and
    return_states -- against the outputs of the reference's own class (tiny_finetune.npz).  Built around a tiny NavCMT the way
    the golden generator builds the reference's (get_vlnbert_models always makes the 768-wide model)."""
    import types
    from oracle.hamt_oracle import make_state_dict, navcmt_param_shapes
    from vln_hamt_amd.modeling import HamtConfig
    from vln_hamt_amd.models.model_HAMT import VLNBertCMT, length2mask
    from vln_hamt_amd.models.vilmodel_cmt import NavCMT
    store = load_npz("tiny_finetune.npz")
    ocfg = tiny_cfg(no_lang_ca=no_lang_ca, act_pred_token="ob" if no_lang_ca else "ob_txt")
    sd = make_state_dict(navcmt_param_shapes(ocfg), seed=9)
    kw = dict(vars(ocfg))
    kw.pop("pretrain_tasks")
    agent = VLNBertCMT.__new__(VLNBertCMT)
    torch.nn.Module.__init__(agent)
    agent.args = types.SimpleNamespace(no_lang_ca=no_lang_ca, feat_dropout=0.4)
    agent.vln_bert = NavCMT(HamtConfig(hamt_precision=prec, **kw))
    agent.vln_bert.load_state_dict(sd, strict=True)
    agent.drop_env = torch.nn.Dropout(p=0.4)
    agent = agent.to(DEV).eval()
    b = to_dev({k: torch.from_numpy(v) for k, v in sub(store, f"{tag}/in/").items()})
    lens = store[f"{tag}/hist_lens"].tolist()
    assert np.array_equal(length2mask(lens, size=4).numpy(), store[f"{tag}/length2mask"])          # bit exact index work
    with torch.no_grad():
        lang = agent("language", txt_ids=b["txt_ids"], txt_masks=b["txt_masks"])
        hs = [agent("history").expand(4, -1)]
        for t in range(3):
            hs.append(agent("history", hist_img_feats=b["hist_img_fts"][:, t].contiguous(), hist_ang_feats=b["hist_ang_fts"][:, t].contiguous(),
                            hist_pano_img_feats=b["hist_pano_img_fts"][:, t].contiguous(), hist_pano_ang_feats=b["hist_pano_ang_fts"][:, t].contiguous(), ob_step=t))
        logits, states = agent("visual", txt_embeds=lang, txt_masks=b["txt_masks"], hist_embeds=hs, hist_lens=lens, ob_img_feats=b["ob_img_fts"],
                               ob_ang_feats=b["ob_ang_fts"], ob_nav_types=b["ob_nav_types"], ob_masks=b["ob_masks"], return_states=True)
        (logits_only,) = agent("visual", txt_embeds=lang, txt_masks=b["txt_masks"], hist_embeds=hs, hist_lens=lens, ob_img_feats=b["ob_img_fts"],
                               ob_ang_feats=b["ob_ang_fts"], ob_nav_types=b["ob_nav_types"], ob_masks=b["ob_masks"])
    errs = {"hist": rel_err(torch.stack(hs, 1), store[f"{tag}/hist"]), "act_logits": rel_err(logits, store[f"{tag}/act_logits"]),
            "states": rel_err(states, store[f"{tag}/states"]), "logits_only": rel_err(logits_only, store[f"{tag}/act_logits"])}
    print(f"[agent {tag} {prec}] " + " ".join(f"{k}={v:.1e}" for k, v in errs.items()))
    assert max(errs.values()) <= TOL[prec], errs


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_critic_vs_reference_goldens(prec):
    """Critic (model_HAMT.py:258-269) against the reference class's output."""
    import types
    from oracle.hamt_oracle import make_state_dict
    from vln_hamt_amd.models.model_HAMT import Critic
    store = load_npz("tiny_finetune.npz")
    sd = make_state_dict({"state2value.0.weight": (512, 768), "state2value.0.bias": (512,), "state2value.3.weight": (1, 512),
                          "state2value.3.bias": (1,)}, seed=int(store["critic/sd_seed"]))
    critic = Critic(types.SimpleNamespace(dropout=0.5, hamt_precision=prec))
    critic.load_state_dict(sd, strict=True)
    critic = critic.to(DEV).eval()
    with torch.no_grad():
        val = critic(torch.from_numpy(store["critic/state"]).to(DEV))
    e = rel_err(val, store["critic/value"])
    print(f"[critic {prec}] err {e:.1e}")
    assert val.shape == (8,) and e <= TOL[prec], e


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_rxr_shape_rollout_vs_oracle(prec):
    """BASELINE config 5's shape through the real model size (never under pytest before): XLM-R vocabulary 250 002,
    514 positions, LayerNorm eps 1e-5, 160-token instructions, image features 512-d, 20 history steps (21 history tokens),
    `no_lang_ca` (run_rxr.sh:3, 16, 32-36; vlnbert_init.py:33-41) -- language once, 20 x history, visual over the full
    history -- against the pinned oracle on the same weights."""
    from oracle.hamt_oracle import HamtOracle, OracleConfig, make_state_dict, navcmt_param_shapes
    from vln_hamt_amd.modeling import HamtConfig
    from vln_hamt_amd.models.vilmodel_cmt import NavCMT
    ocfg = OracleConfig(vocab_size=250002, max_position_embeddings=514, layer_norm_eps=1e-5, image_feat_size=512, no_lang_ca=True,
                        act_pred_token="ob_txt")
    sd = make_state_dict(navcmt_param_shapes(ocfg), seed=41)
    kw = dict(vars(ocfg))
    kw.pop("pretrain_tasks")
    model = NavCMT(HamtConfig(hamt_precision=prec, **kw))
    model.load_state_dict(sd, strict=True)
    model = model.to(DEV).eval()
    B, L, T, V, D = 2, 160, 20, 36, 512
    g = torch.Generator().manual_seed(3)
    txt_ids = torch.randint(5, 250002, (B, L), generator=g)
    txt_masks = torch.arange(L)[None] < torch.tensor([[L], [L - 37]])
    pano, pang = torch.randn(T, B, V, D, generator=g), torch.randn(T, B, V, 4, generator=g)
    img, ang = torch.randn(T, B, D, generator=g), torch.randn(T, B, 4, generator=g)
    ob_img, ob_ang = torch.randn(B, V + 1, D, generator=g), torch.randn(B, V + 1, 4, generator=g)
    nav = torch.zeros(B, V + 1, dtype=torch.long)
    nav[:, :5] = 1
    nav[:, V] = 2
    ob_masks = torch.ones(B, V + 1, dtype=torch.bool)
    hist_masks = torch.arange(T + 1)[None] < torch.tensor([[T + 1], [T - 6]])

    def run(m, dev):
        d = lambda t: t.to(dev)
        lang = m("language", txt_ids=d(txt_ids), txt_masks=d(txt_masks))
        hs = [m("history").expand(B, -1)]
        for t in range(T):
            hs.append(m("history", hist_img_feats=d(img[t]), hist_ang_feats=d(ang[t]), ob_step_ids=d(torch.tensor([t])),
                        hist_pano_img_feats=d(pano[t]), hist_pano_ang_feats=d(pang[t])))
        hist = torch.stack(hs, 1)
        out = m("visual", txt_embeds=lang, hist_embeds=hist, txt_masks=d(txt_masks), hist_masks=d(hist_masks), ob_img_feats=d(ob_img),
                ob_ang_feats=d(ob_ang), ob_nav_types=d(nav), ob_masks=d(ob_masks))
        return hist, out
    with torch.no_grad():
        hist, out = run(model, DEV)
        orc = HamtOracle(sd, ocfg)
        rhist, rout = run(lambda mode, **k: orc.ft_forward(mode, **k), "cpu")
    errs = {"hist": rel_err(hist, rhist)}
    for n, a, r in zip(("act_logits", "txt", "hist_out", "ob_out"), out, rout):
        errs[n] = rel_err(a, r)
    print(f"[rxr shape {prec}] " + " ".join(f"{k}={v:.1e}" for k, v in errs.items()))
    assert hist.shape == (B, T + 1, 768) and out[0].shape == (B, V + 1)
    assert max(errs.values()) <= TOL[prec], errs


def test_graph_replay_matches_eager_steps(tiny):
    """hipGraph-captured training steps (vln_hamt_amd.graph) == the same steps launched eagerly (dropout off so that
    masks cannot differ; the rest is the same kernels, the optimizer table refreshed on the host per replay).
    eps = 1.0 makes the Adam update ~linear in the gradient: with the default 1e-6 a parameter whose gradient is
    rounding noise around 0 moves by +-lr, and two EAGER runs already differ by 4e-4 after 4 steps (atomic-add order
    in the embedding scatter), which would make the comparison meaningless."""
    from vln_hamt_amd.graph import GraphedTrainStep
    from vln_hamt_amd.optim import AdamW, clip_grad_norm_
    from vln_hamt_amd.optim.misc import NO_DECAY
    from vln_hamt_amd.synth import make_batch, make_itm_rng
    store, cfg, sd = tiny

    def make():
        m = build(cfg, sd, "bf16", train=True)
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
        named = list(m.named_parameters())
        groups = [{'params': [p for n, p in named if not any(nd in n for nd in NO_DECAY)], 'weight_decay': 0.01},
                  {'params': [p for n, p in named if any(nd in n for nd in NO_DECAY)], 'weight_decay': 0.0}]
        return m, AdamW(groups, lr=1e-3, betas=(0.9, 0.98), eps=1.0)

    seq = ["sap", "mlm", "itm", "sap", "mlm", "itm", "mrc", "sap", "mrc"]
    batches = {}
    for t in set(seq):
        b = make_batch(t, 4, cfg, seed=sum(map(ord, t)), txt_len=20, hist_len=4, ragged=True, device=DEV)
        if t == "itm":
            r = make_itm_rng(b, seed=3)
            b["itm_neg_idxs"], b["itm_shuffled_pos_ids"] = r["neg_idxs"], r["shuffled_pos_ids"]
        batches[t] = b
    m1, o1 = make()
    for t in seq:
        m1(batches[t], t, True).mean().backward()
        clip_grad_norm_(m1.parameters(), 5.0, optimizer=o1)
        o1.step()
        o1.zero_grad()
    m2, o2 = make()
    gs = GraphedTrainStep(m2, o2, 5.0)
    losses = []
    for t in seq:
        losses.append(float(gs.step(t, batches[t], t)))
    gs.finish()                      # (the update of the last replayed step is applied by the next replay or by finish())
    torch.cuda.synchronize()
    assert len(gs.graphs) == 4
    worst = 0.0
    for (k, a), (_, b) in zip(m1.named_parameters(), m2.named_parameters()):
        worst = max(worst, float((a - b).abs().max()))
    print(f"[graph vs eager] worst parameter difference after {len(seq)} steps: {worst:.2e}")
    assert worst < 2e-5, worst      # total parameter movement over the 9 steps is ~9e-3


@pytest.mark.parametrize("wire", ["fp32", "bf16"])
def test_overlapped_grad_sync_matches_single_process_steps(tiny, wire):
    """The multi-GPU step (parallel.OverlappedGradSync: forward/backward graph, grouped weight gradients launched from a
    stored plan with the arena all-reduces on a side stream, update graph) with a one-rank RCCL group == the plain
    single-process steps, both through graphs and eagerly.  One rank makes the collective an identity, so any difference
    is plumbing: missing / doubled gradients, wrong accumulate flags, stream ordering.  wire = bf16: the gradients cross
    the (one-rank) exchange rounded to bf16, so the parameters agree to that resolution only."""
    import torch.distributed as dist
    from vln_hamt_amd.graph import GraphedTrainStep
    from vln_hamt_amd.optim import AdamW, clip_grad_norm_
    from vln_hamt_amd.optim.misc import NO_DECAY
    from vln_hamt_amd.parallel import OverlappedGradSync, broadcast_params
    from vln_hamt_amd.synth import make_batch, make_itm_rng
    from vln_hamt_amd import wgrad
    if not wgrad.ENABLED:
        pytest.skip("HAMT_NO_DEFER_WGRAD: no queued weight gradients to overlap with")
    store, cfg, sd = tiny

    def make():
        m = build(cfg, sd, "bf16", train=True)
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
        named = list(m.named_parameters())
        groups = [{'params': [p for n, p in named if not any(nd in n for nd in NO_DECAY)], 'weight_decay': 0.01},
                  {'params': [p for n, p in named if any(nd in n for nd in NO_DECAY)], 'weight_decay': 0.0}]
        return m, AdamW(groups, lr=1e-3, betas=(0.9, 0.98), eps=1.0)

    seq = ["sap", "mlm", "itm", "sap", "mlm", "itm", "mrc", "sap"]
    batches = {}
    for t in set(seq):
        b = make_batch(t, 4, cfg, seed=sum(map(ord, t)), txt_len=20, hist_len=4, ragged=True, device=DEV)
        if t == "itm":
            r = make_itm_rng(b, seed=3)
            b["itm_neg_idxs"], b["itm_shuffled_pos_ids"] = r["neg_idxs"], r["shuffled_pos_ids"]
        batches[t] = b
    m1, o1 = make()
    for t in seq:
        m1(batches[t], t, True).mean().backward()
        clip_grad_norm_(m1.parameters(), 5.0, optimizer=o1)
        o1.step()
        o1.zero_grad()

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        for use_graph in (True, False):
            m2, o2 = make()
            o2.materialize()
            broadcast_params(o2)
            sync = OverlappedGradSync(o2, n_groups=3, wire=wire)
            try:
                n0 = wgrad.stats["problems"]
                if use_graph:
                    gs = GraphedTrainStep(m2, o2, 5.0, grad_sync=sync)
                    for t in seq:
                        gs.step(t, batches[t], t)
                    assert all(ent[3] is not None and len(ent[3].groups) >= 1 for ent in gs.graphs.values())
                else:
                    for t in seq:
                        m2(batches[t], t, True).mean().backward()
                        sync(o2)
                        clip_grad_norm_(m2.parameters(), 5.0, optimizer=o2)
                        o2.step()
                        o2.zero_grad()
                assert wgrad.stats["problems"] > n0          # the queue did feed the plan
            finally:
                sync.close()
            torch.cuda.synchronize()
            worst = 0.0
            for (k, a), (_, b) in zip(m1.named_parameters(), m2.named_parameters()):
                worst = max(worst, float((a - b).abs().max()))
            print(f"[overlapped sync, graph={use_graph}, wire={wire}] worst parameter difference after {len(seq)} steps: {worst:.2e}")
            assert worst < (2e-5 if wire == "fp32" else 2e-4), (use_graph, worst)
            if wire == "bf16":
                assert worst > 0.0
    finally:
        if created:
            dist.destroy_process_group()


def test_two_stream_cross_layers_match_single_stream():
    """R2R-canon size, training mode, B=16: the vision side of the cross-modal layers on a second stream (streams.py)
    gives the same losses and gradients as the single-stream order, run after run (a cross-stream race would show as a
    run-to-run difference).  Dropout off: the call order, hence the per-call mask ids, differs between the two orders."""
    from oracle.hamt_oracle import OracleConfig, make_state_dict, pretrain_param_shapes
    from vln_hamt_amd import blocks, ops, streams
    from vln_hamt_amd.synth import make_batch, make_itm_rng
    if not blocks.ENABLED:
        pytest.skip("HAMT_NO_FUSED_BLOCKS: the fine-grained ablation path is compared against the goldens only")
    cfg = OracleConfig()
    sd = make_state_dict(pretrain_param_shapes(cfg), seed=11)
    model = build(cfg, sd, "bf16", train=True)
    for mod in model.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    batches = {}
    for task in ("sap", "mlm", "itm"):
        b = make_batch(task, 16, cfg, seed=5 + len(task), txt_len=80, hist_len=5, ragged=True, device=DEV)
        if task == "itm":
            r = make_itm_rng(b, seed=9)
            b["itm_neg_idxs"], b["itm_shuffled_pos_ids"] = r["neg_idxs"], r["shuffled_pos_ids"]
        batches[task] = b

    def run(two):
        streams.set_two_stream(two)
        out = {}
        try:
            for task, b in batches.items():
                ops.manual_seed(77, torch.device(DEV))
                model.zero_grad(set_to_none=True)
                loss = model(b, task, True)
                loss.mean().backward()
                torch.cuda.synchronize()
                gn = torch.stack([p.grad.double().norm() for p in model.parameters() if p.grad is not None])
                out[task] = (loss.detach().double().cpu(), gn.cpu())
        finally:
            streams.set_two_stream(True)
        return out

    ref = run(False)
    ref2 = run(False)     # run-to-run noise floor of the single-stream order (atomic adds in the embedding scatter)
    for rep in range(3):
        got = run(True)
        for task in batches:
            scale = float(ref[task][1].max())
            dl = float((got[task][0] - ref[task][0]).abs().max())
            dg = float((got[task][1] - ref[task][1]).abs().max()) / scale
            floor = float((ref2[task][1] - ref[task][1]).abs().max()) / scale
            print(f"[two-stream rep {rep} {task}] max loss diff {dl:.2e}, max grad-norm diff {dg:.2e} of the largest norm "
                  f"(single-stream run-to-run: {floor:.2e})")
            assert dl <= 1e-6 and dg <= max(1e-6, 10 * floor), (rep, task, dl, dg, floor)


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("task", ["mlm", "sap", "sar", "sprel", "mrc", "itm"])
def test_unread_outputs_of_the_last_cross_layer_are_dead_code(task, prec):
    """What a task head does not read of the LAST cross-modal layer is not computed (vilmodel.LXRTXLayer.forward): the whole vision
    side for MLM / SAR, the whole text side for MRC / SPREL, and of a side of which only some rows are read (the masked positions for
    MLM, the [CLS] rows for SAP / SAR / ITM) the feed-forward block of the other rows.  Against the full computation (HAMT_NO_DCE
    behaviour) on the same weights and batch, dropout off: SPREL / MRC skip whole kernels only and must agree bit for bit; where the
    feed-forward block runs on fewer rows another GEMM tile is picked, so: fp32 mode to summation-order noise, bf16 mode to the
    re-rolled bf16 roundings of that one block.  The skipped side's parameters get no gradient either way."""
    from oracle.hamt_oracle import OracleConfig, make_state_dict, pretrain_param_shapes
    from vln_hamt_amd import ops
    from vln_hamt_amd.model import vilmodel
    from vln_hamt_amd.synth import make_batch, make_itm_rng
    cfg = OracleConfig()
    sd = make_state_dict(pretrain_param_shapes(cfg), seed=13)
    model = build(cfg, sd, prec, train=True)
    for mod in model.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    b = make_batch(task, 8, cfg, seed=21, txt_len=80, hist_len=5, ragged=True, device=DEV)
    if task == "itm":
        r = make_itm_rng(b, seed=9)
        b["itm_neg_idxs"], b["itm_shuffled_pos_ids"] = r["neg_idxs"], r["shuffled_pos_ids"]

    def run(dce):
        old = vilmodel.DEAD_SIDE_ELIMINATION
        vilmodel.DEAD_SIDE_ELIMINATION = dce
        try:
            ops.manual_seed(5, torch.device(DEV))
            model.zero_grad(set_to_none=True)
            loss = model(b, task, True).mean()
            loss.backward()
            torch.cuda.synchronize()
            return loss.detach().double().cpu(), {n: (None if p.grad is None else p.grad.detach().double().cpu()) for n, p in model.named_parameters()}
        finally:
            vilmodel.DEAD_SIDE_ELIMINATION = old

    l_full, g_full = run(False)
    l_dce, g_dce = run(True)
    exact = task in ("sprel", "mrc")
    dl = abs(float(l_full - l_dce)) / max(1.0, abs(float(l_full)))
    assert dl <= (0.0 if exact else (1e-5 if prec == "fp32" else 5e-3)), (float(l_full), float(l_dce))
    last = f"bert.encoder.x_layers.{cfg.num_x_layers - 1}."
    dead = {"mlm": ("visn_self_att", "visn_inter", "visn_output"), "sar": ("visn_self_att", "visn_inter", "visn_output"),
            "sprel": ("lang_self_att", "lang_inter", "lang_output"), "mrc": ("lang_self_att", "lang_inter", "lang_output")}.get(task, ())
    n_dead, dot, na, nb, worst = 0, 0.0, 0.0, 0.0, 0.0
    gmax = max(float(g.abs().max()) for g in g_full.values() if g is not None)
    for n, g in g_full.items():
        h = g_dce[n]
        if n.startswith(last) and any(d in n for d in dead):
            n_dead += 1
            assert (g is None or not bool(g.any())) and (h is None or not bool(h.any())), n
            continue
        gz, hz = g is None or not bool(g.any()), h is None or not bool(h.any())
        assert gz == hz, n
        if gz:
            continue
        if exact:
            assert torch.equal(g, h), (n, float((g - h).abs().max()))
        dot += float((g * h).sum()); na += float((g * g).sum()); nb += float((h * h).sum())
        worst = max(worst, float((g - h).abs().max()) / max(float(g.abs().max()), 1e-4 * gmax))     # (tensors that cancel to noise: against the global scale)
    cos = dot / (na * nb) ** 0.5
    print(f"[dce {task} {prec}] loss diff {dl:.2e}, gradient cosine {cos:.7f}, norm ratio {(nb / na) ** 0.5:.6f}, worst per-tensor max diff {worst:.2e} of its max")
    assert n_dead >= (10 if dead else 0), n_dead
    if prec == "fp32":
        assert cos >= 1 - 1e-8 and worst <= 1e-3, (cos, worst)
    else:
        assert cos >= 0.9995 and abs((nb / na) ** 0.5 - 1) <= 5e-3, (cos, (nb / na) ** 0.5)


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("no_lang_ca", [False, True])
def test_finetune_rollout_backward_vs_oracle(prec, no_lang_ca):
    """Finetune twin, training direction (next row N2): a 3-step rollout -- language once, then history / visual per
    step with the growing history -- with an imitation loss on every step's action logits (agent_cmt.py:476-518 order),
    ONE backward over the whole rollout; parameter gradients against the oracle's autograd on the same inputs."""
    from oracle.hamt_oracle import HamtOracle, make_state_dict, navcmt_param_shapes
    from vln_hamt_amd.models.vilmodel_cmt import NavCMT
    from vln_hamt_amd.modeling import HamtConfig
    store = load_npz("tiny_finetune.npz")
    tag = "nolangca" if no_lang_ca else "ca"
    extra = dict(no_lang_ca=True, act_pred_token="ob") if no_lang_ca else dict(no_lang_ca=False, act_pred_token="ob_txt")
    ocfg = tiny_cfg(**extra)
    for k in ("hidden_dropout_prob", "attention_probs_dropout_prob", "pred_head_dropout_prob"):
        setattr(ocfg, k, 0.0)
    sd = make_state_dict(navcmt_param_shapes(ocfg), seed=9)
    kw = dict(vars(ocfg))
    kw.pop("pretrain_tasks")
    model = NavCMT(HamtConfig(hamt_precision=prec, **kw))
    model.load_state_dict(sd, strict=True)
    model = model.to(DEV).train()
    bc = {k: torch.from_numpy(v) for k, v in sub(store, f"{tag}/in/").items()}
    B = bc["txt_ids"].shape[0]
    target = (bc["ob_nav_types"] != 0).int().argmax(1)          # a navigable view (the others have logit -inf)

    def rollout(fwd, b, dev, ce):
        lang = fwd("language", txt_ids=b["txt_ids"], txt_masks=b["txt_masks"])
        hs = [fwd("history").expand(B, -1)]
        loss = 0.0
        for t in range(3):
            hist = torch.stack(hs, 1)
            out = fwd("visual", txt_embeds=lang, hist_embeds=hist, txt_masks=b["txt_masks"], hist_masks=b["hist_masks"][:, :t + 1].contiguous(),
                      ob_img_feats=b["ob_img_fts"], ob_ang_feats=b["ob_ang_fts"], ob_nav_types=b["ob_nav_types"], ob_masks=b["ob_masks"])
            loss = loss + ce(out[0], target.to(dev)).mean()
            hs.append(fwd("history", hist_img_feats=b["hist_img_fts"][:, t].contiguous(), hist_ang_feats=b["hist_ang_fts"][:, t].contiguous(),
                          ob_step_ids=torch.tensor([t], device=dev), hist_pano_img_feats=b["hist_pano_img_fts"][:, t].contiguous(),
                          hist_pano_ang_feats=b["hist_pano_ang_fts"][:, t].contiguous()))
        return loss

    from vln_hamt_amd import ops
    loss = rollout(model, to_dev(bc), DEV, ops.cross_entropy)
    loss.backward()
    torch.cuda.synchronize()
    # oracle (CPU, fp32 autograd)
    osd = {k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in sd.items()}
    orc = HamtOracle(osd, ocfg, training=True)      # (every dropout probability is 0)
    oloss = rollout(lambda mode, **k: orc.ft_forward(mode, **k), bc, "cpu",
                    lambda x, y: torch.nn.functional.cross_entropy(x, y, reduction="none"))
    oloss.backward()
    assert rel_err(loss.detach(), oloss.detach()) <= TOL[prec]
    ref = {k: v.grad for k, v in osd.items() if v.grad is not None}
    got = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
    gmax = max(float(v.norm()) for v in ref.values())
    num = den = dot = 0.0
    worst = 0.0
    for k, r in ref.items():
        if float(r.norm()) == 0.0 and k not in got:
            continue
        g = got[k].detach().cpu().double()
        r = r.double()
        worst = max(worst, abs(float(g.norm()) - float(r.norm())) / max(float(r.norm()), 5e-2 * gmax))
        dot += float((g * r).sum()); num += float((g * g).sum()); den += float((r * r).sum())
    cos = dot / math.sqrt(num * den)
    print(f"[finetune rollout bwd {tag} {prec}] loss {float(loss):.5f} vs {float(oloss):.5f}; global grad cosine {cos:.6f}; "
          f"worst per-parameter norm error {worst:.2e}")
    assert cos >= (0.99999 if prec == "fp32" else 0.995), cos
    assert worst <= (2e-3 if prec == "fp32" else 6e-2), worst


def test_graphed_inference_rollout_matches_eager():
    """graph.GraphedInference (one captured graph per history length) returns what the eager no-grad call returns, for
    fresh inputs on every replay (the finetune rollout's `visual` / `history` steps)."""
    from oracle.hamt_oracle import make_state_dict, navcmt_param_shapes
    from vln_hamt_amd.graph import GraphedInference
    from vln_hamt_amd.models.vilmodel_cmt import NavCMT
    from vln_hamt_amd.modeling import HamtConfig
    store = load_npz("tiny_finetune.npz")
    ocfg = tiny_cfg(no_lang_ca=True, act_pred_token="ob")
    kw = dict(vars(ocfg))
    kw.pop("pretrain_tasks")
    model = NavCMT(HamtConfig(hamt_precision="bf16", **kw))
    model.load_state_dict(make_state_dict(navcmt_param_shapes(ocfg), seed=9), strict=True)
    model = model.to(DEV).eval()
    b = to_dev({k: torch.from_numpy(v) for k, v in sub(store, "nolangca/in/").items()})
    B = b["txt_ids"].shape[0]
    with torch.no_grad():
        lang = model("language", txt_ids=b["txt_ids"], txt_masks=b["txt_masks"])
        vis = lambda hist, hm, oi, oa: model("visual", txt_embeds=lang, hist_embeds=hist, txt_masks=b["txt_masks"], hist_masks=hm,
                                             ob_img_feats=oi, ob_ang_feats=oa, ob_nav_types=b["ob_nav_types"], ob_masks=b["ob_masks"])
        gv = GraphedInference(vis)
        hs = [model("history").expand(B, -1).contiguous()]
        for rep in range(2):                          # second round replays the graphs captured in the first, on new data
            hs = hs[:1]
            for t in range(3):
                hist = torch.stack(hs, 1)
                hm = b["hist_masks"][:, :t + 1].contiguous()
                oi = b["ob_img_fts"] * (1.0 + 0.25 * rep)
                want = vis(hist, hm, oi, b["ob_ang_fts"])
                got = gv(("visual", t + 1), hist, hm, oi, b["ob_ang_fts"])
                for w, g_ in zip(want, got):
                    fin = torch.isfinite(w)
                    assert torch.equal(torch.isfinite(g_), fin)
                    assert float((w[fin] - g_[fin]).abs().max()) == 0.0, (rep, t)
                hs.append(model("history", hist_img_feats=b["hist_img_fts"][:, t].contiguous(), hist_ang_feats=b["hist_ang_fts"][:, t].contiguous(),
                                ob_step_ids=torch.tensor([t], device=DEV), hist_pano_img_feats=b["hist_pano_img_fts"][:, t].contiguous(),
                                hist_pano_ang_feats=b["hist_pano_ang_fts"][:, t].contiguous()))
        assert len(gv.graphs) == 3


def _tiny_navcmt(no_lang_ca=True, train=False, p_drop=None):
    from oracle.hamt_oracle import make_state_dict, navcmt_param_shapes
    from vln_hamt_amd.models.vilmodel_cmt import NavCMT
    from vln_hamt_amd.modeling import HamtConfig
    ocfg = tiny_cfg(no_lang_ca=no_lang_ca, act_pred_token="ob" if no_lang_ca else "ob_txt")
    if p_drop is not None:
        for k in ("hidden_dropout_prob", "attention_probs_dropout_prob", "pred_head_dropout_prob"):
            setattr(ocfg, k, p_drop)
    kw = dict(vars(ocfg))
    kw.pop("pretrain_tasks")
    model = NavCMT(HamtConfig(hamt_precision="bf16", **kw))
    model.load_state_dict(make_state_dict(navcmt_param_shapes(ocfg), seed=9), strict=True)
    return model.to(DEV).train(train)


def test_rollout_caches_match_the_plain_rollout():
    """Row N2: the step-invariant text side of a `no_lang_ca` rollout is projected to keys / values once (in `language` mode,
    riding on the returned tensors) and the history lives in a device-resident buffer (models.model_HAMT.HistoryCache) -- both
    must give exactly the logits of the plain path (re-projecting every step, re-stacking a Python list: the reference's
    agent_cmt.py:305-397 / vilmodel_cmt.py:701-709).  Also: refreshing a kept language result in place for the next episode
    (copy_language_), which is what a hipGraph-captured `visual` step needs."""
    import types
    from vln_hamt_amd.graph import GraphedInference
    from vln_hamt_amd.models.model_HAMT import HistoryCache, VLNBertCMT
    from vln_hamt_amd.models.vilmodel_cmt import copy_language_
    store = load_npz("tiny_finetune.npz")
    model = _tiny_navcmt()
    agent = VLNBertCMT.__new__(VLNBertCMT)
    torch.nn.Module.__init__(agent)
    agent.args, agent.vln_bert, agent.drop_env = types.SimpleNamespace(no_lang_ca=True, feat_dropout=0.0), model, torch.nn.Dropout(0.0)
    agent.eval()
    b = to_dev({k: torch.from_numpy(v) for k, v in sub(store, "nolangca/in/").items()})
    B, H = b["txt_ids"].shape[0], 128
    hist_step = lambda t: agent("history", hist_img_feats=b["hist_img_fts"][:, t].contiguous(), hist_ang_feats=b["hist_ang_fts"][:, t].contiguous(),
                                hist_pano_img_feats=b["hist_pano_img_fts"][:, t].contiguous(), hist_pano_ang_feats=b["hist_pano_ang_fts"][:, t].contiguous(), ob_step=t)
    vkw = dict(txt_masks=b["txt_masks"], ob_img_feats=b["ob_img_fts"], ob_ang_feats=b["ob_ang_fts"], ob_nav_types=b["ob_nav_types"], ob_masks=b["ob_masks"])
    with torch.no_grad():
        lang = agent("language", txt_ids=b["txt_ids"], txt_masks=b["txt_masks"])
        assert all(hasattr(t, "_hamt_xkv") for t in lang[:-1])
        plain = [t.clone() for t in lang]                       # same values, no cached projections
        cache = HistoryCache(B, 8, H, DEV).reset(agent("history"))
        hs = [agent("history").expand(B, -1)]
        for t in range(3):
            lens = [t + 1] * B
            (want,) = agent("visual", txt_embeds=plain, hist_embeds=hs, hist_lens=lens, **vkw)
            (got,) = agent("visual", txt_embeds=lang, hist_embeds=cache, hist_lens=lens, **vkw)
            fin = torch.isfinite(want)
            assert torch.equal(torch.isfinite(got), fin) and float((want[fin] - got[fin]).abs().max()) == 0.0, t
            h = hist_step(t)
            hs.append(h)
            cache.append(h)
        assert len(cache) == 4 and torch.equal(cache.view(), torch.stack(hs, 1))
        # next episode: other instructions into the SAME tensors; a captured visual step must follow
        gv = GraphedInference(lambda hist: agent.vln_bert("visual", txt_embeds=lang, hist_embeds=hist, hist_masks=torch.ones(B, 4, dtype=torch.bool, device=DEV), **vkw))
        first = [t.clone() for t in gv("v4", cache.view().contiguous())]
        ids2 = b["txt_ids"].flip(0).contiguous()
        lang2 = agent("language", txt_ids=ids2, txt_masks=b["txt_masks"].flip(0).contiguous())
        vkw2 = dict(vkw, txt_masks=b["txt_masks"].flip(0).contiguous())
        want2 = agent.vln_bert("visual", txt_embeds=[t.clone() for t in lang2], hist_embeds=cache.view().contiguous(), hist_masks=torch.ones(B, 4, dtype=torch.bool, device=DEV), **vkw2)
        copy_language_(lang, lang2)
        vkw["txt_masks"].copy_(vkw2["txt_masks"])
        got2 = gv("v4", cache.view().contiguous())
        fin = torch.isfinite(want2[0])
        assert float((want2[0][fin] - got2[0][fin]).abs().max()) == 0.0
        assert float((first[0][fin] - got2[0][fin]).abs().max()) > 0.0          # the episodes do differ


def _rollout_loss(model, b, T):
    """imitation-learning rollout of T steps: language once, visual + history per step with the growing history"""
    from vln_hamt_amd import ops
    B = b["txt_ids"].shape[0]
    lang = model("language", txt_ids=b["txt_ids"], txt_masks=b["txt_masks"])
    hs = [model("history").expand(B, -1)]
    loss = 0.0
    for t in range(T):
        out = model("visual", txt_embeds=lang, hist_embeds=torch.stack(hs, 1), txt_masks=b["txt_masks"], hist_masks=b["hist_masks"][:, :t + 1].contiguous(),
                    ob_img_feats=b["ob_img_fts"], ob_ang_feats=b["ob_ang_fts"], ob_nav_types=b["ob_nav_types"], ob_masks=b["ob_masks"])
        loss = loss + ops.cross_entropy(out[0], b["ob_action_viewindex"]).mean()
        hs.append(model("history", hist_img_feats=b["hist_img_fts"][:, t].contiguous(), hist_ang_feats=b["hist_ang_fts"][:, t].contiguous(),
                        ob_step_ids=b["step_ids"][t:t + 1], hist_pano_img_feats=b["hist_pano_img_fts"][:, t].contiguous(),
                        hist_pano_ang_feats=b["hist_pano_ang_fts"][:, t].contiguous()))
    return loss


def test_graphed_rollout_training_step_matches_eager():
    """Row N2, training direction: a whole imitation-learning rollout (language once, visual + history per step with the growing
    history, a cross-entropy per step -- agent_cmt.py:248-529 order) and its ONE backward, clip and AdamW captured as a single
    hipGraph (GraphedTrainStep with a loss_fn) == the same steps launched eagerly, on fresh batches under one key."""
    from vln_hamt_amd import ops
    from vln_hamt_amd.graph import GraphedTrainStep
    from vln_hamt_amd.optim import AdamW, clip_grad_norm_
    from vln_hamt_amd.synth import make_batch
    T = 3

    def rollout_loss(model, b, _task):
        return _rollout_loss(model, b, T)

    cfg = tiny_cfg(no_lang_ca=True, act_pred_token="ob")
    bs = []
    for i in range(4):
        b = make_batch("sap", 4, cfg, seed=70 + i, txt_len=24, hist_len=T, device=DEV)
        b["step_ids"] = torch.arange(T, device=DEV)
        bs.append(b)
    models = []
    for graphed in (False, True):
        m = _tiny_navcmt(train=True, p_drop=0.0)
        o = AdamW([{"params": list(m.parameters()), "weight_decay": 0.01}], lr=1e-3, betas=(0.9, 0.98), eps=1.0)
        losses = []
        if graphed:
            gs = GraphedTrainStep(m, o, 5.0, loss_fn=rollout_loss)
            for b in bs:
                losses.append(float(gs.step("rollout", b, "rollout")))
            gs.finish()
            assert len(gs.graphs) == 1
        else:
            for b in bs:
                loss = rollout_loss(m, b, None)
                loss.backward()
                clip_grad_norm_(m.parameters(), 5.0, optimizer=o)
                o.step()
                o.zero_grad()
                losses.append(float(loss))
        models.append((m, losses))
    torch.cuda.synchronize()
    (m1, l1), (m2, l2) = models
    # (two runs of the SAME launches already differ at this level: the step-position / token-type embedding gradients are atomic
    # scatter-adds, and a bf16 operand one ulp off moves a loss of ~5 by ~1e-4; the losses fall by 0.1-0.3 per step)
    assert max(abs(a - c) for a, c in zip(l1, l2)) < 5e-4, (l1, l2)
    worst = max(float((a - c).abs().max()) for (_, a), (_, c) in zip(m1.named_parameters(), m2.named_parameters()))
    print(f"[graphed rollout step] losses {l2}; worst parameter difference vs eager {worst:.2e}")
    assert worst < 1e-4, worst


class _NanScratch:
    """While active, every float CUDA buffer the package takes from torch.empty is pre-filled with NaN (the package's modules see
    a stand-in for the `torch` name whose `empty` fills)."""

    def __init__(self, on=True):
        self.on = on

    def __enter__(self):
        import vln_hamt_amd.blocks as blocks_m
        import vln_hamt_amd.blocks_preln as preln_m
        import vln_hamt_amd.ops as ops_m
        import vln_hamt_amd.wgrad as wgrad_m
        import vln_hamt_amd.model.vision_transformer as vit_m
        real_empty = torch.empty

        class _T:
            def __getattr__(self, k):
                return getattr(torch, k)

            @staticmethod
            def empty(*a, **k):
                t = real_empty(*a, **k)
                if t.is_floating_point() and t.is_cuda:
                    t.fill_(float("nan"))
                return t
        self.saved = [(m, m.torch) for m in (blocks_m, preln_m, ops_m, wgrad_m, vit_m)]
        if self.on:
            for m, _ in self.saved:
                m.torch = _T()
        return self

    def __exit__(self, *a):
        for m, t in self.saved:
            m.torch = t


@pytest.mark.parametrize("B", [5, 32, 64])
def test_uninitialised_memory_never_reaches_a_result(B):
    """Every scratch / output buffer of the package comes from torch.empty.  With each float buffer pre-filled with NaN (what the
    caching allocator can hand out: a freed block keeps its bytes) a forward + backward of every task at full width must give the
    SAME loss and the same finite gradients as without: nothing may read an element it (or a kernel before it) did not write --
    padding rows / columns, ragged last tiles, clamped rows.  B = 5 makes every row count ragged, B = 32 is the soak's, B = 64 the
    bench's (the 256-square forward / dgrad tiles only run there)."""
    import bench
    import vln_hamt_amd.ops as ops_m
    from vln_hamt_amd.synth import make_batch, make_itm_rng
    dev = torch.device(DEV)
    ops_m.manual_seed(1, dev)
    model, cfg = bench.build_model("bf16", dev)
    for mod in model.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    named = list(model.named_parameters())
    results = []
    for poisoned in (False, True):
        with _NanScratch(poisoned):
            out = {}
            for it, task in enumerate(["mlm", "sap", "sar", "sprel", "mrc", "itm"]):
                b = make_batch(task, B, cfg, seed=300 + it, txt_len=80, hist_len=5, ragged=True, mlm_exact=7 if task == "mlm" else None, device=dev)
                if task == "itm":
                    r = make_itm_rng(b, seed=it)
                    b["itm_neg_idxs"], b["itm_shuffled_pos_ids"] = r["neg_idxs"], r["shuffled_pos_ids"]
                for p in model.parameters():
                    p.grad = None
                loss = model(b, task, True).mean()
                loss.backward()
                torch.cuda.synchronize()
                out[task] = (float(loss), {n: p.grad.detach().clone() for n, p in named if p.grad is not None})
            results.append(out)
    clean, dirty = results
    for task in clean:
        l0, g0 = clean[task]
        l1, g1 = dirty[task]
        assert l1 == l1 and abs(l0 - l1) <= 1e-5 * max(1.0, abs(l0)), (task, l0, l1)
        assert set(g0) == set(g1), task
        for n in g0:
            a, c = g0[n].double(), g1[n].double()
            assert bool(torch.isfinite(c).all()), (task, n, "non-finite gradient with NaN-filled scratch")
            # Legitimate run-to-run differences (tools/determinism_check.py, same with one stream): the order of fp32 atomic adds
            # -- ~1e-7 on embedding tables / shared LayerNorm parameters; in ITM a row gather with repeated indices (the negative
            # histories) sits in the middle of the graph, its backward scatter-adds into an ACTIVATION gradient and the bf16 images
            # downstream re-round: up to ~1e-3 on every weight.  (SPREL's anchor gather and ITM's text replication did the same until
            # they became broadcasts: 5e-3 at B = 5, which failed this test once.)  The key bias of an attention has a zero true
            # gradient (rounding noise only).
            if n.endswith("key.bias"):
                continue
            # (ITM's net gradient is what is left of five candidates' terms cancelling.  The other tasks: 1e-4 held in > 40 runs and
            # failed once in a full-suite run at 1.6e-3 of a 6e-3 gradient -- atomics order on two streams; a poisoned read shows
            # as NaN / inf or as an O(1) difference, so 5e-3 loses nothing of what this test is for)
            tol = {"itm": 5e-2}.get(task, 5e-3)
            assert float((a - c).abs().max()) <= tol * max(float(a.abs().max()), 1e-6), (task, n, float((a - c).abs().max()), float(a.abs().max()))
    del model
    torch.cuda.empty_cache()


@pytest.mark.parametrize("no_lang_ca", [True, False])
def test_uninitialised_memory_never_reaches_a_rollout_result(no_lang_ca):
    """The same for the finetune model: an imitation-learning rollout (language once, history / visual per step) and its backward."""
    from vln_hamt_amd.synth import make_batch
    T = 3
    m = _tiny_navcmt(no_lang_ca=no_lang_ca, train=True, p_drop=0.0)
    named = list(m.named_parameters())
    b = make_batch("sap", 5, tiny_cfg(no_lang_ca=True, act_pred_token="ob"), seed=77, txt_len=23, hist_len=T, device=DEV)
    b["step_ids"] = torch.arange(T, device=DEV)
    res = []
    for poisoned in (False, True):
        with _NanScratch(poisoned):
            for p in m.parameters():
                p.grad = None
            loss = _rollout_loss(m, b, T)
            loss.backward()
            torch.cuda.synchronize()
            res.append((float(loss), {n: p.grad.detach().clone() for n, p in named if p.grad is not None}))
    (l0, g0), (l1, g1) = res
    assert l1 == l1 and abs(l0 - l1) <= 1e-5 * max(1.0, abs(l0)), (l0, l1)
    for n in g0:
        a, c = g0[n].double(), g1[n].double()
        assert bool(torch.isfinite(c).all()), n
        if not n.endswith("key.bias"):
            assert float((a - c).abs().max()) <= 1e-3 * max(float(a.abs().max()), 1e-6), (n, float((a - c).abs().max()), float(a.abs().max()))


@pytest.mark.parametrize("ragged", [False, True])
def test_training_soak_losses_fall(ragged):
    """(ragged: instructions of 20-80 tokens and 0-7 history steps per sample -- the text packing path of round 3: packed text layers,
    hamt_attn_varlen_*, filler sequences of the bucketed row count.)
    End to end at full size: 120 graph-replayed steps of the six-task mix over 12 fixed synthetic batches (B = 32, lr warm-up to
    5e-5, clip 5.0) must stay finite and overfit (tools/soak.py, shortened).  This is the test that catches what the parity tests
    cannot: a NaN that only some inputs / some step produces (round 2: the folded bias column sums of the 256-square weight-gradient
    tile let 0 x NaN from an out-of-range row of the ragged last tile of the 30 522-row MLM decoder into valid sums -- every parity
    test passed, training diverged after ~40 steps)."""
    import collections
    import bench
    from vln_hamt_amd import ops
    from vln_hamt_amd.graph import GraphedTrainStep
    from vln_hamt_amd.optim import AdamW
    from vln_hamt_amd.optim.misc import NO_DECAY
    from vln_hamt_amd.parallel import TaskSchedule
    from vln_hamt_amd.synth import make_batch, make_itm_rng
    dev = torch.device(DEV)
    ops.manual_seed(1, dev)
    model, cfg = bench.build_model("bf16", dev)
    named = list(model.named_parameters())
    opt = AdamW([{"params": [p for n, p in named if not any(nd in n for nd in NO_DECAY)], "weight_decay": 0.01},
                 {"params": [p for n, p in named if any(nd in n for nd in NO_DECAY)], "weight_decay": 0.0}], lr=5e-5, betas=(0.9, 0.98))
    gs = GraphedTrainStep(model, opt, 5.0)
    sched = TaskSchedule(cyclic=True)
    batches, hist = {}, collections.defaultdict(list)
    for s in range(120):
        task = sched.task_at(s)
        key = (task, s % 12)
        if key not in batches:
            b = make_batch(task, 32, cfg, seed=100 + s % 12, txt_len=80, hist_len=7 if ragged else 5, ragged=ragged,
                           mlm_exact=12 if (task == "mlm" and not ragged) else None, device=dev)
            assert ("txt_pack_idx" in b) == ragged
            if task == "itm":
                r = make_itm_rng(b, seed=s)
                b["itm_neg_idxs"], b["itm_shuffled_pos_ids"] = r["neg_idxs"], r["shuffled_pos_ids"]
            batches[key] = b
        for g in opt.param_groups:
            g["lr"] = 5e-5 * min(1.0, (s + 1) / 50.0)
        hist[task].append(gs.step(key, batches[key], task).detach().clone())
    torch.cuda.synchronize()
    assert bool(torch.isfinite(opt._flat_p).all()) and bool(torch.isfinite(opt._flat_m).all()) and bool(torch.isfinite(opt._flat_v).all())
    drop = {}
    for task, v in hist.items():
        v = [float(x) for x in v]
        assert all(x == x and abs(x) < 1e4 for x in v), (task, v)
        n = max(1, len(v) // 5)
        drop[task] = (sum(v[:n]) / n, sum(v[-n:]) / n)
    print("[soak 120 steps] " + ", ".join(f"{t} {a:.3f} -> {b:.3f}" for t, (a, b) in drop.items()))
    assert drop["mlm"][1] < drop["mlm"][0] - 1.0, drop
    assert drop["sap"][1] < drop["sap"][0] - 0.1, drop
    assert drop["mrc"][1] < drop["mrc"][0] - 0.1, drop
    del gs, opt, model
    torch.cuda.empty_cache()


class _GateChecker:
    """Every read of an nn.Parameter through module attribute access while an overlapped optimizer update is in flight must
    happen on a stream that already waits for the chunk holding it (optim.AdamW.attach: module pre-hooks, container
    declarations, explicit streams.gate calls)."""

    def __enter__(self):
        from vln_hamt_amd import streams
        self.bad = []
        self.orig = orig = torch.nn.Module.__getattr__
        bad = self.bad

        def patched(mod, name):
            v = orig(mod, name)
            if isinstance(v, torch.nn.Parameter):
                for o in streams.pending_updates:
                    if not all(g.check_read(v) for g in o._gates()):
                        bad.append(f"{type(mod).__name__}.{name}")
            return v
        torch.nn.Module.__getattr__ = patched
        return self

    def __exit__(self, *a):
        torch.nn.Module.__getattr__ = self.orig


@pytest.mark.parametrize("which", ["pretrain", "rollout_no_lang_ca", "rollout_lang_ca"])
def test_overlapped_update_gates_every_parameter_read(tiny, which):
    """The container declarations / explicit gates of the model mirrors are complete: with the update of step t running on
    its own stream, no forward pass of step t+1 reads a parameter before the stream it runs on waits for that parameter's
    chunk -- and the overlapped run still reproduces the in-stream run."""
    from vln_hamt_amd.optim import AdamW, clip_grad_norm_
    from vln_hamt_amd.synth import make_batch, make_itm_rng
    store, cfg, sd = tiny

    def run(attach):
        if which == "pretrain":
            m = build(cfg, sd, "bf16", train=True)
            seq = ["sap", "mlm", "sar", "itm", "mrc", "sprel", "mlm"]
            bs = []
            for i, t in enumerate(seq):
                b = make_batch(t, 4, cfg, seed=70 + i, txt_len=20, hist_len=4, ragged=True, device=DEV)
                if t == "itm":
                    r = make_itm_rng(b, seed=3)
                    b["itm_neg_idxs"], b["itm_shuffled_pos_ids"] = r["neg_idxs"], r["shuffled_pos_ids"]
                bs.append(b)
            loss_of = lambda b, t: m(b, t, True).mean()
        else:
            T = 3
            m = _tiny_navcmt(no_lang_ca=which == "rollout_no_lang_ca", train=True, p_drop=0.0)
            seq = ["r"] * 4
            bs = []
            for i in range(4):
                b = make_batch("sap", 4, tiny_cfg(no_lang_ca=True, act_pred_token="ob"), seed=70 + i, txt_len=24, hist_len=T, device=DEV)
                b["step_ids"] = torch.arange(T, device=DEV)
                bs.append(b)
            loss_of = lambda b, t: _rollout_loss(m, b, T)
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
        o = AdamW([{"params": list(m.parameters()), "weight_decay": 0.01}], lr=1e-3, betas=(0.9, 0.98), eps=1.0)
        if attach:
            o.attach(m, chunk_elems=1 << 12)
        losses = []
        for b, t in zip(bs, seq):
            loss = loss_of(b, t)
            loss.backward()
            clip_grad_norm_(m.parameters(), 5.0, optimizer=o)
            o.step()
            o.zero_grad()
            losses.append(float(loss))
        o.wait_update()
        torch.cuda.synchronize()
        return m, losses

    m1, l1 = run(False)
    with _GateChecker() as chk:
        m2, l2 = run(True)
    assert not chk.bad, sorted(set(chk.bad))
    assert max(abs(a - c) for a, c in zip(l1, l2)) < 5e-4, (l1, l2)
    worst = max(float((a - c).abs().max()) for (_, a), (_, c) in zip(m1.named_parameters(), m2.named_parameters()))
    assert worst < 1e-4, worst


@pytest.mark.parametrize("task", ["mlm", "sap", "itm"])
def test_text_packing_matches_the_padded_batch(tiny, task):
    """A ragged batch with a text packing plan (`txt_pack_idx` / `txt_cu` / `txt_unpack_idx`: the nine text-only layers run on the real
    tokens back to back, self-attention per sequence by hamt_attn_varlen_*, vilmodel.NavPreTrainedModel._text) against the same
    batch computed the reference's way (every padded position through every layer, vilmodel.py:441-443): same losses, same text
    embeddings at every REAL position, same history / observation embeddings, same parameter gradients (dropout off).  bf16 mode
    (the packed form exists on the bf16 path); the two runs pick different GEMM tiles for their different row counts, hence 1e-2."""
    from vln_hamt_amd.synth import make_batch, make_itm_rng
    _, cfg, sd = tiny
    m = build(cfg, sd, "bf16")
    named = list(m.named_parameters())
    b_pack = make_batch(task, 24, cfg, seed=91, txt_len=64, hist_len=4, ragged=True, device=DEV)
    assert "txt_pack_idx" in b_pack and b_pack["txt_pack_idx"].numel() < b_pack["txt_ids"].numel()
    assert int(b_pack["txt_cu"][-1]) == b_pack["txt_pack_idx"].numel()
    if task == "itm":
        r = make_itm_rng(b_pack, seed=3)
        b_pack["itm_neg_idxs"], b_pack["itm_shuffled_pos_ids"] = r["neg_idxs"], r["shuffled_pos_ids"]
    b_pad = {k: v for k, v in b_pack.items() if k not in ("txt_pack_idx", "txt_cu", "txt_unpack_idx")}
    outs = []
    for b in (b_pack, b_pad):
        for _, p in named:
            p.grad = None
        loss = m(b, task, True)
        loss.mean().backward()
        emb = None
        if task != "itm":
            with torch.no_grad():
                if "txt_pack_idx" in b:
                    b["txt_ids"]._hamt_pack = (b["txt_pack_idx"], b["txt_cu"], b["txt_unpack_idx"])
                else:
                    b["txt_ids"]._hamt_pack = None
                g = b.get
                emb = m.bert(g("txt_ids"), g("txt_masks"), g("hist_img_fts"), g("hist_ang_fts"), g("hist_pano_img_fts"), g("hist_pano_ang_fts"),
                             g("hist_masks"), g("ob_img_fts"), g("ob_ang_fts"), g("ob_nav_types"), g("ob_masks"))
        outs.append((loss.detach().clone(), emb, {n: p.grad.detach().clone() for n, p in named if p.grad is not None}))
    (l1, e1, g1), (l0, e0, g0) = outs
    assert rel_err(l1, l0.cpu().numpy()) <= 1e-2
    if e1 is not None:
        real = b_pack["txt_masks"].unsqueeze(-1)
        assert rel_err(e1[0] * real, (e0[0] * real).cpu().numpy()) <= 1e-2                 # text at the real positions
        assert rel_err(e1[1], e0[1].cpu().numpy()) <= 1e-2                                # history
        if e1[2] is not None:
            assert rel_err(e1[2], e0[2].cpu().numpy()) <= 1e-2
    assert set(g1) == set(g0)
    num = sum(float((g1[n].double() * g0[n].double()).sum()) for n in g0)
    den = (sum(float((g1[n].double() ** 2).sum()) for n in g0) * sum(float((g0[n].double() ** 2).sum()) for n in g0)) ** 0.5
    print(f"[text packing {task}] gradient cosine packed vs padded {num / den:.6f}")
    assert num / den >= (0.995 if task == "itm" else 0.9995), num / den


@pytest.mark.parametrize("lens", [[80, 80, 3, 17], [1, 2, 128, 64, 33], [16] * 9])
def test_attention_varlen_matches_per_sequence_attention(lens):
    """hamt_attn_varlen_fwd / _bwd on sequences packed back to back == hamt_attn_small_* run on each sequence alone (same kernels,
    same per-(head, query) dropout stream keyed by the sequence index): outputs, lse, dq / dk / dv."""
    import ctypes as C
    from vln_hamt_amd import _lib as L, ops
    heads, H = 2, 128
    S = max(lens)
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=DEV)
    M = int(cu[-1])
    g = torch.Generator(device=DEV).manual_seed(5)
    qkv = torch.randn(M, 3 * H, device=DEV, generator=g).to(torch.bfloat16)
    do = torch.randn(M, H, device=DEV, generator=g).to(torch.bfloat16)
    q, k, v = qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:]
    lib, p, rng = L.load(), ops._p, ops.rng_state(torch.device(DEV))
    d = L.AttnDesc(len(lens), heads, S, S, 64, 3 * H, 3 * H, 3 * H, H, L.HAMT_BF16, L.HAMT_BF16, 0.125, 0.0, 11, L.PREC_BF16)
    o = torch.full((M, H), float("nan"), device=DEV, dtype=torch.bfloat16)
    lse = torch.zeros(len(lens) * heads * S, device=DEV)
    L.check(lib.hamt_attn_varlen_fwd(C.byref(d), p(q), p(k), p(v), p(cu), p(o), p(lse), p(rng), ops._stream()), "hamt_attn_varlen_fwd")
    dqkv = torch.full((M, 3 * H), float("nan"), device=DEV, dtype=torch.bfloat16)
    L.check(lib.hamt_attn_varlen_bwd(C.byref(d), p(q), p(k), p(v), p(cu), p(o), p(do), p(lse), p(dqkv[:, :H]), p(dqkv[:, H:2 * H]), p(dqkv[:, 2 * H:]),
                                     p(rng), ops._stream()), "hamt_attn_varlen_bwd")
    torch.cuda.synchronize()
    assert bool(torch.isfinite(o).all()) and bool(torch.isfinite(dqkv).all())
    for b, n in enumerate(lens):
        r0 = int(cu[b])
        qb = qkv[r0:r0 + n].float()
        qq, kk, vv = (qb[:, i * H:(i + 1) * H].view(n, heads, 64).transpose(0, 1).double().requires_grad_() for i in range(3))
        sc = qq @ kk.transpose(-1, -2) * 0.125
        out = (torch.softmax(sc, -1) @ vv)
        out.backward(do[r0:r0 + n].float().view(n, heads, 64).transpose(0, 1).double())
        close(o[r0:r0 + n].float(), out.transpose(0, 1).reshape(n, H), 1e-2, f"varlen output seq {b}")
        want_lse = torch.logsumexp(sc, -1)                                   # [heads, n]
        got_lse = lse.view(len(lens), heads, S)[b, :, :n]
        close(got_lse, want_lse, 1e-3, f"lse seq {b}")
        for i, (t, nm) in enumerate(((qq, "dq"), (kk, "dk"), (vv, "dv"))):
            close(dqkv[r0:r0 + n, i * H:(i + 1) * H].float(), t.grad.transpose(0, 1).reshape(n, H), 2e-2, f"{nm} seq {b}")


@pytest.mark.gpu
@pytest.mark.parametrize("task", ["sap", "mlm"])
def test_packed_step_replays_with_other_lengths(tiny, task):
    """The text packing plan is DATA of a captured step, not part of it: a step captured on one ragged batch and replayed with another
    batch of the same key (same shapes, same bucketed packed row count; other instruction lengths, other row ranges in `txt_cu`) gives
    the loss the eager step gives on that batch -- nothing derived from the first batch's lengths is baked into the graph (a stale plan
    would hand tokens to the wrong sequences: a different loss, not a slightly different one)."""
    from vln_hamt_amd.graph import GraphedTrainStep
    from vln_hamt_amd.optim import AdamW
    from vln_hamt_amd.synth import make_batch, text_pack_plan
    _, cfg, sd = tiny
    L, B = 48, 16
    b1 = make_batch(task, B, cfg, seed=41, txt_len=L, hist_len=4, ragged=True, mlm_exact=3 if task == "mlm" else None, device=DEV)
    assert "txt_pack_idx" in b1
    # the second batch: the same samples in another order (every tensor rolled along the batch), the plan rebuilt for the new order
    b2 = {k: (torch.roll(v, 5, 0) if torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == B else v) for k, v in b1.items()
          if not k.startswith("txt_pack") and k not in ("txt_cu", "txt_unpack_idx")}
    lens2 = b2["txt_masks"].sum(1).cpu().numpy()
    plan = text_pack_plan(lens2, L)
    b2["txt_pack_idx"], b2["txt_cu"], b2["txt_unpack_idx"] = (t.to(DEV) for t in plan)
    if "txt_label_idx" in b2:      # (the masked positions moved with their samples)
        b2["txt_label_idx"] = (b2["txt_labels"] != -1).reshape(-1).nonzero(as_tuple=False).squeeze(1)
    assert GraphedTrainStep.key_for(task, b1) == GraphedTrainStep.key_for(task, b2)
    assert not torch.equal(b1["txt_cu"], b2["txt_cu"])

    def make():
        m = build(cfg, sd, "bf16", train=True)
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
        return m, AdamW([{"params": list(m.parameters()), "weight_decay": 0.0}], lr=0.0, betas=(0.9, 0.98), eps=1.0)

    m1, _ = make()
    want = [float(m1(b, task, True).mean()) for b in (b1, b2, b1)]
    m2, o2 = make()
    gs = GraphedTrainStep(m2, o2, 5.0)
    key = GraphedTrainStep.key_for(task, b1)
    got = [float(gs.step(key, b, task)) for b in (b1, b2, b1)]      # capture on b1, replay with b2, replay with b1 (lr = 0: same weights)
    for w, g in zip(want, got):
        assert abs(w - g) <= 2e-3 * max(1.0, abs(w)), (want, got)


@pytest.mark.gpu
@pytest.mark.parametrize("packed_side,with_pair", [("q", False), ("k", False), ("q", True), ("k", True)])
def test_attention_varlen_cross_matches_per_pair_attention(packed_side, with_pair):
    """hamt_attn_varlen_cross_fwd / _bwd -- ONE side packed back to back, the other at a fixed stride with an additive key mask --
    against fp64 attention of each (query sequence, key sequence) pair: outputs, lse, dq / dk / dv; the filler query sequences (behind
    `n_pairs`, or named -1 by a `pair` map that also permutes who attends to whom) give zero outputs and zero dq, and touch no dk / dv."""
    import ctypes as C
    from vln_hamt_amd import _lib as L, ops
    heads, H, Sf = 2, 128, 11                    # fixed-stride side: 11 rows per sample, the last rows masked for some samples
    n = 5
    if with_pair:     # packed sequences in this order; -1 = a filler; the numbers = the fixed-stride sample each one belongs to
        owner = [2, -1, 0, 4, -1, 1, 3] if packed_side == "q" else [3, -1, 0, 2, 4, -1, 1]
        plens = [5, 30, 37, 1, 7, 80, 16] if packed_side == "q" else [16, 9, 5, 80, 1, 0, 37]
    else:
        owner = [0, 1, 2, 3, 4] + ([-1, -1] if packed_side == "q" else [-1])
        plens = [5, 37, 1, 80, 16] + ([30, 7] if packed_side == "q" else [9])
    Sp = max(plens)
    cu = torch.tensor([0] + list(np.cumsum(plens)), dtype=torch.int32, device=DEV)
    Mp, Mf = int(cu[-1]), n * Sf
    g = torch.Generator(device=DEV).manual_seed(9)
    valid_f = [11, 4, 11, 7, 1]
    mask = torch.zeros(n, Sf, device=DEV)
    for b, m in enumerate(valid_f):
        mask[b, m:] = -10000.0
    lib, p, rng = L.load(), ops._p, ops.rng_state(torch.device(DEV))
    seq_of = {o: i for i, o in enumerate(owner) if o >= 0}         # fixed-stride sample -> its packed sequence
    if packed_side == "q":
        q = torch.randn(Mp, H, device=DEV, generator=g).to(torch.bfloat16)
        kv = torch.randn(Mf, 2 * H, device=DEV, generator=g).to(torch.bfloat16)
        nseq, Sq, Sk, cu_q, cu_k, am = len(plens), Sp, Sf, cu, None, mask
        pair = torch.tensor(owner, dtype=torch.int32, device=DEV) if with_pair else None
    else:
        q = torch.randn(Mf, H, device=DEV, generator=g).to(torch.bfloat16)
        kv = torch.randn(Mp, 2 * H, device=DEV, generator=g).to(torch.bfloat16)
        nseq, Sq, Sk, cu_q, cu_k, am = n, Sf, Sp, None, cu, None
        pair = torch.tensor([seq_of[b] for b in range(n)], dtype=torch.int32, device=DEV) if with_pair else None
    Mq = q.shape[0]
    do = torch.randn(Mq, H, device=DEV, generator=g).to(torch.bfloat16)
    d = L.AttnDesc(nseq, heads, Sq, Sk, 64, H, 2 * H, 2 * H, H, L.HAMT_BF16, L.HAMT_BF16, 0.125, 0.0, 13, L.PREC_BF16)
    o = torch.full((Mq, H), float("nan"), device=DEV, dtype=torch.bfloat16)
    lse = torch.full((nseq * heads * Sq,), float("nan"), device=DEV)
    L.check(lib.hamt_attn_varlen_cross_fwd(C.byref(d), p(q), p(kv[:, :H]), p(kv[:, H:]), p(cu_q), p(cu_k), n, p(pair), p(am), p(o), p(lse), p(rng),
                                           ops._stream()), "hamt_attn_varlen_cross_fwd")
    dq = torch.full((Mq, H), float("nan"), device=DEV, dtype=torch.bfloat16)
    dkv = torch.full_like(kv, 7.0)                                 # (rows no sample owns must stay untouched)
    L.check(lib.hamt_attn_varlen_cross_bwd(C.byref(d), p(q), p(kv[:, :H]), p(kv[:, H:]), p(cu_q), p(cu_k), n, p(pair), p(am), p(o), p(do), p(lse),
                                           p(dq), p(dkv[:, :H]), p(dkv[:, H:]), p(rng), ops._stream()), "hamt_attn_varlen_cross_bwd")
    torch.cuda.synchronize()
    assert bool(torch.isfinite(o).all()) and bool(torch.isfinite(dq).all())
    hv = lambda t, m: t.float().view(m, heads, 64).transpose(0, 1).double()
    for b in range(n):             # b = the fixed-stride sample, sq = its packed sequence
        sq = seq_of[b]
        if packed_side == "q":
            qi, q0, nq, k0, nk, add = sq, int(cu[sq]), plens[sq], b * Sf, Sf, mask[b].double()
        else:
            qi, q0, nq, k0, nk, add = b, b * Sf, Sf, int(cu[sq]), plens[sq], torch.zeros(plens[sq], device=DEV, dtype=torch.float64)
        qq = hv(q[q0:q0 + nq], nq).requires_grad_()
        kk = hv(kv[k0:k0 + nk, :H], nk).requires_grad_()
        vv = hv(kv[k0:k0 + nk, H:], nk).requires_grad_()
        sc = qq @ kk.transpose(-1, -2) * 0.125 + add
        out = torch.softmax(sc, -1) @ vv
        out.backward(hv(do[q0:q0 + nq], nq))
        tag = f"cross varlen ({packed_side}, pair map {with_pair})"
        close(o[q0:q0 + nq].float(), out.transpose(0, 1).reshape(nq, H), 1e-2, f"{tag} output pair {b}")
        close(lse.view(nseq, heads, Sq)[qi, :, :nq], torch.logsumexp(sc, -1), 1e-3, f"{tag} lse pair {b}")
        close(dq[q0:q0 + nq].float(), qq.grad.transpose(0, 1).reshape(nq, H), 2e-2, f"{tag} dq pair {b}")
        close(dkv[k0:k0 + nk, :H].float(), kk.grad.transpose(0, 1).reshape(nk, H), 2e-2, f"{tag} dk pair {b}")
        close(dkv[k0:k0 + nk, H:].float(), vv.grad.transpose(0, 1).reshape(nk, H), 2e-2, f"{tag} dv pair {b}")
    for i, ow in enumerate(owner):
        if ow >= 0 or plens[i] == 0:
            continue
        r0, r1 = int(cu[i]), int(cu[i + 1])
        if packed_side == "q":      # a filler query sequence: zeros out, zero dq
            assert float(o[r0:r1].float().abs().max()) == 0.0 and float(dq[r0:r1].float().abs().max()) == 0.0
        else:                       # filler key rows: not written
            assert bool((dkv[r0:r1] == 7.0).all())


# ------------------------------------------------------------------------------------------- two ranks on one GPU
def _two_rank_schedule(long_run):
    """(task sequence, per-task batch kwargs): the short plumbing run, or -- `long_run` -- 24 steps that alternate four tasks with
    DIFFERENT batch shapes at the production eps, so that the weight-gradient launch groups (and with them anything derived from a
    step's plan) differ from step to step while AdamW's moments decide the update"""
    if not long_run:
        return ["sap", "mlm", "sap", "mrc", "mlm", "sap"], {}, dict(lr=1e-3, eps=1.0)
    seq = ["sap", "mlm", "sar", "mrc", "mlm", "sap", "mrc", "sar"] * 3
    shapes = {"sap": dict(txt_len=20, hist_len=4), "mlm": dict(txt_len=28, hist_len=2), "sar": dict(txt_len=12, hist_len=5), "mrc": dict(txt_len=16, hist_len=3)}
    # lr = 1e-6: with sign-like Adam steps (eps = 1e-6 is far below every gradient) training is chaotic -- at lr = 2e-4 fp32
    # summation-order noise grew to 1e-3 of parameter difference within 24 steps in the UNSHARDED exchange too; tiny steps keep the
    # gradients of the compared runs equal, so that a parameter difference counts wrong updates in units of lr
    return seq, shapes, dict(lr=1e-6, eps=1e-6)


def _two_rank_batch(t, r, cfg, shapes):
    """rank r's batch of task t: its own seed AND its own padded shape (instructions 12 tokens longer, one history step fewer per rank
    index) -- the operands of the weight-gradient problems then have different row counts on the two ranks, as they do in a real
    job where every rank pads to its local longest instruction (the exchange schedule once depended on them)"""
    from vln_hamt_amd.synth import make_batch
    kw = dict(shapes.get(t, dict(txt_len=20, hist_len=4)))
    kw["txt_len"] += 12 * r if r < 2 else 4 * r + 2      # (<= 50 of the tiny model's 64 positions at rank 7)
    kw["hist_len"] = max(1, kw["hist_len"] - r % 4)
    return make_batch(t, 4, cfg, seed=100 * r + sum(map(ord, t)), ragged=True, device=DEV, **kw)


def _two_rank_worker(rank, world, port, out_dir, wire, use_graph, sharded=False, long_run=False, skip_at=None, acc=1):
    """One data-parallel rank (gloo carries the collectives of CUDA tensors, so two ranks can share the box's single
    GPU): the product's multi-GPU step on this rank's own batches.  use_graph == "wrapped": the reference's OWN loop lines
    (main_r2r.py:150-156, 237-281) around `wrap_model` -- no exchange call, no optimizer argument to clip_grad_norm_, a plain
    optimizer.step()."""
    import torch.distributed as dist
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from oracle.hamt_oracle import make_state_dict, pretrain_param_shapes
    from vln_hamt_amd.graph import GraphedTrainStep
    from vln_hamt_amd.optim import AdamW, clip_grad_norm_
    from vln_hamt_amd.optim.misc import NO_DECAY
    from vln_hamt_amd.parallel import OverlappedGradSync, ShardedGradSync, broadcast_params
    from vln_hamt_amd.synth import make_batch
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = tiny_cfg()
    sd = make_state_dict(pretrain_param_shapes(cfg), seed=5)
    m = build(cfg, sd, "bf16", train=True)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    named = list(m.named_parameters())
    seq, shapes, hyp = _two_rank_schedule(long_run)
    o = AdamW([{'params': [p for n, p in named if not any(nd in n for nd in NO_DECAY)], 'weight_decay': 0.01},
               {'params': [p for n, p in named if any(nd in n for nd in NO_DECAY)], 'weight_decay': 0.0}], betas=(0.9, 0.98), **hyp)
    wrapped = use_graph == "wrapped"
    if wrapped:
        from vln_hamt_amd.utils.misc import wrap_model
        os.environ["HAMT_SHARDED"] = "1" if sharded else "0"
        os.environ["HAMT_GRAD_WIRE"] = wire
        os.environ["HAMT_SYNC_GROUPS"] = "3"
        if rank == 1:
            with torch.no_grad():
                for p_ in m.parameters():      # rank 1 starts from other weights: wrap_model must bring rank 0's (DDP's broadcast at wrap)
                    p_.add_(0.01)
        model = wrap_model(m, torch.device("cuda", 0), 0, gradient_accumulation_steps=acc) if acc > 1 else wrap_model(m, torch.device("cuda", 0), 0)
        assert model is not m and model.module is m
        named = list(model.named_parameters())      # (main_r2r.py builds the optimizer from the WRAPPED model: names gain `module.`)
        o = AdamW([{'params': [p for n, p in named if not any(nd in n for nd in NO_DECAY)], 'weight_decay': 0.01},
                   {'params': [p for n, p in named if any(nd in n for nd in NO_DECAY)], 'weight_decay': 0.0}], betas=(0.9, 0.98), **hyp)
        o.zero_grad()
        o.step()                                     # main_r2r.py:229-230
        batches = {t: _two_rank_batch(t, rank, cfg, shapes) for t in set(seq)}
        # sharded: the next forward runs under the parameters' all-gathers of the update (ShardedGradSync.attach_gather) -- every read of a
        # parameter must sit on a stream that already waits for the range holding it (the read trap of the overlapped-update test)
        chk = _GateChecker()
        chk.__enter__()
        gated_seen = False
        try:
            for i_, t in enumerate(seq):
                gated_seen = gated_seen or bool(o._gather_ov is not None and o._gather_ov.pending)
                loss = model(batches[t], task=t, compute_loss=True)
                loss = loss.mean()
                if acc > 1:                          # main_r2r.py:242-250
                    loss = loss / acc
                loss.backward()
                if (i_ + 1) % acc:
                    continue
                clip_grad_norm_(model.parameters(), 5.0)
                if i_ != skip_at:                    # (skip_at: a loop that drops this pass -- NaN guard, early `continue` -- and only zeroes the gradients)
                    o.step()
                o.zero_grad()
            chk.__exit__()
            assert not chk.bad, sorted(set(chk.bad))
            assert gated_seen == bool(sharded), "sharded exchange: a forward pass must have started under pending all-gathers (and only then)"
            sync = model.grad_sync
            assert sync is not None and bool(getattr(sync, "sharded", False)) == bool(sharded), sync
            torch.cuda.synchronize()
            if sharded:
                torch.save(o._flat_p16.detach().cpu(), os.path.join(out_dir, f"shadow{rank}.pt"))
                torch.save(o._flat_p[o._n_shadow_only:].detach().cpu(), os.path.join(out_dir, f"fp32read{rank}.pt"))
                sync.gather_state()
            torch.cuda.synchronize()
            torch.save([], os.path.join(out_dir, f"exchanges{rank}.pt"))
            torch.save(o._flat_p.detach().cpu(), os.path.join(out_dir, f"params{rank}.pt"))
            torch.save((o._flat_m.detach().cpu(), o._flat_v.detach().cpu()), os.path.join(out_dir, f"moments{rank}.pt"))
        finally:
            chk.__exit__()
            model.close()
            dist.destroy_process_group()
        return
    o.materialize()
    broadcast_params(o)
    sync = (ShardedGradSync if sharded else OverlappedGradSync)(o, n_groups=3, wire=wire)
    batches = {t: _two_rank_batch(t, rank, cfg, shapes) for t in set(seq)}
    owned0 = sync.owned() if sharded else None
    sync.log = []
    losses_ = []
    try:
        if use_graph:
            gs = GraphedTrainStep(m, o, 5.0, grad_sync=sync)
            for t in seq:
                losses_.append(gs.step(t, batches[t], t).detach().clone())
                assert not sharded or sync.owned() == owned0, "ownership moved between steps"
        else:
            for t in seq:
                m(batches[t], t, True).mean().backward()
                sync(o)
                if sharded:
                    o.prepare_step()
                    sync.update(5.0)
                    assert sync.owned() == owned0, "ownership moved between steps"
                else:
                    clip_grad_norm_(m.parameters(), 5.0, optimizer=o)
                    o.step()
                o.zero_grad()
        torch.cuda.synchronize()
        if sharded:
            torch.save(o._flat_p16.detach().cpu(), os.path.join(out_dir, f"shadow{rank}.pt"))     # what the next forward would read
            torch.save(o._flat_p[o._n_shadow_only:].detach().cpu(), os.path.join(out_dir, f"fp32read{rank}.pt"))
            try:
                o.state_dict()
                raise AssertionError("state_dict() of a sharded optimizer must refuse before gather_state()")
            except RuntimeError:
                pass
            sync.gather_state()
            sd_ = o.state_dict()                   # what ModelSaver would write from rank 0 (utils/save.py:42-45): now complete
            assert len(sd_["state"]) > 0
        torch.cuda.synchronize()
        torch.save(list(sync.log), os.path.join(out_dir, f"exchanges{rank}.pt"))
        torch.save([float(x) for x in losses_], os.path.join(out_dir, f"losses{rank}.pt"))
        torch.save(o._flat_p.detach().cpu(), os.path.join(out_dir, f"params{rank}.pt"))
        torch.save((o._flat_m.detach().cpu(), o._flat_v.detach().cpu()), os.path.join(out_dir, f"moments{rank}.pt"))
        if sharded and long_run:      # what a resumed run would load (utils/save.py:42-45 -> optimizer.load_state_dict): the gathered state
            m_all, v_all, steps = o._flat_m.clone(), o._flat_v.clone(), o._steps.copy()
            o2 = AdamW([{'params': g_['params'], 'weight_decay': g_['weight_decay']} for g_ in o.param_groups], betas=(0.9, 0.98), **hyp)
            o2.load_state_dict(sd_)
            torch.cuda.synchronize()
            assert torch.equal(o2._flat_m, m_all) and torch.equal(o2._flat_v, v_all) and (o2._steps == steps).all()
    finally:
        sync.close()
        dist.destroy_process_group()


@pytest.mark.parametrize("sharded", [True, False])
def test_exchange_schedule_is_independent_of_the_batch(sharded):
    """the order and the bounds of the range collectives must not depend on anything rank-local.  Full-size
    model, the exchange object of a data-parallel job, eager steps over batches that differ the way two ranks' batches differ --
    padded to 80 / 56 tokens, a PACKED ragged batch (another row bucket), a SAP batch without history (other parameters queued),
    other tasks: every step must issue exactly the static ranges (arena layout and world size only) in arena order."""
    from oracle.hamt_oracle import OracleConfig, make_state_dict, pretrain_param_shapes
    from vln_hamt_amd import wgrad
    from vln_hamt_amd.optim import AdamW
    from vln_hamt_amd.optim.misc import NO_DECAY
    from vln_hamt_amd.parallel import OverlappedGradSync, ShardedGradSync
    from vln_hamt_amd.synth import make_batch, make_itm_rng
    if not wgrad.ENABLED:
        pytest.skip("HAMT_NO_DEFER_WGRAD")
    cfg = OracleConfig()
    m = build(cfg, make_state_dict(pretrain_param_shapes(cfg), seed=3), "bf16", train=True)
    named = list(m.named_parameters())
    o = AdamW([{'params': [p for n, p in named if not any(nd in n for nd in NO_DECAY)], 'weight_decay': 0.01},
               {'params': [p for n, p in named if any(nd in n for nd in NO_DECAY)], 'weight_decay': 0.0}], lr=1e-5, betas=(0.9, 0.98))
    o.materialize()
    sync = (ShardedGradSync if sharded else OverlappedGradSync)(o, n_groups=4, wire="bf16")
    cases = [("mlm", dict(txt_len=80, hist_len=5)), ("mlm", dict(txt_len=56, hist_len=3)), ("mlm", dict(txt_len=80, hist_len=7, ragged=True)),
             ("sap", dict(txt_len=80, hist_len=5)), ("sap", dict(txt_len=40, hist_len=0)), ("itm", dict(txt_len=80, hist_len=5)),
             ("sprel", dict(txt_len=72, hist_len=6, ragged=True)), ("mrc", dict(txt_len=80, hist_len=5))]
    seqs, plans = [], []
    try:
        static = list(sync._static_ranges())
        for i, (task, kw) in enumerate(cases):
            b = make_batch(task, 8, cfg, seed=40 + i, device=DEV, **kw)
            if task == "itm":
                r = make_itm_rng(b, seed=3)
                b["itm_neg_idxs"], b["itm_shuffled_pos_ids"] = r["neg_idxs"], r["shuffled_pos_ids"]
            sync.log = []
            m(b, task, True).mean().backward()
            sync(o)
            if sharded:
                o.prepare_step()
                sync.update(5.0)
            else:
                o.step()
            o.zero_grad()
            seqs.append(list(sync.log))
        torch.cuda.synchronize()
    finally:
        sync.close()
    assert len(static) >= 4
    for (task, kw), q in zip(cases, seqs):
        assert q == static, (task, kw, q, static)


@pytest.mark.parametrize("wire,use_graph,sharded,long_run", [("fp32", False, False, False), ("fp32", True, False, False), ("bf16", False, False, False),
                                                             ("fp32", False, True, False), ("fp32", True, True, False), ("bf16", True, True, False),
                                                             ("fp32", True, True, True), ("fp32", False, True, True), ("bf16", True, True, True),
                                                             ("fp32", True, False, True), ("bf16", True, False, True),
                                                             ("fp32", "wrapped", False, False), ("fp32", "wrapped", True, False),
                                                             ("bf16", "wrapped", True, True), ("bf16", "wrapped", False, True),
                                                             ("fp32", "wrapped", True, 2), ("fp32", "wrapped", False, 2),
                                                             ("fp32", "wrapped", True, 3), ("bf16", "wrapped", True, 3), ("fp32", "wrapped", False, 3)])
def test_two_ranks_on_one_gpu_match_averaged_gradients(tmp_path, wire, use_graph, sharded, long_run):
    """world_size = 2 for real: two processes, different batches, the product's overlapped exchange (gloo moves the
    CUDA tensors) -- against one process that computes both ranks' gradients on the same weights, averages them,
    clips and steps.  Both ranks must also end with identical parameters.  sharded: parallel.ShardedGradSync (reduce-scatter,
    AdamW over the owned slices, all-gather of the bf16 shadow / the fp32-read region) -- what each rank's next forward would
    read (shadow arena, fp32-read region) must be identical on both ranks BEFORE the masters are gathered.  long_run: 24 steps at the
    production eps = 1e-6 alternating four tasks with different batch shapes; exp_avg / exp_avg_sq (gathered: `gather_state`) are
    compared too, and every rank's owned segments must stay where they were at construction (ownership once followed
    each step's launch-group cuts, so elements changed owner and met stale moments)."""
    import socket
    import torch.multiprocessing as mp
    from oracle.hamt_oracle import make_state_dict, pretrain_param_shapes
    from vln_hamt_amd import wgrad
    from vln_hamt_amd.optim import AdamW, clip_grad_norm_
    from vln_hamt_amd.optim.misc import NO_DECAY
    from vln_hamt_amd.synth import make_batch
    if not wgrad.ENABLED:
        pytest.skip("HAMT_NO_DEFER_WGRAD")
    # long_run == 2: the short run with its THIRD pass dropped after backward + clip (no optimizer.step(), only zero_grad()): the pending
    # exchange must be discarded -- the next backward exchanges again instead of raising, and no stale norm reaches the next update
    # long_run == 3: gradient accumulation over two backward passes per update (main_r2r.py:242-250) -- the first pass of a pair only sums
    # into the local arena, the second exchanges the sums: ONE exchange per update, sharded or not (VERDICT r4 missing 4)
    skip_at = 2 if long_run == 2 else None
    acc = 2 if long_run == 3 else 1
    long_run = long_run is True
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_two_rank_worker, args=(2, port, str(tmp_path), wire, use_graph, sharded, long_run, skip_at, acc), nprocs=2, join=True)
    p0, p1 = torch.load(os.path.join(str(tmp_path), "params0.pt")), torch.load(os.path.join(str(tmp_path), "params1.pt"))
    assert torch.equal(p0, p1), "ranks diverged"
    x0, x1 = torch.load(os.path.join(str(tmp_path), "exchanges0.pt")), torch.load(os.path.join(str(tmp_path), "exchanges1.pt"))
    assert (len(x0) > 0 or use_graph == "wrapped") and x0 == x1, "the ranks issued different sequences of range exchanges"
    if sharded:
        for name in ("shadow", "fp32read"):
            a, b_ = torch.load(os.path.join(str(tmp_path), f"{name}0.pt")), torch.load(os.path.join(str(tmp_path), f"{name}1.pt"))
            assert torch.equal(a, b_), f"{name}: the ranks would run their next forward on different weights"
    # ---- reference: one process, both ranks' gradients per step
    cfg = tiny_cfg()
    sd = make_state_dict(pretrain_param_shapes(cfg), seed=5)
    m = build(cfg, sd, "bf16", train=True)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    named = list(m.named_parameters())
    seq, shapes, hyp = _two_rank_schedule(long_run)
    o = AdamW([{'params': [p for n, p in named if not any(nd in n for nd in NO_DECAY)], 'weight_decay': 0.01},
               {'params': [p for n, p in named if any(nd in n for nd in NO_DECAY)], 'weight_decay': 0.0}], betas=(0.9, 0.98), **hyp)
    o.materialize()
    bs = [{t: _two_rank_batch(t, r, cfg, shapes) for t in set(seq)} for r in range(2)]
    ref_losses = [[], []]
    for i_ in range(0, len(seq), acc):
        if i_ == skip_at:
            continue
        for t in seq[i_:i_ + acc]:                  # (rank 0's micro-batches of this update: gradients accumulate in .grad)
            l_ = m(bs[0][t], t, True).mean()
            ref_losses[0].append(float(l_))
            (l_ / acc).backward()
        o._pack_grads()
        g0 = o._flat_g.clone()
        o.zero_grad()
        for t in seq[i_:i_ + acc]:
            l_ = m(bs[1][t], t, True).mean()
            ref_losses[1].append(float(l_))
            (l_ / acc).backward()
        o._pack_grads()
        if long_run and wire == "bf16":
            # the wire's own arithmetic (DDP bf16_compress_hook: halve, round to bf16, sum in bf16): where the two ranks' gradients
            # nearly cancel, the rounded average differs from the fp32 one by 100 % or changes sign, and a sign-like Adam step
            # (eps = 1e-6) then differs by a whole lr -- 312 of 1.35 M elements against the fp32 average; that is the wire format,
            # not the exchange, so the reference rounds the same way
            o._flat_g.copy_(((g0 * 0.5).to(torch.bfloat16) + (o._flat_g * 0.5).to(torch.bfloat16)).float())
        else:
            o._flat_g.add_(g0).mul_(0.5)
        clip_grad_norm_(m.parameters(), 5.0, optimizer=o)
        o.step()
        o.zero_grad()
    torch.cuda.synchronize()
    for r_ in range(2):          # (diagnostic) per-step losses of the ranks against the one-process reference
        f_ = os.path.join(str(tmp_path), f"losses{r_}.pt")
        if os.path.exists(f_):
            got_l = torch.load(f_)
            if got_l:
                d_ = [abs(a - b) / max(1e-12, abs(b)) for a, b in zip(got_l, ref_losses[r_])]
                k_ = max(range(len(d_)), key=lambda i: d_[i])
                print(f"    rank {r_}: worst per-step loss difference {d_[k_]:.2e} at step {k_} ({seq[k_]}); first step over 1e-5: "
                      f"{next((i for i, x in enumerate(d_) if x > 1e-5), None)}")
    ref = o._flat_p.detach().cpu()
    mr, vr = o._flat_m.detach().cpu(), o._flat_v.detach().cpu()
    diff = (p0 - ref).abs()
    if long_run:
        # at eps = 1e-6 an element whose gradient is rounding noise around an exact zero (a key bias: softmax is invariant to a
        # per-query shift) takes +-lr steps of random sign -- in the reference too; such elements (sqrt(exp_avg_sq) below 1e-5 of
        # gradients that are 1e-3 .. 1e-1 here) say nothing about the exchange and are left out
        diff = torch.where(vr.sqrt() > 1e-5, diff, torch.zeros_like(diff))
    worst = float(diff.max())
    at = int(diff.argmax())
    who = next((n for (n, p_), off in zip(named, [o._offs[o._index_of[id(p_)]] for _, p_ in named]) if off <= at < off + p_.numel()), "?")
    print(f"[two ranks, wire={wire}, graph={use_graph}, sharded={sharded}] worst parameter difference after {len(seq)} steps: {worst:.2e} ({who})")
    m0, v0 = torch.load(os.path.join(str(tmp_path), "moments0.pt"))
    m1, v1 = torch.load(os.path.join(str(tmp_path), "moments1.pt"))
    tol = 2e-3 if (wire == "fp32" or long_run) else 3e-2
    em = float((m0 - mr).abs().max()) / float(mr.abs().max())
    ev = float((v0 - vr).abs().max()) / float(vr.abs().max())
    n_off = int((diff > 0.5 * hyp["lr"]).sum())
    print(f"    elements off by more than half a step: {n_off} of {diff.numel()}")
    if n_off:        # which parameters (diagnostic for a failing run)
        offs = [o._offs[o._index_of[id(p_)]] for _, p_ in named]
        rows = sorted(((int((diff[off:off + p_.numel()] > 0.5 * hyp["lr"]).sum()), float(diff[off:off + p_.numel()].max()), n) for (n, p_), off in zip(named, offs)), reverse=True)
        for cnt, mx, n in rows[:12]:
            if cnt:
                print(f"        {cnt:7d} elements, worst {mx:.2e}: {n}")
    print(f"    exp_avg / exp_avg_sq max difference relative to their scale: {em:.2e} / {ev:.2e}")
    if em >= tol or ev >= tol:       # where (diagnostic for a failing run)
        offs_ = [o._offs[o._index_of[id(p_)]] for _, p_ in named]
        dm = (m0 - mr).abs()
        rows_ = sorted(((float(dm[off:off + p_.numel()].max()), int((dm[off:off + p_.numel()] > tol * float(mr.abs().max())).sum()), n) for (n, p_), off in zip(named, offs_)), reverse=True)
        for mx, cnt, n in rows_[:8]:
            print(f"        exp_avg off by up to {mx:.2e} in {cnt} elements: {n}")
    # (long run: in units of the learning rate -- an element that met stale or missing moments is off by about lr per step)
    # (round 4 gated the bf16-wire long run by a COUNT of elements off by a step: the word / position tables' gradients were summed by float
    # atomics, one ulp of run-to-run difference flipped a rounded two-rank average here and there.  The scatter-adds are ordered now
    # -- hamt_scatter_add_rows_ordered, tools/grad_bitwise_repeat.py: no tensor varies -- and the exact bound is back)
    assert worst < (0.3e-6 if long_run else (2e-5 if wire == "fp32" else 2e-4)), (worst, who)
    if sharded:
        assert torch.equal(m0, m1) and torch.equal(v0, v1), "gather_state left the ranks with different moments"
    assert em < tol and ev < tol, (em, ev)


@pytest.mark.parametrize("wire,use_graph,sharded,acc", [("fp32", "wrapped", True, 1), ("fp32", "wrapped", False, 1), ("fp32", "wrapped", True, 2),
                                                        ("bf16", True, True, 1)])
def test_eight_ranks_on_one_gpu_match_averaged_gradients(tmp_path, wire, use_graph, sharded, acc):
    """world_size = 8 -- BASELINE config 3's -- for real: eight processes on the one GPU (gloo carries the CUDA tensors), every rank its own
    batches with its own padded shapes, through the product's exchange: the reference's loop lines around `wrap_model` (sharded
    reduce-scatter / owned-slice AdamW / all-gather, and all-reduce; one case with gradient accumulation over two passes per update), and
    the captured sharded step with the bf16 wire.  Against ONE process that computes the eight ranks' gradients on the same weights,
    averages, clips and steps.  World-8 `shard_cuts` / ownership / all-gathers had CPU invariants only; RCCL over xGMI
    itself cannot run here (one GPU per box)."""
    import socket
    import torch.multiprocessing as mp
    from oracle.hamt_oracle import make_state_dict, pretrain_param_shapes
    from vln_hamt_amd import wgrad
    from vln_hamt_amd.optim import AdamW, clip_grad_norm_
    from vln_hamt_amd.optim.misc import NO_DECAY
    if not wgrad.ENABLED:
        pytest.skip("HAMT_NO_DEFER_WGRAD")
    W = 8
    s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]; s_.close()
    mp.spawn(_two_rank_worker, args=(W, port, str(tmp_path), wire, use_graph, sharded, False, None, acc), nprocs=W, join=True)
    ps = [torch.load(os.path.join(str(tmp_path), f"params{r}.pt")) for r in range(W)]
    for r in range(1, W):
        assert torch.equal(ps[0], ps[r]), f"rank {r} diverged from rank 0"
    xs = [torch.load(os.path.join(str(tmp_path), f"exchanges{r}.pt")) for r in range(W)]
    assert all(x == xs[0] for x in xs), "the ranks issued different sequences of range exchanges"
    if sharded:
        for name in ("shadow", "fp32read"):
            a = torch.load(os.path.join(str(tmp_path), f"{name}0.pt"))
            for r in range(1, W):
                assert torch.equal(a, torch.load(os.path.join(str(tmp_path), f"{name}{r}.pt"))), f"{name}: rank {r} would run its next forward on other weights"
    cfg = tiny_cfg()
    m = build(cfg, make_state_dict(pretrain_param_shapes(cfg), seed=5), "bf16", train=True)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    named = list(m.named_parameters())
    seq, shapes, hyp = _two_rank_schedule(False)
    o = AdamW([{'params': [p for n, p in named if not any(nd in n for nd in NO_DECAY)], 'weight_decay': 0.01},
               {'params': [p for n, p in named if any(nd in n for nd in NO_DECAY)], 'weight_decay': 0.0}], betas=(0.9, 0.98), **hyp)
    o.materialize()
    bs = [{t: _two_rank_batch(t, r, cfg, shapes) for t in set(seq)} for r in range(W)]
    for i_ in range(0, len(seq), acc):
        tot = torch.zeros_like(o._flat_g)
        share = (lambda g: (g / W).to(torch.bfloat16).float()) if wire == "bf16" else (lambda g: g / W)    # (bf16 wire: DDP bf16_compress_hook's rounding)
        for r in range(W):
            for t in seq[i_:i_ + acc]:
                (m(bs[r][t], t, True).mean() / acc).backward()
            o._pack_grads()
            if r < W - 1:
                tot += share(o._flat_g)
                o.zero_grad()
            else:
                o._flat_g.copy_(share(o._flat_g) + tot)
        clip_grad_norm_(m.parameters(), 5.0, optimizer=o)
        o.step()
        o.zero_grad()
    torch.cuda.synchronize()
    ref = o._flat_p.detach().cpu()
    worst = float((ps[0] - ref).abs().max())
    mr, vr = o._flat_m.detach().cpu(), o._flat_v.detach().cpu()
    m0, v0 = torch.load(os.path.join(str(tmp_path), "moments0.pt"))
    em, ev = float((m0 - mr).abs().max()) / float(mr.abs().max()), float((v0 - vr).abs().max()) / float(vr.abs().max())
    print(f"[eight ranks, wire={wire}, graph={use_graph}, sharded={sharded}, acc={acc}] worst parameter difference after {len(seq) // acc} updates: "
          f"{worst:.2e}; exp_avg / exp_avg_sq {em:.2e} / {ev:.2e}")
    tol = 2e-3 if wire == "fp32" else 3e-2
    # Eight addends: gloo's ring sums them in another order than this loop, the averaged gradients differ in the last bit, and within a few
    # steps a ReLU of the SAP head or a bf16 rounding of an activation flips on one side only -- a handful of moment elements (43 of 16 384 of
    # next_action.net.0.weight, 27 of an embedding row: round 6, identical for the sharded and the all-reduce exchange) then sit 1 - 3 % of
    # the scale apart while the parameters agree to 2e-6.  (Two addends commute: the two-rank test never sees it.)  An exchange fault -- a
    # rank's share missing, a range scaled or placed wrongly -- moves whole ranges: the gate is the FRACTION of elements beyond the bound.
    frac_m = float(((m0 - mr).abs() > tol * float(mr.abs().max())).float().mean())
    frac_v = float(((v0 - vr).abs() > tol * float(vr.abs().max())).float().mean())
    print(f"    fraction of exp_avg / exp_avg_sq elements beyond {tol:g} of the scale: {frac_m:.2e} / {frac_v:.2e}")
    if em >= tol or ev >= tol:       # where (diagnostic)
        offs_ = [o._offs[o._index_of[id(p_)]] for _, p_ in named]
        dm = (m0 - mr).abs()
        rows_ = sorted(((float(dm[off:off + p_.numel()].max()), int((dm[off:off + p_.numel()] > tol * float(mr.abs().max())).sum()), p_.numel(), n) for (n, p_), off in zip(named, offs_)), reverse=True)
        for mx, cnt, num, n in rows_[:10]:
            print(f"        exp_avg off by up to {mx:.2e} in {cnt} of {num} elements: {n}")
    assert worst < (2e-5 if wire == "fp32" else 2e-4), worst
    assert frac_m < 1e-3 and frac_v < 1e-3 and em < 0.1 and ev < 0.1, (em, ev, frac_m, frac_v)
    if sharded:
        for r in range(1, W):
            mr_, vr_ = torch.load(os.path.join(str(tmp_path), f"moments{r}.pt"))
            assert torch.equal(m0, mr_) and torch.equal(v0, vr_), f"gather_state left rank {r} with other moments"
