"""Row N4 (input pipeline), CPU: the product's trajectory / view-feature readers, per-sample input builder and task-mixing loader
against goldens produced by the REFERENCE's own classes (oracle/gen_goldens.py r2r_data / loader: pretrain_src/data/r2r_data.py
`MultiStepNavData`, data/loader.py `MetaLoader` / `build_dataloader`) on the committed tiny dataset tests/golden/r2r_tiny/."""
import os
import sys
import types

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")
TINY = os.path.join(GOLD, "r2r_tiny")
DIMS = dict(image_feat_size=16, image_prob_size=10, angle_feat_size=4)
CASES = [(0, 0, 0, True, False, True, True, False), (0, 1, 2, True, True, True, True, False), (0, 0, 3, True, False, True, True, True),
         (1, 0, 1, False, True, False, False, None), (1, 0, 4, True, True, True, True, False), (1, 0, 4, True, False, True, True, True),
         (2, 0, 2, True, False, True, True, True), (2, 1, 3, True, True, True, True, False), (3, 0, 1, True, False, True, True, False),
         (3, 0, 0, True, False, True, False, True)]


def _kw(**extra):
    d = dict(traj_files=[os.path.join(TINY, "traj.jsonl"), os.path.join(TINY, "traj2.jsonl")], img_ft_file=os.path.join(TINY, "img_fts.npz"),
             scanvp_cands_file=os.path.join(TINY, "scanvp_cands.json"), connectivity_dir=TINY, max_txt_len=12, max_act_len=6, **DIMS)
    d.update(extra)
    return d


@pytest.fixture(scope="module")
def gold():
    return dict(np.load(os.path.join(GOLD, "r2r_data.npz")))


def _same(got, want, what):
    want = np.asarray(want)
    got = np.asarray(got)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    if want.dtype.kind in "iub":
        assert got.dtype.kind in "iub" and np.array_equal(got, want), what                    # index work: bit exact
    elif want.dtype.kind == "U":
        assert str(got) == str(want), what
    else:
        assert got.dtype == want.dtype or got.ndim == 0, (what, got.dtype, want.dtype)
        assert np.array_equal(got, want), (what, float(np.abs(got.astype(np.float64) - want).max()))     # same arithmetic: bit exact


@pytest.mark.parametrize("pano", [True, False])
def test_multistep_nav_data_matches_the_reference_class(gold, pano):
    from vln_hamt_amd.data.r2r_data import MultiStepNavData
    db = MultiStepNavData(hist_enc_pano=pano, **_kw())
    tag = "pano" if pano else "nopano"
    _same(db.traj_refer, gold[f"{tag}/traj_refer"], "traj_refer")
    _same(db.traj_step_refer, gold[f"{tag}/traj_step_refer"], "traj_step_refer")
    for c, (i, j, t, ob, probs, act, prog, cand) in enumerate(CASES):
        out = db.get_input(i, j, t, return_ob=ob, return_hist_img_probs=probs, return_ob_action=act, return_ob_progress=prog, ob_cand_pano_view=cand)
        keys = {k.split("/")[2] for k in gold if k.startswith(f"{tag}/case{c}/")}
        assert keys == set(out), (c, keys ^ set(out))
        for k, v in out.items():
            base = f"{tag}/case{c}/{k}"
            if base + "/emptylist" in gold:
                assert isinstance(v, list) and len(v) == 0, base
            elif base + "/str" in gold:
                assert v == str(gold[base + "/str"]), base
            elif k == "ob_progress":            # distances are sums of float64 edge lengths: equal up to the order of tie-breaking
                assert abs(float(v) - float(gold[base])) < 1e-9, base
            else:
                _same(v, gold[base], base)


def test_angle_tables_and_graph_distances(gold):
    from vln_hamt_amd.data import r2r_data as rd
    _same(np.stack(rd.get_all_point_angle_feature(4), 0), gold["angle_features"], "angle features")
    _same(np.stack(rd.get_all_point_rel_angles(), 0), gold["rel_angles"], "relative angles")
    graphs, dist = rd.load_nav_graphs(TINY)
    for scan, d in dist.items():
        vps = sorted(d)
        assert "excluded" not in d
        got = np.asarray([[d[a][b] for b in vps] for a in vps])
        assert np.abs(got - gold[f"dist/{scan}"]).max() < 1e-9, scan
    assert rd.softmax(np.array([[0.0, 0.0]]))[0, 0] == 0.5


def test_validation_subsample_uses_the_global_numpy_stream(gold):
    from vln_hamt_amd.data.r2r_data import MultiStepNavData
    np.random.seed(5)
    db = MultiStepNavData(val_sample_num=4, **_kw())
    _same(db.traj_refer, gold["val/traj_refer"], "val traj_refer")
    _same(db.traj_step_refer, gold["val/traj_step_refer"], "val traj_step_refer")


def test_view_feature_store_backends(tmp_path):
    from vln_hamt_amd.data.r2r_data import ViewFeatureStore, read_jsonl
    arrays = dict(np.load(os.path.join(TINY, "img_fts.npz")))
    key = sorted(arrays)[3]
    for k, a in arrays.items():
        np.save(os.path.join(str(tmp_path), k + ".npy"), a)
    for store in (ViewFeatureStore(os.path.join(TINY, "img_fts.npz")), ViewFeatureStore(str(tmp_path), in_memory=True)):
        got = store.get(key)
        assert got.dtype == np.float32 and np.array_equal(got, arrays[key].astype(np.float32)) and key in store and "nope" not in store
    assert store.get(key) is store.get(key)                                      # in_memory: cached block
    # HDF5 is read through h5py, which this image does not ship: the reader must say so, not substitute anything
    try:
        import h5py  # noqa: F401
    except ImportError:
        with pytest.raises(ImportError, match="h5py"):
            ViewFeatureStore(os.path.join(str(tmp_path), "features.hdf5")).get(key)
    items = list(read_jsonl(os.path.join(TINY, "traj2.jsonl")))                  # trailing blank line skipped
    assert len(items) == 1 and items[0]["scan"] == "scanB" and "guide_path" in items[0]


# ------------------------------------------------------------------------------------------------ MetaLoader / build_dataloader
RATIOS = {"mlm": 5, "sap": 1, "itm": 2}


def _sets():
    from torch.utils.data import TensorDataset
    return {"mlm": TensorDataset(torch.arange(0, 23)), "sap": TensorDataset(torch.arange(100, 107)), "itm": TensorDataset(torch.arange(200, 210))}


def _opts(**kw):
    d = dict(train_batch_size=4, val_batch_size=3, local_rank=-1, n_workers=0, pin_mem=False)
    d.update(kw)
    return types.SimpleNamespace(**d)


def _col(items):
    return torch.stack([it[0] for it in items])


@pytest.mark.parametrize("accum", [1, 2])
def test_metaloader_reference_sampling_reproduces_the_reference_sequence(accum):
    """same torch seed -> the same task draws, batches, epoch restarts and reshuffles as data/loader.py's MetaLoader"""
    from vln_hamt_amd.data.loader import MetaLoader, build_dataloader
    g = np.load(os.path.join(GOLD, "loader.npz"))
    torch.manual_seed(100 + accum)
    sets = _sets()
    loaders = {n: (build_dataloader(n, sets[n], _col, True, _opts())[0], r, (lambda e: None)) for n, r in RATIOS.items()}
    ml = MetaLoader(loaders, accum_steps=accum, distributed=False, device=None)
    assert ml.sampling == "reference"
    names, flat, lens = [], [], []
    for step, (task, batch) in enumerate(ml):
        if step == 60:
            break
        names.append(list(RATIOS).index(task))
        flat += batch.tolist()
        lens.append(len(batch))
    assert names == g[f"accum{accum}/task"].tolist()
    assert lens == g[f"accum{accum}/lens"].tolist() and flat == g[f"accum{accum}/ids"].tolist()


def test_build_dataloader_attributes_match_the_reference():
    from vln_hamt_amd.data.loader import build_dataloader
    g = np.load(os.path.join(GOLD, "loader.npz"))
    attrs = []
    for task in ("mlm", "itm"):
        for train in (True, False):
            ld, pre = build_dataloader(task, _sets()["mlm"], _col, train, _opts())
            assert pre(3) is None
            attrs.append([ld.batch_size, int(type(ld.sampler).__name__ == "RandomSampler"), int(ld.drop_last), ld.num_workers, int(ld.pin_memory)])
    assert attrs == g["attrs"].tolist()


def test_metaloader_shared_seed_schedule_needs_no_collective():
    """distributed default: the task of draw k is a pure function of (seed, k) -- two 'ranks' built independently agree on 400
    draws without talking, accum_steps repeats a draw, and the mix follows the ratios."""
    from vln_hamt_amd.data.loader import MetaLoader, build_dataloader

    def tasks(seed, accum, n):
        sets = _sets()
        loaders = {k: (build_dataloader(k, sets[k], _col, True, _opts())[0], r, (lambda e: None)) for k, r in RATIOS.items()}
        ml = MetaLoader(loaders, accum_steps=accum, distributed=True, device=None, seed=seed)      # (no process group: nothing is broadcast)
        assert ml.sampling == "shared_seed"
        out = []
        for step, (task, batch) in enumerate(ml):
            if step == n:
                break
            out.append(task)
        return out

    torch.manual_seed(1)
    a = tasks(7, 1, 400)
    torch.manual_seed(2)            # a different torch stream on the "other rank": irrelevant to the schedule
    b = tasks(7, 1, 400)
    assert a == b and tasks(8, 1, 50) != a[:50]
    frac = a.count("mlm") / len(a)
    assert abs(frac - 5 / 8) < 0.08, frac
    c = tasks(7, 2, 40)
    assert c[0::2] == c[1::2] and c[0::2] == a[:20]


# ------------------------------------------------------------------------------------------------ the six task datasets
TASK_DS_CASES = [("mlm", [0, 2, 5]), ("mrc", [1, 3, 4]), ("itm", [0, 4]), ("sap", [0, 3, 7, 11, 14]), ("sar", [1, 2, 9, 13]), ("sprel", [0, 5, 8, 12])]


@pytest.mark.parametrize("task,idxs", TASK_DS_CASES)
def test_task_datasets_match_the_reference_classes(task, idxs):
    """Same python / numpy / torch seeds -> the same items as the reference's MlmDataset ... SprelDataset (r2r_tasks.py): word
    masking and labels, region masks with zeroed views, view / angle kills, action targets, SPREL anchors and targets.  The
    items then go through this package's own *_collate + device unpack (the path a DataLoader takes)."""
    import random
    from vln_hamt_amd.data import r2r_tasks as T
    from vln_hamt_amd.data.r2r_data import MultiStepNavData
    g = np.load(os.path.join(GOLD, "r2r_tasks.npz"))
    db = MultiStepNavData(**_kw())
    tok = types.SimpleNamespace(cls_token_id=101, sep_token_id=102, mask_token_id=103, pad_token_id=0)
    ds = {"mlm": lambda: T.MlmDataset(db, tok), "mrc": lambda: T.MrcDataset(db, tok, 0.5), "itm": lambda: T.ItmDataset(db, tok),
          "sap": lambda: T.SapDataset(db, tok, 0.3, 0.43), "sar": lambda: T.SarDataset(db, tok, 0.3, 0.43),
          "sprel": lambda: T.SprelDataset(db, tok, 0.3, 0.43)}[task]()
    assert len(ds) == int(g[f"{task}/len"])
    items = []
    for i in idxs:
        random.seed(1000 + i); np.random.seed(2000 + i); torch.manual_seed(3000 + i)
        item = ds[i]
        items.append(item)
        keys = {k.split("/")[2] for k in g.files if k.startswith(f"{task}/{i}/")}
        assert keys == set(item), (task, i, keys ^ set(item))
        for k, v in item.items():
            want = g[f"{task}/{i}/{k}"]
            got = v.numpy() if torch.is_tensor(v) else np.asarray(v)
            assert got.shape == want.shape and got.dtype == want.dtype, (task, i, k, got.dtype, want.dtype, got.shape, want.shape)
            assert np.array_equal(got, want), (task, i, k)
    # the items are what the collate functions take (host packing only: no GPU here)
    packed = getattr(T, f"{task}_collate")(items)
    assert packed.B == len(items) and packed.nbytes > 0


@pytest.mark.gpu
def test_files_to_training_steps_end_to_end():
    """The whole input side in front of the model, as main_r2r.py wires it (:156-206, 231-281): trajectory / feature files ->
    MultiStepNavData -> the task datasets -> build_dataloader (packing collate, pinned) -> MetaLoader -> PrefetchLoader -> device
    batches -> MultiStepNavCMTPreTraining.forward / backward / clip / AdamW.  Device batches must equal the reference-style collation
    of the same items (numpy oracle), losses must be finite and the six tasks must all have been drawn."""
    import random
    from oracle.collate_oracle import COLLATE as ORACLE_COLLATE
    from oracle.hamt_oracle import OracleConfig, make_state_dict, pretrain_param_shapes
    from vln_hamt_amd import data as D
    from vln_hamt_amd.model.pretrain_cmt import MultiStepNavCMTPreTraining
    from vln_hamt_amd.modeling import HamtConfig
    from vln_hamt_amd.optim import AdamW, clip_grad_norm_
    dev = torch.device("cuda")
    db = D.MultiStepNavData(**_kw(max_txt_len=20))
    tok = types.SimpleNamespace(cls_token_id=101, sep_token_id=102, mask_token_id=103, pad_token_id=0)
    dsets = {"mlm": D.MlmDataset(db, tok), "mrc": D.MrcDataset(db, tok, 0.5), "itm": D.ItmDataset(db, tok),
             "sap": D.SapDataset(db, tok, 0.3, 0.43), "sar": D.SarDataset(db, tok, 0.3, 0.43), "sprel": D.SprelDataset(db, tok, 0.3, 0.43)}
    # a padded device batch == the reference-style collation of the same items
    random.seed(3); np.random.seed(3)
    items = [dsets["sap"][i] for i in (0, 4, 9, 13)]
    got = D.move_to_cuda(D.sap_collate(items), dev)
    want = ORACLE_COLLATE["sap"](items)
    for k, v in want.items():
        if v is None:
            assert got[k] is None, k
        elif isinstance(v, np.ndarray):
            g = got[k].cpu().numpy()
            assert g.dtype == v.dtype and np.array_equal(g, v), k
    # the training loop
    ocfg = OracleConfig.tiny(hidden_size=128, num_attention_heads=2, intermediate_size=256, image_feat_size=16, image_prob_size=10, vocab_size=30522)
    kw = dict(vars(ocfg))
    kw["pretrain_tasks"] = set(ocfg.pretrain_tasks)
    model = MultiStepNavCMTPreTraining(HamtConfig(hamt_precision="bf16", **kw))
    model.load_state_dict(make_state_dict(pretrain_param_shapes(ocfg), seed=9))
    model = model.to(dev).train()
    opts = types.SimpleNamespace(train_batch_size=4, val_batch_size=4, local_rank=-1, n_workers=0, pin_mem=True)
    ratios = {"mlm": 5, "sap": 1, "sar": 1, "sprel": 1, "mrc": 2, "itm": 2}
    torch.manual_seed(0)
    loaders = {t: (D.build_dataloader(t, dsets[t], D.COLLATE[t], True, opts)[0], r, (lambda e: None)) for t, r in ratios.items()}
    meta = D.PrefetchLoader(D.MetaLoader(loaders, accum_steps=1, distributed=False, device=dev), dev, text_pack=True)
    opt = AdamW([{"params": list(model.parameters()), "weight_decay": 0.01}], lr=5e-5, betas=(0.9, 0.98))
    seen, losses, n_packed = set(), [], 0
    for step, (task, batch) in enumerate(meta):
        if step == 40:
            break
        seen.add(task)
        assert batch["txt_ids"].is_cuda and batch["txt_masks"].dtype == torch.bool
        n_packed += "txt_pack_idx" in batch
        loss = model(batch, task=task, compute_loss=True).mean()
        loss.backward()
        clip_grad_norm_(model.parameters(), 5.0, optimizer=opt)
        opt.step()
        opt.zero_grad()
        losses.append(float(loss))
    assert seen == set(ratios), seen
    assert all(np.isfinite(l) for l in losses), losses


def test_text_pack_plan_invariants():
    """synth.text_pack_plan: the packed rows are the real tokens in row-major order, then filler rows up to the bucket; every row belongs
    to exactly one sequence of at most L rows; `cu` has ONE length per (B, L, bucket) whatever the lengths (unused filler slots are
    empty sequences), so a captured step's key depends on the bucketed row count only; unpack sends real positions to their own row."""
    from vln_hamt_amd.synth import text_pack_plan
    rng = np.random.default_rng(3)
    L, B = 80, 24
    shapes = set()
    for bucket in (128, 512):
        for _ in range(20):
            lens = rng.integers(1, L + 1, B)
            lens[rng.integers(B)] = L
            plan = text_pack_plan(lens, L, bucket=bucket)
            if plan is None:
                assert (int(lens.sum()) + bucket - 1) // bucket * bucket >= B * L
                continue
            pack, cu, unpack = (t.numpy() for t in plan)
            M = int(lens.sum())
            assert pack.shape[0] % bucket == 0 and pack.shape[0] - M < bucket and int(cu[-1]) == pack.shape[0]
            valid = (np.arange(L)[None] < lens[:, None]).reshape(-1)
            assert (pack[:M] == np.flatnonzero(valid)).all()
            assert (np.diff(cu) >= 0).all() and (np.diff(cu) <= L).all() and (cu[:B + 1] == np.concatenate([[0], np.cumsum(lens)])).all()
            assert cu.shape[0] == B + 1 + (bucket + L - 1) // L
            assert (unpack[np.flatnonzero(valid)] == np.arange(M)).all() and unpack.min() >= 0 and unpack.max() < M
            shapes.add((bucket, cu.shape[0]))
    assert len(shapes) == 2
