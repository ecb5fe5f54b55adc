"""Row N4: batch collation.  CPU: the numpy oracle against goldens produced by the reference's own *_collate
functions (oracle/gen_goldens.py gen_collate), and the host packing logic of vln_hamt_amd.data (layout, prefix tables,
payload bytes) unpacked by a plain-numpy reader.  GPU: PackedBatch.to_device (hamt_unpack_padded / hamt_seq_masks) bit
exact against the same goldens, against the oracle at bench sizes, through DataLoader(pin_memory=True) + PrefetchLoader."""
import numpy as np
import pytest
import torch

from _util import load_npz

CASES = [("mlm", 5, 11, False), ("mrc", 4, 12, False), ("itm", 3, 13, False), ("sap", 5, 14, False), ("sap", 3, 15, True),
         ("sar", 4, 16, False), ("sprel", 4, 17, False), ("sprel", 2, 18, True)]
DIMS = dict(feat=8, ang=4, prob=10, max_txt=12, max_hist=4, views=36)


def _samples(task, n, seed, first, **dims):
    from vln_hamt_amd.synth import make_samples
    return make_samples(task, n, seed, first_step=first, **(dims or DIMS))


def _check_against_golden(store, tag, got):
    keys = {k.split("/")[1] for k in store if k.startswith(tag + "/")}
    assert keys == set(got), (tag, keys ^ set(got))
    for k in keys:
        if f"{tag}/{k}/none" in store:
            assert got[k] is None, (tag, k)
        elif f"{tag}/{k}/list" in store:
            assert isinstance(got[k], list) and len(got[k]) == int(store[f"{tag}/{k}/list"])
        else:
            exp = store[f"{tag}/{k}"]
            g = got[k].cpu().numpy() if torch.is_tensor(got[k]) else got[k]
            assert g.dtype == exp.dtype and g.shape == exp.shape and np.array_equal(g, exp), (tag, k, g.dtype, exp.dtype, g.shape, exp.shape)


@pytest.mark.parametrize("task,n,seed,first", CASES)
def test_collate_oracle_matches_reference_goldens(task, n, seed, first):
    from oracle.collate_oracle import COLLATE
    _check_against_golden(load_npz("collate.npz"), f"{task}{seed}", COLLATE[task](_samples(task, n, seed, first)))


def _numpy_unpack(pb):
    """what the device kernels do, restated over the host buffer (test-only reader of the PackedBatch layout)"""
    raw = pb.buf.numpy()
    out = dict(pb.lists)
    pre = {f: raw[o:o + 4 * (pb.B + 1)].view(np.int32) for f, o in pb.prefix_off.items()}
    for f, lens in pb.lens.items():
        assert np.array_equal(np.diff(pre[f]), lens) and pre[f][0] == 0
    from vln_hamt_amd.data.collate import HIST_FIELDS
    for name, (off, row_shape, dtype, fam, pad) in pb.fields.items():
        if pb.hist_none and name in HIST_FIELDS:
            out[name] = None
            continue
        npdt = torch.empty((), dtype=dtype).numpy().dtype
        rows, maxlen = int(pre[fam][-1]), max(pb.lens[fam])
        n_row = int(np.prod(row_shape, dtype=np.int64))
        src = raw[off:off + rows * n_row * npdt.itemsize].view(npdt).reshape((rows,) + tuple(row_shape))
        dst = np.frombuffer(bytes([pad]) * (pb.B * maxlen * n_row * npdt.itemsize), dtype=npdt).reshape((pb.B, maxlen) + tuple(row_shape)).copy()
        for b in range(pb.B):
            dst[b, :pb.lens[fam][b]] = src[pre[fam][b]:pre[fam][b + 1]]
        out[name] = dst
    for fam, add in (("txt", 0), ("hist", 1), ("ob", 0)):
        if fam in pb.lens:
            lens = np.asarray(pb.lens[fam], dtype=np.int64) + add
            out[f"{fam}_masks"] = np.arange(lens.max())[None] < lens[:, None]
            out[f"{fam}_lens"] = lens
    for name, (off, shape, dtype) in pb.per_sample.items():
        npdt = torch.empty((), dtype=dtype).numpy().dtype
        out[name] = raw[off:off + int(np.prod(shape)) * npdt.itemsize].view(npdt).reshape(shape)
    return out


@pytest.mark.parametrize("task,n,seed,first", CASES)
def test_packed_batch_layout_reproduces_reference_goldens(task, n, seed, first):
    """host side of the product path: packing is loss-free and carries everything the device needs"""
    from vln_hamt_amd.data import COLLATE
    pb = COLLATE[task](_samples(task, n, seed, first))
    assert pb.buf.dtype == torch.uint8 and all(o % 64 == 0 for o, *_ in pb.fields.values())
    _check_against_golden(load_npz("collate.npz"), f"{task}{seed}", _numpy_unpack(pb))


def test_device_collation_refuses_cpu():
    from vln_hamt_amd import _lib
    from vln_hamt_amd.data import sap_collate
    with pytest.raises(_lib.HamtError):
        sap_collate(_samples("sap", 2, 1, False)).to_device("cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("task,n,seed,first", CASES)
def test_device_collation_matches_reference_goldens(task, n, seed, first):
    from vln_hamt_amd.data import COLLATE
    got = COLLATE[task](_samples(task, n, seed, first)).to_device("cuda")
    assert all(v is None or isinstance(v, list) or v.is_cuda for v in got.values())
    _check_against_golden(load_npz("collate.npz"), f"{task}{seed}", got)


@pytest.mark.gpu
@pytest.mark.parametrize("task", ["mlm", "mrc", "itm", "sap", "sar", "sprel"])
def test_device_collation_bench_size_vs_oracle(task):
    """B = 64 at the real widths (768-d features, 1000 classes, 80 tokens), pinned transport, and again into static
    targets (`out=`, the captured graph's input tensors): bit exact against the oracle."""
    from oracle.collate_oracle import COLLATE as ORACLE
    from vln_hamt_amd.data import COLLATE
    dims = dict(feat=768, ang=4, prob=1000, max_txt=80, max_hist=5, views=36)
    smp = _samples(task, 64, 77, False, **dims)
    exp = ORACLE[task](_samples(task, 64, 77, False, **dims))
    pb = COLLATE[task](smp).pin_memory()
    assert pb.buf.is_pinned()
    got = pb.to_device("cuda")
    static = {k: torch.full_like(v, 3) for k, v in got.items() if torch.is_tensor(v)}
    got2 = COLLATE[task](smp).to_device("cuda", out=static)
    for k, e in exp.items():
        if isinstance(e, list):
            continue
        for g in (got[k], got2[k]):
            a = g.cpu().numpy()
            assert a.dtype == e.dtype and np.array_equal(a, e), (task, k)
        assert got2[k].data_ptr() == static[k].data_ptr(), k          # written in place


@pytest.mark.gpu
def test_prefetch_loader_with_pinning_dataloader():
    """torch DataLoader(pin_memory=True, collate_fn=sap_collate) -> PrefetchLoader: same batches, in order, as collating
    directly; the DataLoader pins exactly the packed buffer (PackedBatch.pin_memory)."""
    from oracle.collate_oracle import sap_collate as oracle_collate
    from vln_hamt_amd.data import PrefetchLoader, sap_collate
    smp = _samples("sap", 22, 5, False)
    dl = torch.utils.data.DataLoader(smp, batch_size=4, shuffle=False, collate_fn=sap_collate, pin_memory=True, num_workers=0)
    pl = PrefetchLoader(dl, torch.device("cuda"))
    assert len(pl) == 6 and pl.batch_size == 4                       # attribute forwarding (loader.py:122-124)
    n = 0
    for i, batch in enumerate(pl):
        exp = oracle_collate(_samples("sap", 22, 5, False)[4 * i:4 * i + 4])
        for k, e in exp.items():
            if e is None:
                assert batch[k] is None
            elif not isinstance(e, list):
                assert np.array_equal(batch[k].cpu().numpy(), e), (i, k)
        n += 1
    assert n == 6
