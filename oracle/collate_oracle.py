"""CPU restatement (numpy) of the reference's batch collation (pretrain_src/data/common.py:5-29 and the six *_collate
functions of pretrain_src/data/r2r_tasks.py:95-125, 202-226, 268-288, 343-380, 444-482, 559-596).

TEST INFRASTRUCTURE: only tests/, tools' CPU-baseline legs and oracle/gen_goldens.py import this file; the product
(vln_hamt_amd.data) packs ragged samples into one pinned buffer and pads ON THE DEVICE.  Pinned by
tests/golden/collate.npz, which oracle/gen_goldens.py produced by running the reference's own *_collate functions."""
import numpy as np


def _np(t):
    return t.numpy() if hasattr(t, "numpy") else np.asarray(t)


def pad_tensors(tensors, lens=None, pad=0):
    """common.py:5-20: B x [T, ...] -> [B, max T, ...]"""
    arrs = [_np(t) for t in tensors]
    lens = [a.shape[0] for a in arrs] if lens is None else lens
    out = np.full((len(arrs), max(lens)) + arrs[0].shape[1:], pad, dtype=arrs[0].dtype)
    for i, (a, l) in enumerate(zip(arrs, lens)):
        out[i, :l] = a
    return out


def pad_sequence(seqs, padding_value=0):
    """torch.nn.utils.rnn.pad_sequence(batch_first=True) as the collates call it"""
    return pad_tensors(seqs, pad=padding_value)


def gen_seq_masks(seq_lens, max_len=None):
    """common.py:22-29"""
    seq_lens = np.asarray(seq_lens)
    max_len = int(seq_lens.max()) if max_len is None else max_len
    return np.arange(max_len)[None, :] < seq_lens[:, None]


def _text(batch):
    batch["txt_ids"] = pad_sequence(batch["txt_ids"], 0)
    batch["txt_masks"] = gen_seq_masks(batch["txt_lens"])
    batch["txt_lens"] = np.asarray(batch["txt_lens"], dtype=np.int64)


def _hist(batch, allow_none):
    if allow_none and max(batch["hist_lens"]) == 0:        # "all are in first step" (r2r_tasks.py:359-365)
        for k in ("hist_img_fts", "hist_ang_fts", "hist_pano_img_fts", "hist_pano_ang_fts"):
            if k in batch:
                batch[k] = None
    else:
        for k in ("hist_img_fts", "hist_ang_fts", "hist_pano_img_fts", "hist_pano_ang_fts"):
            if k in batch:
                batch[k] = pad_tensors(batch[k], lens=batch["hist_lens"], pad=0)


def _hist_tail(batch):
    lens = [x + 1 for x in batch["hist_lens"]]             # "added a special token"
    batch["hist_masks"] = gen_seq_masks(lens)
    batch["hist_lens"] = np.asarray(lens, dtype=np.int64)


def _obs(batch):
    batch["ob_img_fts"] = pad_tensors(batch["ob_img_fts"], lens=batch["ob_lens"], pad=0)
    batch["ob_ang_fts"] = pad_tensors(batch["ob_ang_fts"], lens=batch["ob_lens"], pad=0)
    batch["ob_nav_types"] = pad_sequence(batch["ob_nav_types"], 0)
    batch["ob_masks"] = gen_seq_masks(batch["ob_lens"])
    batch["ob_lens"] = np.asarray(batch["ob_lens"], dtype=np.int64)


def _lists(inputs):
    return {k: [x[k] for x in inputs] for k in inputs[0].keys()}


def mlm_collate(inputs):
    b = _lists(inputs)
    b["txt_labels"] = pad_sequence(b["txt_labels"], -1)
    _text(b); _hist(b, False); _hist_tail(b)
    return b


def mrc_collate(inputs):
    b = _lists(inputs)
    _text(b); _hist(b, False)
    b["hist_mrc_masks"] = pad_sequence(b["hist_mrc_masks"], 0)
    b["hist_img_probs"] = pad_tensors(b["hist_img_probs"], lens=b["hist_lens"], pad=0)
    _hist_tail(b)
    return b


def itm_collate(inputs):
    b = _lists(inputs)
    _text(b); _hist(b, False); _hist_tail(b)
    return b


def sap_collate(inputs):
    b = _lists(inputs)
    _text(b); _obs(b); _hist(b, True); _hist_tail(b)
    b["ob_action_viewindex"] = np.asarray(b["ob_action_viewindex"], dtype=np.int64)
    return b


def sar_collate(inputs):
    b = _lists(inputs)
    _text(b); _obs(b); _hist(b, True); _hist_tail(b)
    b["ob_action_angles"] = np.asarray(b["ob_action_angles"]).astype(np.float32)
    b["ob_progress"] = np.asarray(b["ob_progress"]).astype(np.float32)
    return b


def sprel_collate(inputs):
    b = _lists(inputs)
    _text(b); _obs(b); _hist(b, True); _hist_tail(b)
    b["sp_anchor_idxs"] = np.asarray(b["sp_anchor_idxs"], dtype=np.int64)
    b["sp_targets"] = np.asarray(b["sp_targets"]).astype(np.float32)
    return b


COLLATE = {"mlm": mlm_collate, "mrc": mrc_collate, "itm": itm_collate, "sap": sap_collate, "sar": sar_collate,
           "sprel": sprel_collate}
