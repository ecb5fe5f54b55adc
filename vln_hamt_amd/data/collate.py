"""Batch collation with the padding done ON THE DEVICE.

The reference's `*_collate` functions (pretrain_src/data/r2r_tasks.py:95-125, 202-226, 268-288, 343-380, 444-482,
559-596) build every padded tensor on the host (`pad_tensors` / `pad_sequence` / `gen_seq_masks`, data/common.py:5-29:
a zero fill plus one slice copy per sample and field), the DataLoader's pin thread copies all of them again into pinned
memory, and `move_to_cuda` (loader.py:77-87) issues one H2D copy per tensor.  Here the same-named functions return a
`PackedBatch`: every ragged field's rows packed back to back -- no padding, no zero fill -- in ONE byte buffer (one copy
per sample and field, nothing else touches the data on the host; `pin_memory()` makes the DataLoader pin exactly that
buffer), which crosses PCIe in one copy; `to_device` then runs `hamt_unpack_padded` / `hamt_seq_masks` per field and
returns the dict the reference's collate + move_to_cuda would have produced, bit for bit (tests/golden/collate.npz, made
with the reference's own functions).  `out=` lets the kernels write straight into the static input tensors of a
captured step graph.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional

import numpy as np
import torch

from .. import _lib as L

# field -> (length family, pad byte).  pad byte 0xFF on int64 rows = -1 (txt_labels, r2r_tasks.py:111)
RAGGED = {"txt_ids": ("txt", 0x00), "txt_labels": ("txt", 0xFF),
          "hist_img_fts": ("hist", 0), "hist_ang_fts": ("hist", 0), "hist_pano_img_fts": ("hist", 0), "hist_pano_ang_fts": ("hist", 0),
          "hist_img_probs": ("hist", 0), "hist_mrc_masks": ("hist", 0),
          "ob_img_fts": ("ob", 0), "ob_ang_fts": ("ob", 0), "ob_nav_types": ("ob", 0)}
PER_SAMPLE = {"ob_action_viewindex": torch.int64, "sp_anchor_idxs": torch.int64, "ob_action_angles": torch.float32,
              "ob_progress": torch.float32, "sp_targets": torch.float32}
HIST_FIELDS = ("hist_img_fts", "hist_ang_fts", "hist_pano_img_fts", "hist_pano_ang_fts")
_ALIGN = 64


def _rup(n, a=_ALIGN):
    return (n + a - 1) // a * a


class PackedBatch:
    """One byte buffer + the layout to unpack it.  `fields`: name -> (offset, row_shape, dtype, family, pad_byte);
    `per_sample`: name -> (offset, shape, dtype); `lens`: family -> python list; `prefix_off`: family -> offset of the
    int32 prefix table [B + 1]; `lists`: keys the reference's collate leaves as python lists."""

    def __init__(self, task, B, buf, fields, per_sample, lens, prefix_off, lists, hist_none):
        self.task, self.B, self.buf = task, B, buf
        self.fields, self.per_sample, self.lens, self.prefix_off, self.lists, self.hist_none = fields, per_sample, lens, prefix_off, lists, hist_none

    def pin_memory(self):
        """DataLoader(pin_memory=True) calls this on custom batch types (torch/utils/data/_utils/pin_memory.py)."""
        self.buf = self.buf.pin_memory()
        return self

    @property
    def nbytes(self) -> int:
        return self.buf.numel()

    def to_device(self, device, out: Optional[Dict[str, torch.Tensor]] = None, text_pack: bool = False) -> Dict[str, object]:
        """H2D copy of the buffer + unpack kernels on the current stream -> the reference's collated batch on `device`.
        `out`: tensors to fill in place where key, shape and dtype match (static inputs of a captured graph).
        `text_pack`: add the text packing plan of the batch (`txt_pack_idx`, `txt_cu`, `txt_unpack_idx`: synth.text_pack_plan over the
        instruction lengths) -- three small index tensors beyond the reference's keys, with which the model's text-only layers skip
        the padded positions (model.vilmodel.NavPreTrainedModel._text); absent when the batch has (almost) no padding.  An integer
        is the bucket of the packed row count (default 128 rows): a captured step (graph.GraphedTrainStep) is keyed by that count,
        so a coarser bucket trades a little padding for fewer captures."""
        device = torch.device(device)
        if device.type != "cuda":
            raise L.HamtError("PackedBatch.to_device: the collation kernels have no CPU path")
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        lib = L.load()
        dbuf = self.buf.to(device, non_blocking=True)
        st = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
        base = dbuf.data_ptr()
        B = self.B
        res: Dict[str, object] = dict(self.lists)

        def target(name, shape, dtype):
            t = out.get(name) if out is not None else None
            if t is not None and tuple(t.shape) == tuple(shape) and t.dtype == dtype and t.device == device and t.is_contiguous():
                return t
            return torch.empty(shape, dtype=dtype, device=device)

        for name, (off, row_shape, dtype, fam, pad) in self.fields.items():
            if self.hist_none and name in HIST_FIELDS:
                res[name] = None
                continue
            maxlen = max(self.lens[fam])
            t = target(name, (B, maxlen) + tuple(row_shape), dtype)
            row_bytes = int(np.prod(row_shape, dtype=np.int64)) * t.element_size()
            L.check(lib.hamt_unpack_padded(C.c_void_p(base + off), C.c_void_p(base + self.prefix_off[fam]), B, maxlen, row_bytes, pad,
                                           C.c_void_p(t.data_ptr()), st), "hamt_unpack_padded")
            res[name] = t
        for fam, add in (("txt", 0), ("hist", 1), ("ob", 0)):          # hist: "+ 1: added a special token" (r2r_tasks.py:122)
            if fam not in self.lens:
                continue
            maxlen = max(self.lens[fam]) + add
            m = target(f"{fam}_masks", (B, maxlen), torch.bool)
            ln = target(f"{fam}_lens", (B,), torch.int64)
            L.check(lib.hamt_seq_masks(C.c_void_p(base + self.prefix_off[fam]), add, B, maxlen, C.c_void_p(m.data_ptr()),
                                       C.c_void_p(ln.data_ptr()), st), "hamt_seq_masks")
            res[f"{fam}_masks"], res[f"{fam}_lens"] = m, ln
        for name, (off, shape, dtype) in self.per_sample.items():
            n = int(np.prod(shape, dtype=np.int64))
            src = dbuf[off:off + n * torch.empty((), dtype=dtype).element_size()].view(dtype).view(shape)
            t = out.get(name) if out is not None else None
            if t is not None and tuple(t.shape) == tuple(shape) and t.dtype == dtype:
                t.copy_(src)
                res[name] = t
            else:
                res[name] = src
        if text_pack and "txt" in self.lens:
            from ..synth import text_pack_plan
            plan = text_pack_plan(self.lens["txt"], max(self.lens["txt"]), bucket=128 if text_pack is True else int(text_pack))
            if plan is not None:
                for name, t in zip(("txt_pack_idx", "txt_cu", "txt_unpack_idx"), plan):
                    res[name] = t.pin_memory().to(device, non_blocking=True)
        dbuf.record_stream(torch.cuda.current_stream(device))
        return res


def _pack(task: str, inputs: List[dict]) -> PackedBatch:
    B = len(inputs)
    keys = list(inputs[0].keys())
    fams = {"txt": [int(x["txt_lens"]) for x in inputs], "hist": [int(x["hist_lens"]) for x in inputs]}
    if "ob_lens" in keys:
        fams["ob"] = [int(x["ob_lens"]) for x in inputs]
    hist_none = task in ("sap", "sar", "sprel") and max(fams["hist"]) == 0          # "all are in first step"
    # ---- layout
    off = 0
    prefix_off, fields, per_sample, lists = {}, {}, {}, {}
    for fam in fams:
        prefix_off[fam] = off
        off = _rup(off + 4 * (B + 1))
    for k in keys:
        if k in RAGGED:
            fam, pad = RAGGED[k]
            t0 = inputs[0][k]
            rows = sum(fams[fam])
            row_shape = tuple(t0.shape[1:])
            fields[k] = (off, row_shape, t0.dtype, fam, pad)
            off = _rup(off + rows * int(np.prod(row_shape, dtype=np.int64)) * t0.element_size())
        elif k in PER_SAMPLE:
            a0 = np.asarray(inputs[0][k])
            shape = (B,) + a0.shape
            per_sample[k] = (off, shape, PER_SAMPLE[k])
            off = _rup(off + int(np.prod(shape, dtype=np.int64)) * torch.empty((), dtype=PER_SAMPLE[k]).element_size())
        elif k not in ("txt_lens", "hist_lens", "ob_lens"):
            lists[k] = [x[k] for x in inputs]
    buf = torch.empty(max(off, _ALIGN), dtype=torch.uint8)
    # ---- fill: one copy per sample and field, no padding, no zero fill of the payload
    for fam, lens in fams.items():
        pre = np.zeros(B + 1, dtype=np.int32)
        np.cumsum(lens, out=pre[1:])
        buf[prefix_off[fam]:prefix_off[fam] + 4 * (B + 1)].view(torch.int32).copy_(torch.from_numpy(pre))
    for k, (o, row_shape, dtype, fam, _) in fields.items():
        rows = sum(fams[fam])
        n_row = int(np.prod(row_shape, dtype=np.int64))
        esz = torch.empty((), dtype=dtype).element_size()
        view = buf[o:o + rows * n_row * esz].view(dtype).view((rows,) + row_shape)
        r = 0
        for x, l in zip(inputs, fams[fam]):
            if l:
                view[r:r + l].copy_(x[k][:l])
            r += l
    for k, (o, shape, dtype) in per_sample.items():
        esz = torch.empty((), dtype=dtype).element_size()
        vals = torch.as_tensor(np.asarray([np.asarray(x[k]) for x in inputs])).to(dtype)
        buf[o:o + vals.numel() * esz].view(dtype).view(shape).copy_(vals)
    return PackedBatch(task, B, buf, fields, per_sample, fams, prefix_off, lists, hist_none)


def mlm_collate(inputs):
    return _pack("mlm", inputs)


def mrc_collate(inputs):
    return _pack("mrc", inputs)


def itm_collate(inputs):
    return _pack("itm", inputs)


def sap_collate(inputs):
    return _pack("sap", inputs)


def sar_collate(inputs):
    return _pack("sar", inputs)


def sprel_collate(inputs):
    return _pack("sprel", inputs)


COLLATE = {"mlm": mlm_collate, "mrc": mrc_collate, "itm": itm_collate, "sap": sap_collate, "sar": sar_collate, "sprel": sprel_collate}
