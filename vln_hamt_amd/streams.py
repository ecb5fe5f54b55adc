"""Second compute stream for independent sub-graphs of the model (the vision side of the cross-modal layers).

Rules kept by the callers (model/vilmodel.py): every side region starts with ``side.wait_stream(main)`` and ends with
``main.wait_stream(side)``; every tensor that crosses is registered with ``share`` (``record_stream`` on it and on its
bf16 image), so the caching allocator never hands its memory to the other stream's later allocations while kernels of
this stream may still read it.  The deferred weight-gradient launch (wgrad.py) joins the side streams before it reads
operands their backward kernels produced.  HAMT_NO_XSTREAM=1 disables the second stream (ablation / debugging)."""
from __future__ import annotations

import os

import torch

_ENABLED = [os.environ.get("HAMT_NO_XSTREAM") is None]
# which of the two regions use the second stream: "all" (default), "xlayers" (the vision side of the cross-modal layers only),
# "trunk" (the history / observation embedders incl. the panorama encoder next to the text layers only) -- measurement switch
_MODE = os.environ.get("HAMT_XSTREAM_MODE", "all")
_side: dict = {}


def two_stream_enabled(where: str = "xlayers") -> bool:
    return _ENABLED[0] and _MODE in ("all", where)


def set_two_stream(flag: bool):
    _ENABLED[0] = bool(flag)


def side_stream(device) -> torch.cuda.Stream:
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    s = _side.get(key)
    if s is None:
        # HAMT_SIDE_PRIORITY = -1 / 0: measurement switch (default stream priority otherwise)
        pr = os.environ.get("HAMT_SIDE_PRIORITY")
        s = _side[key] = torch.cuda.Stream(device=key, priority=int(pr)) if pr is not None else torch.cuda.Stream(device=key)
    return s


def share(t, stream):
    """`t` (and its bf16 image, if it carries one) will be read by kernels on `stream`."""
    if t is None or not torch.is_tensor(t) or not t.is_cuda:
        return
    t.record_stream(stream)
    img = getattr(t, "_hamt_bf16", None)
    if img is not None and torch.is_tensor(img[0]):
        img[0].record_stream(stream)


def join_all():
    """Make the current stream wait for everything enqueued on the side streams (cheap when they are idle)."""
    cur = torch.cuda.current_stream()
    for s in _side.values():
        if s.device == cur.device:
            cur.wait_stream(s)


pending_updates: set = set()     # optimizers (optim.AdamW.attach) whose update may still be running on their update stream


def wait_pending_updates():
    """Order the current stream behind every overlapped optimizer update still in flight (cheap when there is none).  Called by
    whatever writes into the gradient arena outside a module's forward pass (wgrad.py, the embedding-table scatter-add)."""
    for o in list(pending_updates):
        o.wait_update()


def fork(main: torch.cuda.Stream, side: torch.cuda.Stream):
    """`side` continues from where `main` is (side.wait_stream(main)); what `main` already waits for of an overlapped optimizer
    update, `side` now waits for as well."""
    side.wait_stream(main)
    for o in pending_updates:
        o._ov.inherit(side, main)


def join(main: torch.cuda.Stream, side: torch.cuda.Stream):
    """`main` continues behind everything enqueued on `side` (main.wait_stream(side))."""
    main.wait_stream(side)
    for o in pending_updates:
        o._ov.inherit(main, side)


def gate(*what):
    """The caller is about to read these parameters (or the parameters of these modules) directly, i.e. not through a module's
    __call__: order the current stream behind the chunk(s) of an overlapped optimizer update that hold them.  Needed only in
    the forward() of classes that declare `_hamt_container = True` (see optim.AdamW.attach); free when no update is in flight."""
    if not pending_updates:
        return
    for o in list(pending_updates):
        o._ov.gate(what)
