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


_roles: dict = {}


def role_stream(device, role: str, priority=None) -> torch.cuda.Stream:
    """THE stream of a role ("side", "capture", "update", "comm", "lane0", ...) on `device`: created once per process and DISTINCT from
    the stream of every other role.  torch.cuda.Stream() hands out a pool of 32 HIP streams per priority round-robin, so a process
    that builds many short-lived owners (a test suite: one capture stream per GraphedTrainStep, one update stream per attached
    optimizer) would sooner or later give two roles of one training step the SAME HIP stream -- e.g. the capture stream and the
    second compute stream -- and the fork / join events between them become self-dependencies inside a capture (seen as a
    segmentation fault in hipGraph instantiation, depending on how many streams the process had created before)."""
    dev = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    s = _roles.get((dev, role))
    if s is None:
        taken = {st.cuda_stream for (d_, _), st in _roles.items() if d_ == dev}
        for _ in range(64):
            s = torch.cuda.Stream(device=dev, priority=int(priority)) if priority is not None else torch.cuda.Stream(device=dev)
            if s.cuda_stream not in taken:
                break
        else:
            raise RuntimeError(f"no free HIP stream for role {role!r} on device {dev}")
        _roles[(dev, role)] = s
    return s


def side_stream(device) -> torch.cuda.Stream:
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    s = _side.get(key)
    if s is None:
        # (default stream priority: a non-default priority on either compute stream stops the two from overlapping inside a graph
        # replay -- 16.4 vs 9.9 ms per step, DESIGN_HISTORY.md)
        s = _side[key] = role_stream(key, "side")
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
        for g in o._gates():
            g.inherit(side, main)


def join(main: torch.cuda.Stream, side: torch.cuda.Stream):
    """`main` continues behind everything enqueued on `side` (main.wait_stream(side))."""
    main.wait_stream(side)
    for o in pending_updates:
        for g in o._gates():
            g.inherit(main, side)


def gate(*what):
    """The caller is about to read these parameters (or the parameters of these modules) directly, i.e. not through a module's
    __call__: order the current stream behind the chunk(s) of an overlapped optimizer update that hold them.  Needed only in
    the forward() of classes that declare `_hamt_container = True` (see optim.AdamW.attach); free when no update is in flight."""
    if not pending_updates:
        return
    for o in list(pending_updates):
        for g in o._gates():
            g.gate(what)
