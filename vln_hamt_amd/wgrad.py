"""Deferred, grouped weight gradients for the bf16 path.

In the reference every ``nn.Linear`` backward computes dW = dY^T X (and db = column sums of dY) on the spot
(vilmodel.py: all nn.Linear layers, through torch.autograd).  Weight gradients are not on the backward critical
path -- only the optimizer reads them -- and one 768x768 output covers 36 of the 256 CUs, so here the block /
linear backward functions only *queue* (W, b, dY16, X16) and the whole list is handed to ``hamt_wgrad_grouped``
once, when the autograd engine finishes the pass (``queue_callback``): one chip-filling launch per tile class with
full-length K loops, bias sums fused in (one extra MFMA per fragment), no split-K scratch and no reduce pass.

Where the result goes: a parameter that owns a slot in the optimizer's flat gradient arena (``_hamt_grad_slot``,
set by optim.AdamW) gets its gradient written there directly and ``p.grad`` becomes that view, so the optimizer has
nothing to pack; otherwise a fresh fp32 tensor.  A parameter that already has a ``.grad`` when the queue is flushed
(several uses in one pass, the embedding table tied to the MLM decoder, or gradient accumulation over passes) is
accumulated into in place -- the sum autograd's AccumulateGrad would have formed.

Consequence for callers: for queued parameters the Function's backward returns None, i.e. AccumulateGrad hooks
(torch DDP) do not fire for them; multi-GPU training all-reduces the flat arena instead (parallel.allreduce_grads).
``torch.autograd.grad`` w.r.t. such parameters is not supported on the bf16 path (use ``.backward()``).
"""
from __future__ import annotations

import ctypes as C
import os
import threading
import warnings
from typing import List, Optional

import torch

from . import _lib as L

ENABLED = os.environ.get("HAMT_NO_DEFER_WGRAD") is None     # ablation switch: compute every dW immediately
FUSE_SUMSQ = os.environ.get("HAMT_NO_FUSED_SUMSQ") is None  # ablation switch: the gradient norm reads every gradient back
MERGE_PAIRS = os.environ.get("HAMT_WGRAD_NO_PAIRS") is None   # a parameter used twice in a pass: one problem with two operand pairs

stats = {"flushes": 0, "problems": 0, "dropped_stale": 0}


class _Pass:
    """What one backward pass (one autograd graph task) has queued so far."""
    __slots__ = ("items", "vecs", "lnred", "targets", "seen", "launched", "cleared", "fused_ids")

    def __init__(self):
        self.items: List[tuple] = []     # (w, b, dy16, x16)
        self.vecs: List[tuple] = []      # (parameter, gradient tensor): published as .grad at flush
        self.lnred: List[tuple] = []     # (ws, red, M, H, want_dxsum): LayerNorm-backward partials, one grouped launch at flush
        # state of the pass's launch
        self.targets: dict = {}          # id(parameter) -> gradient buffer
        self.seen: dict = {}             # id(buffer) -> writes launched so far
        self.launched = 0                # launches so far
        self.cleared = False             # the optimizers' tile sums of an EARLIER pass have been dropped
        self.fused_ids = None            # buffers whose tile sums this pass's launches handed to the optimizer


class WgradQueue:
    """The deferred work of the backward passes of ONE device (one process per GPU: one queue per process in practice;
    the autograd engine runs a device's backward nodes on that device's worker thread and the end-of-pass callbacks on
    the thread that called .backward(), so the queue cannot be thread-local).  Work is keyed by the autograd graph-task
    id of the pass that queued it, so a pass that died mid-way (an exception in some backward node: its end-of-pass callback never runs)
    cannot leak its operands into the next pass or switch the next pass's flush off: the next pass to queue something
    finds the orphan, drops it (counted in stats["dropped_stale"]) and warns.  `handler` (optional) consumes the item
    list of a pass instead of the default launch (parallel.OverlappedGradSync)."""

    def __init__(self):
        self.passes: dict = {}           # graph-task id -> _Pass
        self.handler = None
        self.dropped: set = set()        # ids whose work was dropped as stale (a nested outer pass would find itself here)

    def current(self) -> _Pass:
        tid = torch._C._current_graph_task_id()
        if tid < 0:
            raise L.HamtError("wgrad: weight gradients can only be deferred from inside a backward pass")
        ps = self.passes.get(tid)
        if ps is None:
            for old in list(self.passes):            # orphans of passes that never reached their end-of-pass callback
                n = len(self.passes[old].items) + len(self.passes[old].lnred)
                del self.passes[old]
                self.dropped.add(old)
                stats["dropped_stale"] += n
                warnings.warn(f"hamt wgrad: dropped {n} queued weight-gradient problems of an earlier backward pass that did "
                              "not finish (exception inside backward?)", RuntimeWarning)
            ps = self.passes[tid] = _Pass()
            torch.autograd.Variable._execution_engine.queue_callback(lambda: self.flush(tid))
            from . import streams
            streams.wait_pending_updates()       # (a backward pass not preceded by a forward pass of the attached model)
        return ps

    def pending(self) -> int:
        return sum(len(p.items) for p in self.passes.values())

    def reset(self):
        """Drop queued work (after an exception inside a backward pass left the queue behind)."""
        self.passes.clear()

    @torch.no_grad()
    def flush(self, tid=None):
        """Launch everything pass `tid` queued and publish the results as ``.grad``.  Runs as the autograd engine's
        end-of-pass callback (on the caller's current stream); harmless when nothing is queued."""
        if tid is None:                              # explicit call: whatever is queued, oldest pass first
            for t in sorted(self.passes):
                self.flush(t)
            return
        if tid in self.dropped:
            self.dropped.discard(tid)
            raise L.HamtError("wgrad: the deferred weight gradients of this backward pass were dropped as stale by a backward "
                              "pass nested inside it; nested backward passes are not supported on the bf16 path "
                              "(HAMT_NO_DEFER_WGRAD=1 computes every weight gradient in line)")
        ps = self.passes.pop(tid, None)
        if ps is None:
            return
        _flush_pass(ps, self.handler)


_queues: dict = {}
_queues_lock = threading.Lock()


def queue(device=None) -> WgradQueue:
    """the queue of `device` (default: the current CUDA device; inside a backward node that is the node's device)"""
    idx = torch.device(device).index if device is not None else None
    if idx is None:
        idx = torch.cuda.current_device() if torch.cuda.is_available() else 0
    q = _queues.get(idx)
    if q is None:
        with _queues_lock:
            q = _queues.setdefault(idx, WgradQueue())
    return q


def set_handler(fn, device=None):
    """Route the end-of-pass item list [(w, b, dy16, x16), ...] of the passes on `device` to `fn` instead of launching
    it here (None: default)."""
    queue(device).handler = fn


def get_handler(device=None):
    return queue(device).handler


def eligible(w: torch.Tensor, dy16: torch.Tensor, x16: torch.Tensor) -> bool:
    """Can dW = dy16^T x16 for parameter `w` go through the grouped kernel?"""
    if not ENABLED:
        return False
    if w.dim() != 2 or w.dtype != torch.float32 or not w.is_leaf or not w.requires_grad:
        return False
    if dy16.dtype != torch.bfloat16 or x16.dtype != torch.bfloat16 or dy16.dim() != 2 or x16.dim() != 2:
        return False
    K = dy16.shape[0]
    if K != x16.shape[0] or K % 64 or dy16.shape[1] != w.shape[0] or x16.shape[1] != w.shape[1]:
        return False
    if dy16.stride(1) != 1 or x16.stride(1) != 1:
        return False
    ldy, ldx = dy16.stride(0), x16.stride(0)
    if ldy % 8 or ldx % 8 or ldy < 64 or ldx < 128 or dy16.data_ptr() % 16 or x16.data_ptr() % 16:
        return False
    return True


def defer(w: torch.Tensor, b: Optional[torch.Tensor], dy16: torch.Tensor, x16: torch.Tensor, rows: Optional[int] = None):
    """Queue dW (+ db when `b` is a parameter that needs a gradient).  Must be called from inside a backward pass.
    `rows`: the valid reduction rows of dy16 / x16 (hamt_wgrad_desc.K_valid); default: all of them count (padding rows zero)."""
    if b is not None and not b.requires_grad:
        b = None
    if rows is not None:
        dy16._hamt_rows = int(rows)          # rides on the operand: the queue entries stay (w, b, dy16, x16)
    dy16._hamt_stream = torch.cuda.current_stream().cuda_stream      # the stream whose kernels produced the operands
    queue(dy16.device).current().items.append((w, b, dy16, x16))


def valid_rows(dy16: torch.Tensor) -> int:
    """hamt_wgrad_desc.K_valid of a queued operand (0: every row counts)"""
    r = int(getattr(dy16, "_hamt_rows", 0))
    return r if 0 < r < dy16.shape[0] else 0


def defer_ln_reduce(ws, red, M, H, want_dxsum, pairs):
    """LayerNorm backward left its per-block partials in `ws` (ops._ln_bwd): sum them into `red` ([3, H]: dgamma, dbeta,
    column sums of dx) in the grouped launch at the end of the pass, then publish `pairs` = [(parameter, row of red)] as
    `.grad` (added to an existing one).  The Function's backward returns None for these parameters."""
    ps = queue(ws.device).current()
    ps.lnred.append((ws, red, M, H, want_dxsum, [(p if (p is not None and t is not None and p.requires_grad) else None) for p, t in pairs]))


def mark_arena_dirty(slot: torch.Tensor):
    """A producer relied on `slot` (a view of the optimizer's flat gradient arena) being zero before it added into it: the arena
    must be re-zeroed before the next pass even if no optimizer step runs in between (optim.AdamW.zero_grad checks the mark;
    the update kernel zeroes the arena itself)."""
    base = slot._base if slot._base is not None else slot
    base._hamt_dirty = True


def publish_slot_grad(p: torch.Tensor, slot: torch.Tensor):
    """`slot` (p's gradient-arena slot) was accumulated into in place during this pass: make it `p.grad` at the end of the pass."""
    mark_arena_dirty(slot)
    queue(slot.device).current().vecs.append((p, slot))


def table_entries(descs, n: int) -> int:
    """upper bound on the launch-table entries of hamt_wgrad_grouped for these problems (include/hamt.h)"""
    return L.workspace_bytes(L.WS_WGRAD_TABLE, *[descs[i].M for i in range(n)]) // L.WGRAD_TABLE_ENTRY


def launch(descs, n: int, table: Optional[torch.Tensor] = None, phase: int = 3):
    """hamt_wgrad_grouped on the current stream; `table` = device scratch for the launch table (allocated here when
    None: from the caching allocator, i.e. from the graph's private pool during a capture).  phase 1 / 2: only write the table /
    only launch from a table written before (hamt_wgrad_grouped_ex)."""
    from .ops import _stream
    if n == 0:
        return
    if table is None:
        assert phase == 3
        table = torch.empty(table_entries(descs, n) * L.WGRAD_TABLE_ENTRY, dtype=torch.uint8, device="cuda")
    L.check(L.load().hamt_wgrad_grouped_ex(n, descs, table.data_ptr(), table.numel(), phase, _stream()), "hamt_wgrad_grouped")


def pending(device=None) -> int:
    return queue(device).pending()


def reset(device=None):
    """Drop the device's queued work (after an exception inside a backward pass left the queue behind)."""
    queue(device).reset()


def flush(device=None):
    """Launch whatever is queued for the device (normally the end-of-pass callback does; see WgradQueue.flush)."""
    queue(device).flush()


def _target(p: torch.Tensor, targets: dict, fresh: list):
    t = targets.get(id(p))
    if t is not None:
        return t, 1
    g = p.grad
    if g is not None and g.dtype == torch.float32 and g.is_contiguous() and g.shape == p.shape:
        targets[id(p)] = g
        return g, 1
    slot = getattr(p, "_hamt_grad_slot", None)
    if slot is not None and slot.shape == p.shape and slot.device == p.device:
        t = slot
    else:
        t = torch.empty(p.shape, dtype=torch.float32, device=p.device)
    targets[id(p)] = t
    fresh.append((p, t))
    return t, 0


@torch.no_grad()
def _flush_pass(ps: _Pass, handler):
    from .ops import _stream
    items, vecs, lnred = ps.items, ps.vecs, ps.lnred
    if not items and not lnred and not vecs:
        return
    from . import streams
    streams.wait_pending_updates()
    streams.join_all()                # operands queued by backward nodes that ran on the second compute stream
    if lnred:                         # every LayerNorm's dgamma / dbeta / bias-gradient partials: one launch
        n = len(lnred)
        descs = (L.LnReduceDesc * n)()
        cur = torch.cuda.current_stream()
        late = []
        uses: dict = {}
        for (_ws, _red, _M, _H, want_dxsum, params) in lnred:
            for j, p in enumerate(params):
                if p is not None and not (j == 2 and not want_dxsum):
                    uses[id(p)] = uses.get(id(p), 0) + 1
        for i, (ws, red, M, H, want_dxsum, params) in enumerate(lnred):
            ws.record_stream(cur)         # (allocated under whichever compute stream ran that LayerNorm backward)
            red.record_stream(cur)
            d = descs[i]
            d.ws, d.M, d.H, d.atomic = ws.data_ptr(), M, H, 0
            ptrs = [None, None, None]
            for j, p in enumerate(params):
                if p is None or (j == 2 and not want_dxsum):
                    continue
                slot = getattr(p, "_hamt_grad_slot", None)
                # (at most two contributions per slot: a two-term fp32 sum does not depend on the order the atomics land in;
                # a parameter used more often -- every step of a finetune rollout -- keeps the ordered sum below)
                if (slot is not None and slot.numel() == H and uses[id(p)] <= 2 and getattr(p, "_hamt_slot_zeroed", False)
                        and (p.grad is None or p.grad.data_ptr() == slot.data_ptr())):
                    # straight into the parameter's (zero-initialised) gradient-arena slot: a parameter shared by several LayerNorm
                    # calls (the cross-attention block runs twice per x-layer) needs no separate accumulation, and nothing to pack
                    ptrs[j] = slot.data_ptr()
                    d.atomic |= 1 << j
                    p.grad = slot
                    mark_arena_dirty(slot)
                else:
                    ptrs[j] = red[j].data_ptr()
                    late.append((p, red[j]))
            d.dgamma, d.dbeta, d.dxsum = ptrs
        table = torch.empty(n * L.LNRED_TABLE_ENTRY, dtype=torch.uint8, device=lnred[0][0].device)
        L.check(L.load().hamt_ln_bwd_reduce_grouped(n, descs, table.data_ptr(), table.numel(), _stream()), "hamt_ln_bwd_reduce_grouped")
        for p, t in late:
            p.grad = t if p.grad is None else p.grad + t
    for p, t in vecs:                    # gradients produced in place in their arena slots during the pass (embedding tables)
        if p.grad is None:
            p.grad = t
        elif p.grad.data_ptr() != t.data_ptr():
            p.grad = p.grad + t
    if not items:
        return
    if handler is not None:
        stats["flushes"] += 1
        stats["problems"] += len(items)
        handler(items)
        return
    _launch_items(ps, items)


def ps_fused_ids(ps: _Pass) -> set:
    """ids of the gradient buffers whose tile sums of squares earlier launches of this pass handed to the optimizer"""
    if ps.fused_ids is None:
        ps.fused_ids = set()
    return ps.fused_ids


@torch.no_grad()
def _launch_items(ps: _Pass, items, later=frozenset()):
    """hamt_wgrad_grouped for `items` on the current stream and publication of the results as `.grad`.  `ps.targets` / `ps.seen` carry which buffers
    a launch of the pass wrote -- a later write to the same buffer accumulates.  `later`: ids of parameters that still have queued problems (their gradients are not final after
    this launch: no tile sums of squares for them)."""
    targets, seen = ps.targets, ps.seen
    fresh: list = []
    refused = False
    # Two problems that write the same buffer (a parameter used twice in the pass) must not share a launch: the k-th
    # write to a buffer goes into the k-th launch group, and groups run in stream order.
    base_seen = dict(seen)
    groups: List[list] = []
    first_entry: dict = {}
    for (w, b, dy16, x16) in items:
        tw, aw = _target(w, targets, fresh)
        tb, ab = (None, 0)
        if b is not None:
            tb, ab = _target(b, targets, fresh)
        k = seen.get(id(tw), 0)
        if tb is not None:
            k = max(k, seen.get(id(tb), 0))
        seen[id(tw)] = k + 1
        if tb is not None:
            seen[id(tb)] = k + 1
        bk = max(base_seen.get(id(tw), 0), base_seen.get(id(tb), 0) if tb is not None else 0)
        g = k - bk                               # writes to these buffers by earlier launches are complete (same stream): only this call's count
        if bk > 0 and id(tw) in ps_fused_ids(ps):
            refused = True                       # a weight an earlier launch of this pass left tile sums for is written again
        # the SECOND use of a parameter in this pass (the cross-attention weights an x-layer shares between its two directions): a second
        # operand pair of the first use's problem (hamt_wgrad_desc.dy2) instead of an accumulating launch of its own behind the first
        e0 = first_entry.get(id(tw)) if (MERGE_PAIRS and g == 1) else None
        if (e0 is not None and e0[7] is None and e0[5] is tb and valid_rows(e0[1]) == 0 and dy16.stride(0) % 8 == 0 and x16.stride(0) % 8 == 0
                and dy16.stride(0) >= 64 and x16.stride(0) >= 128):
            e0[7], e0[8] = dy16, x16
            seen[id(tw)] = k                     # (still ONE problem writing this buffer)
            if tb is not None:
                seen[id(tb)] = k
            continue
        while len(groups) <= g:
            groups.append([])
        entry = [w, dy16, x16, tw, aw or k > 0, tb, ab or k > 0, None, None]
        groups[g].append(entry)
        if g == 0 and bk == 0:
            first_entry[id(tw)] = entry
    lib = L.load()
    # Sum of squares of the gradients while their tiles are still in registers (hamt_wgrad_desc.ss): for every weight whose
    # gradient is STORED once in this pass straight into its gradient-arena slot -- the bulk of the parameters -- so that the
    # global-norm clip does not have to read them back (optim.AdamW.global_grad_sumsq adds the slots' total to the table norm).
    fused, ss_all, opt = [], None, None
    if not ps.cleared:
        ps.cleared = True
        for o in {id(r): r for r in (getattr(it[0], "_hamt_opt", None) for it in items) if r is not None}.values():
            if o() is not None:
                o().clear_fused_sumsq()            # whatever an earlier pass left (gradient accumulation: this pass adds on top)
    if refused:      # (safe fallback: the norm kernel reads those gradients back from memory)
        for o in {id(r): r for r in (getattr(it[0], "_hamt_opt", None) for it in items) if r is not None}.values():
            if o() is not None:
                o().clear_fused_sumsq()
        ps.fused_ids = set()
    if FUSE_SUMSQ and groups and not refused:
        slots_of = lambda w: ((w.shape[0] + 63) // 64) * ((w.shape[1] + 127) // 128)
        n_ss = 0
        for (w, dy16, x16, tw, aw, tb, ab, _dy2, _x2) in groups[0]:
            o = getattr(w, "_hamt_opt", None)
            o = o() if o is not None else None
            slot = getattr(w, "_hamt_grad_slot", None)
            if (o is not None and (opt is None or o is opt) and not aw and seen.get(id(tw), 0) == 1 and id(w) not in later and slot is not None
                    and tw.data_ptr() == slot.data_ptr() and not getattr(w, "_hamt_slot_zeroed", True) and tw.stride(0) == w.shape[1]):
                opt = o
                fused.append((id(tw), w, n_ss))
                n_ss += slots_of(w)
        if fused:
            ss_all = torch.zeros(n_ss, dtype=torch.float32, device=groups[0][0][3].device)
    ss_of = {k: off for (k, _w, off) in fused}
    for gi, grp in enumerate(groups):
        descs = (L.WgradDesc * len(grp))()
        for i, (w, dy16, x16, tw, aw, tb, ab, dy2, x2) in enumerate(grp):
            d = descs[i]
            d.dy, d.x, d.dw, d.db = dy16.data_ptr(), x16.data_ptr(), tw.data_ptr(), (tb.data_ptr() if tb is not None else None)
            d.M, d.N, d.K = w.shape[0], w.shape[1], dy16.shape[0]
            d.ldy, d.ldx, d.ldw = dy16.stride(0), x16.stride(0), tw.stride(0)
            d.accum_dw, d.accum_db = int(bool(aw)), int(bool(ab))
            d.K_valid = valid_rows(dy16)
            d.ss = (ss_all.data_ptr() + 4 * ss_of[id(tw)]) if (gi == 0 and id(tw) in ss_of) else None
            if dy2 is not None:
                d.dy2, d.x2, d.K2, d.ldy2, d.ldx2, d.K2_valid = dy2.data_ptr(), x2.data_ptr(), dy2.shape[0], dy2.stride(0), x2.stride(0), valid_rows(dy2)
        launch(descs, len(grp))
    if fused:
        opt.note_fused_sumsq(ss_all, [w for (_k, w, _o) in fused], append=ps.launched > 0 and bool(ps_fused_ids(ps)))
        ps_fused_ids(ps).update(k for (k, _w, _o) in fused)
    ps.launched += 1
    stats["flushes"] += 1
    stats["problems"] += len(items)
    for p, t in fresh:
        if p.grad is None:
            p.grad = t
        else:                                   # an existing gradient of another dtype / layout
            p.grad.add_(t.to(p.grad.dtype))


# ------------------------------------------------------------------------------------------ planned (arena) mode
class Plan:
    """The queued weight gradients of one backward pass, resolved against the optimizer's flat gradient arena and cut
    into `n_groups` launch groups in arena order, so that a caller can all-reduce the arena range of group g while
    group g+1 is still computing (parallel.OverlappedGradSync).  `ranges` = [(lo, hi, after_group, groups)]: arena
    elements [lo, hi) are final once every launch group in `groups` has run (`after_group` = the last of them, -1 /
    empty: final before any group; groups may run concurrently on two streams, so "the last one" alone is not enough)."""

    def __init__(self):
        self.groups: List[tuple] = []      # (ctypes desc array, count)
        self.tables: List[torch.Tensor] = []   # per-group device scratch for the launch tables (re-used every replay)
        self.ranges: List[tuple] = []
        self.deps: List[set] = []          # deps[g]: earlier groups that wrote a buffer group g accumulates into
        self.keep: list = []               # operand tensors: alive as long as the plan (graph replays re-use their memory)
        self.pack_segs: Optional[list] = None   # wire mode: [(offset, count)] arena segments whose gradients are still fp32 in the
        #                                         gradient arena when the groups have run (everything else active in the wire
        #                                         region was written to the bf16 staging arena by the launches themselves)
        self.graphs: Optional[list] = None      # per group: a captured hipGraph of its launches (graph.GraphedTrainStep)


def build_plan(items, optimizer, n_groups: int = 4, wire=None, cuts=None) -> Optional[Plan]:
    """Resolve `items` to arena slots (every parameter must own one) and publish `.grad` views.  A parameter that
    already has an autograd-produced `.grad` gets accum = 1: the caller packs that gradient into the slot BEFORE the
    groups run.  Returns None when some parameter has no arena slot (caller falls back to flush semantics).

    `wire` = (stage16, scale, n_wire): a data-parallel exchange with bf16 on the wire (parallel.ShardedGradSync).  A weight gradient
    that is STORED once in this pass into an arena slot below element `n_wire` (the GEMM-weight region) is then written by its
    launch as bf16(scale * dW) straight into `stage16` (the exchange's staging arena, same element offsets) -- what the pack pass
    over the fp32 arena would have produced -- and the fp32 slot is not written at all.  `plan.pack_segs` lists what is left for
    the pack kernel in that region: weights written twice in the pass (shared cross-attention weights: fp32 accumulation first),
    and gradients that came through autograd.

    `cuts` (sorted arena offsets, the interior boundaries of the exchange's STATIC ranges in the GEMM-weight region): launch group g =
    the queued weights whose slot starts in static range g, instead of `n_groups` equal-work groups.  A static range is then final
    as soon as ITS group has run -- its collective (and the widening of the owned chunk) starts under the next group -- where
    equal-work cuts, which fall anywhere, left three of four ranges waiting for the last group (round 6: 350 us of serial
    unpack / norm kernels behind the last weight-gradient launch of a one-rank exchange)."""
    return _build_plan(items, optimizer, n_groups, wire, cuts)


def _merge_segs(segs):
    out = []
    for o, c in sorted(segs):
        if out and out[-1][0] + out[-1][1] >= o:
            out[-1] = (out[-1][0], max(out[-1][1], o + c - out[-1][0]))
        else:
            out.append((o, c))
    return out


def _build_plan(items, optimizer, n_groups, wire, cuts=None):
    flat_g = optimizer._flat_g
    base, n_total = flat_g.data_ptr(), flat_g.numel()

    def off(p):
        slot = getattr(p, "_hamt_grad_slot", None)
        if slot is None or not (base <= slot.data_ptr() < base + 4 * n_total):
            return None
        return (slot.data_ptr() - base) // 4

    probs = []
    seen: dict = {}
    for (w, b, dy16, x16) in items:
        ow = off(w)
        ob = off(b) if b is not None else None
        if ow is None or (b is not None and ob is None):
            return None
        probs.append((ow, ob, w, b, dy16, x16))
    probs.sort(key=lambda t: t[0])
    if MERGE_PAIRS:       # a parameter used twice in the pass: one problem with two operand pairs (as _launch_items does)
        merged_probs = []
        for t in probs:
            ow, ob, w, b, dy16, x16 = t
            prev = merged_probs[-1] if merged_probs else None
            if (prev is not None and prev[0] == ow and prev[1] == ob and len(prev) == 6 and valid_rows(prev[4]) == 0 and w.grad is None
                    and (b is None or b.grad is None)):
                merged_probs[-1] = prev + (dy16, x16)
            else:
                merged_probs.append(t)
        probs = merged_probs
    # cut weights: the parameters' sizes, NOT the operands' row counts -- those differ between the ranks of a data-parallel job
    # (per-batch padding to the local longest instruction, the packed text's row bucket), and the launch groups decide when an
    # arena range may be exchanged: the cuts must be the same on every rank that queued the same parameters
    flops = [float(t[2].numel()) for t in probs]
    total, acc, gi = sum(flops), 0.0, 0
    group_of = []
    if cuts:                                # one group per static exchange range (arena layout and world size only: rank-invariant too)
        import bisect
        first = bisect.bisect_right(cuts, probs[0][0]) if probs else 0
        for t in probs:                     # (groups without a queued weight are dropped: numbering starts at the first one used)
            group_of.append(bisect.bisect_right(cuts, t[0]))
        used = sorted(set(group_of))
        renum = {g: i for i, g in enumerate(used)}
        group_of = [renum[g] for g in group_of]
    else:
        for f in flops:                     # equal-work cuts in arena order
            if gi < n_groups - 1 and acc >= (gi + 1) * total / n_groups:
                gi += 1
            group_of.append(gi)
            acc += f
    # a buffer written twice in the pass: the second write must land in a LATER launch than the first
    last_group: dict = {}
    dep_pairs = set()
    for i, (ow, ob, w, b, dy16, x16, *_pair) in enumerate(probs):
        prev = [last_group[o] for o in (ow, ob) if o is not None and o in last_group]
        gmin = max(prev, default=-1) + 1
        group_of[i] = max(group_of[i], gmin)
        dep_pairs.update((group_of[i], g) for g in prev)
        last_group[ow] = group_of[i]
        if ob is not None:
            last_group[ob] = group_of[i]
    ng = max(group_of) + 1
    plan = Plan()
    plan.deps = [{d for (g, d) in dep_pairs if g == gg} for gg in range(ng)]
    written: dict = {}
    per_group: List[list] = [[] for _ in range(ng)]
    n_writes: dict = {}
    for (ow, *_r) in probs:
        n_writes[ow] = n_writes.get(ow, 0) + 1
    direct = set()                         # arena offsets of the weights whose launch writes the bf16 wire value itself
    for i, (ow, ob, w, b, dy16, x16, *pair) in enumerate(probs):
        # (a gradient that exists already -- an autograd tensor the caller packs into the slot before the groups run, or the slot
        # itself holding what was added in place during the pass: an embedding table tied to this weight -- is accumulated onto)
        aw = 1 if (ow in written or w.grad is not None) else 0
        ab = 0
        if ob is not None:
            ab = 1 if (ob in written or b.grad is not None) else 0
            written[ob] = True
        written[ow] = True
        if wire is not None and not aw and n_writes[ow] == 1 and not pair and ow + w.numel() <= wire[2] and w.shape[1] % 8 == 0:
            direct.add(ow)           # (two operand pairs: fp32 slot + the pack pass, as for every weight written twice before)
        per_group[group_of[i]].append((ow, ob, w, b, dy16, x16, aw, ab, pair))
        plan.keep += [dy16, x16] + list(pair)
    for grp in per_group:
        descs = (L.WgradDesc * max(1, len(grp)))()
        for i, (ow, ob, w, b, dy16, x16, aw, ab, pair) in enumerate(grp):
            d = descs[i]
            d.dy, d.x, d.dw, d.db = dy16.data_ptr(), x16.data_ptr(), base + 4 * ow, (base + 4 * ob if ob is not None else None)
            d.M, d.N, d.K = w.shape[0], w.shape[1], dy16.shape[0]
            d.ldy, d.ldx, d.ldw = dy16.stride(0), x16.stride(0), w.shape[1]
            d.accum_dw, d.accum_db = aw, ab
            d.K_valid = valid_rows(dy16)
            if pair:
                d.dy2, d.x2, d.K2, d.ldy2, d.ldx2, d.K2_valid = pair[0].data_ptr(), pair[1].data_ptr(), pair[0].shape[0], pair[0].stride(0), pair[1].stride(0), valid_rows(pair[0])
            if ow in direct:
                d.dw, d.wire_scale = wire[0].data_ptr() + 2 * ow, float(wire[1])
        plan.groups.append((descs, len(grp)))
        plan.tables.append(torch.empty(max(1, table_entries(descs, len(grp))) * L.WGRAD_TABLE_ENTRY, dtype=torch.uint8, device=flat_g.device))
    # arena ranges and the launch group after which each is final
    lows = [min(t[0] for t in grp) for grp in per_group if grp]
    bounds = [0] + lows[1:] + [n_total]
    for r in range(len(bounds) - 1):
        lo, hi = bounds[r], bounds[r + 1]
        after, touched = -1, set()
        for g, grp in enumerate(per_group):
            for (ow, ob, w, b, *_rest) in grp:
                if lo <= ow < hi or (ob is not None and lo <= ob < hi):
                    after = max(after, g)
                    touched.add(g)
        if hi > lo:
            plan.ranges.append((lo, hi, after, frozenset(touched)))
    for (ow, ob, w, b, *_r) in probs:       # publish .grad (arena views) so that "has a gradient" == "is active"
        if w.grad is None:
            w.grad = w._hamt_grad_slot
        if b is not None and b.grad is None:
            b.grad = b._hamt_grad_slot
    if wire is not None:
        # what the pack kernel still has to convert in the wire region: every parameter there that has a gradient now (queued
        # weights just got theirs published; the rest came through autograd) and was not written directly
        segs = []
        for p, o in zip(optimizer._params, optimizer._offs):
            if o >= wire[2]:
                break
            if p.grad is not None and o not in direct:
                segs.append((o, (p.numel() + 7) // 8 * 8))
        plan.pack_segs = _merge_segs(segs)
        plan.keep.append(wire[0])
    return plan


def launch_group(plan: Plan, g: int, phase: int = 3):
    descs, n = plan.groups[g]
    if n:
        launch(descs, n, table=plan.tables[g], phase=phase)
