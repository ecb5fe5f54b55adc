"""HF-style AdamW (pretrain_src/optim/adamw.py:13-112) as HIP kernels over flat fp32 arenas.

On the first step (or `materialize()`) every parameter is re-homed into ONE contiguous fp32 arena
(``p.data`` becomes a view; names, shapes and values are unchanged), with matching flat gradient /
exp_avg / exp_avg_sq arenas and a bf16 shadow arena that the update kernel refreshes in the same pass
(the MFMA GEMMs read their weight operands from it, so no per-step cast of the weights is needed).

``step()`` = three launches, independent of the number of parameters:
  1. gradients are packed into the arena (one fused multi-tensor copy) -- except the weight / bias gradients of the
     bf16 GEMM path, which the grouped weight-gradient launch (wgrad.py) already wrote into their arena slots;
  2. ``clip_grad_norm_`` reduces the global L2 norm on device and defers the scaling into the update;
  3. ``hamt_adamw_table``: per-parameter {lr, bias-corrected step size, weight decay, active} come from a small
     device table refreshed by the host each step.  Parameters whose ``grad is None`` are skipped like the
     reference's ``continue`` (:70-71) and keep their own step count for the bias correction (:93-97).
The launch sequence is static, so a whole training step can be captured in a hipGraph (`prepare_step` does the
host part before a replay, `launch_step` is what gets captured).
"""
from __future__ import annotations

import ctypes as C
import math
import os
import weakref

import numpy as np
from typing import Iterable, List, Optional

import torch
from torch.optim import Optimizer

from .. import _lib as L
from .. import streams
from ..ops import _p, _stream

KEEP_GRAD = os.environ.get("HAMT_ZERO_ALL_GRADS") is None      # ablation: zero every gradient slot in the update (34 B / parameter)
ALIGN = 8   # elements: keeps every tensor 32-byte aligned in the fp32 arenas and 16-byte aligned in the bf16 shadow
REGION_ALIGN = 512   # elements: region / arena ends (8-element granules x up to 64 ranks)


def shadow_only(p) -> bool:
    """Is `p` read by the model ONLY through the optimizer's bf16 shadow (a GEMM weight of the bf16 path)?  Everything else --
    vectors, embedding tables (gathered in fp32; `_hamt_fp32_read`, set by modeling.HamtPreTrainedModel), skinny or odd-width
    matrices that go through the exact-fp32 GEMM -- is read from the fp32 master."""
    return p.dim() == 2 and p.shape[1] % 64 == 0 and p.shape[0] % 8 == 0 and p.shape[0] >= 64 and not getattr(p, "_hamt_fp32_read", False)


class _Overlap:
    """State of AdamW.attach(): the update stream, the chunk list with one event per chunk, the module hooks."""

    def __init__(self, opt, model, chunk_elems, chunks=None, stream=None):
        self.opt, self.model = opt, model
        from .. import streams
        self.stream = stream if stream is not None else streams.role_stream(opt._flat_p.device, "update")      # (one stream per role and process: streams.role_stream)
        # chunk 0: the fp32-read region (biases, LayerNorm, embedding tables: the first kernels of a forward pass read those);
        # then the GEMM-weight region in arena (= registration = use) order, cut at parameter boundaries
        # (`chunks` given -- parallel.ShardedGradSync.attach_gather: the static exchange ranges in the order their all-gathers are
        # issued on `stream`; they cut THROUGH parameters, so a parameter belongs to the last chunk that holds any of it)
        n_a, n = opt._n_shadow_only, opt._n
        if chunks is None:
            ends = [int(e) for e in opt._ends.tolist()]
            chunks = [(n_a, n - n_a)] if n > n_a else []
            lo = 0
            for e in ends:
                if e > n_a:
                    break
                if e - lo >= chunk_elems:
                    chunks.append((lo, e - lo))
                    lo = e
            if lo < n_a:
                chunks.append((lo, n_a - lo))
        self.chunks = chunks
        self.events = [torch.cuda.Event() for _ in chunks]
        self.chunk_of = {}
        for p, o in zip(opt._params, opt._offs):
            last = o + max(p.numel(), 1) - 1
            for i, (f, c) in enumerate(chunks):
                if f <= last < f + c:
                    self.chunk_of[id(p)] = i
                    break
        self.pending = False
        self.waited: dict = {}           # stream id -> highest chunk this stream already waits for
        self.wait_cache: dict = {}
        self.handles = [m.register_forward_pre_hook(self._pre) for m in model.modules()]
        self.handles.append(model.register_forward_hook(self._post))

    def remove(self):
        for h in self.handles:
            h.remove()
        self.handles = []

    def _own_chunk(self, mod) -> int:
        """the chunk a stream has to wait for before `mod`'s forward runs: none for a declared container (its forward reads
        parameters only through child modules' __call__ or behind an explicit streams.gate), else the last chunk that holds
        any parameter of its subtree"""
        if getattr(mod, "_hamt_container", False) or isinstance(mod, (torch.nn.ModuleList, torch.nn.ModuleDict, torch.nn.Sequential)):
            return -1
        c = -1
        for p in mod.parameters():
            c = max(c, self.chunk_of.get(id(p), -1))
        return c

    def _wait(self, c: int):
        if c < 0:
            return
        st = torch.cuda.current_stream()
        if self.waited.get(st.cuda_stream, -1) >= c:
            return
        st.wait_event(self.events[c])
        self.waited[st.cuda_stream] = c

    def inherit(self, dst, src):
        c = self.waited.get(src.cuda_stream, -1)
        if c > self.waited.get(dst.cuda_stream, -1):
            self.waited[dst.cuda_stream] = c

    def gate(self, what):
        if not self.pending:
            return
        c = -1
        for w in what:
            if isinstance(w, torch.nn.Module):
                for p in w.parameters():
                    c = max(c, self.chunk_of.get(id(p), -1))
            elif w is not None:
                c = max(c, self.chunk_of.get(id(w), -1))
        self._wait(c)

    def _pre(self, mod, args):
        if not self.pending:
            return
        c = self.wait_cache.get(mod)
        if c is None:
            c = self.wait_cache[mod] = self._own_chunk(mod)
        self._wait(c)

    def check_read(self, p):
        """debugging aid (tests): is a read of parameter `p` on the current stream ordered behind its chunk?"""
        if not self.pending:
            return True
        c = self.chunk_of.get(id(p), -1)
        return c < 0 or self.waited.get(torch.cuda.current_stream().cuda_stream, -1) >= c

    def _post(self, mod, args, out):
        self.opt.wait_update()

    def mark_pending(self):
        """the chunks' events have just been recorded on `stream` (in chunk order): gate every later parameter read behind them"""
        from .. import streams
        self.pending = True
        self.waited.clear()
        streams.pending_updates.add(self.opt)


class AdamW(Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0, correct_bias=True):
        if lr < 0.0:
            raise ValueError("Invalid learning rate: {} - should be >= 0.0".format(lr))
        if not 0.0 <= betas[0] < 1.0:
            raise ValueError("Invalid beta parameter: {} - should be in [0.0, 1.0[".format(betas[0]))
        if not 0.0 <= betas[1] < 1.0:
            raise ValueError("Invalid beta parameter: {} - should be in [0.0, 1.0[".format(betas[1]))
        if not 0.0 <= eps:
            raise ValueError("Invalid epsilon value: {} - should be >= 0.0".format(eps))
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, correct_bias=correct_bias))
        self._built = False
        self._pending_clip = None   # (gnorm_sq device scalar, max_norm)
        self._packed = False
        self._active: Optional[List[bool]] = None
        self._ov = None             # _Overlap: the update runs on its own stream next to the following forward pass (attach())
        self._gather_ov = None      # _Overlap over the sharded exchange's ranges: the next forward is gated range by range behind the all-gathers
        self._table_ready = False   # the device table already describes the step whose gradients are packed now
        self._dp_wrappers = weakref.WeakSet()   # parallel.ArenaDataParallel wrappers whose backward passes fill this optimizer's arena
        self._counted = None        # flags of the step host_table() COUNTED and no update launch has consumed yet (zero_grad() takes it back)
        self._table_lrs = None
        self._fused = None          # (partial sums tensor, parameter index array): gradients whose sum of squares wgrad.py already has
        self._sharded_sync = None   # parallel.ShardedGradSync attached to this optimizer (state lives sharded over the ranks)
        self._shard_stale = False

    # ---------------------------------------------------------------- arenas
    def _build(self):
        ps, gidx = [], []
        for gi, group in enumerate(self.param_groups):
            for p in group["params"]:
                ps.append(p)
                gidx.append(gi)
        if not ps:
            raise ValueError("AdamW: no parameters")
        if not ps[0].is_cuda:
            raise L.HamtError("AdamW: parameters must live on the GPU (no CPU fallback)")
        b0 = self.param_groups[0]["betas"], self.param_groups[0]["eps"]
        for g in self.param_groups:
            if (g["betas"], g["eps"]) != b0:
                raise ValueError("AdamW: betas/eps must be the same in every group (they are in the reference)")
        dev = ps[0].device
        # Arena order: first the parameters the forward / backward only ever READ THROUGH THE bf16 SHADOW (the GEMM weights),
        # then everything that is read in fp32 (biases, LayerNorm, embedding tables, skinny / odd-width weights).  A sharded
        # data-parallel optimizer (parallel.ShardedGradSync) all-gathers the first region as bf16 and the second as fp32;
        # both regions end on a multiple of REGION_ALIGN elements so that they split evenly over up to 64 ranks.  Relative
        # order inside a region is the registration order (query / key / value weights stay adjacent: blocks._packed).
        order = sorted(range(len(ps)), key=lambda i: 0 if shadow_only(ps[i]) else 1)
        ps, gidx = [ps[i] for i in order], [gidx[i] for i in order]
        offs, n, n_a = [], 0, None
        for p in ps:
            if n_a is None and not shadow_only(p):
                n = (n + REGION_ALIGN - 1) // REGION_ALIGN * REGION_ALIGN
                n_a = n
            offs.append(n)
            n += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        n = (n + REGION_ALIGN - 1) // REGION_ALIGN * REGION_ALIGN
        self._n_shadow_only = n if n_a is None else n_a
        self._params, self._gidx, self._offs, self._n = ps, gidx, offs, n
        self._flat_p = torch.zeros(n, dtype=torch.float32, device=dev)
        for p, o in zip(ps, offs):
            self._flat_p[o:o + p.numel()].copy_(p.data.reshape(-1))
            p.data = self._flat_p[o:o + p.numel()].view(p.shape)
        self._flat_g = torch.zeros_like(self._flat_p)
        self._flat_m = torch.zeros_like(self._flat_p)
        self._flat_v = torch.zeros_like(self._flat_p)
        self._flat_p16 = torch.empty(n, dtype=torch.bfloat16, device=dev)
        L.check(L.load().hamt_cast_f32_bf16(n, _p(self._flat_p), _p(self._flat_p16), _stream()), "hamt_cast_f32_bf16")
        self._steps = np.zeros(len(ps), dtype=np.int64)
        # GEMM weights of the bf16 path: their gradient slots are STORED by the grouped weight-gradient launch (or copied over by
        # _pack_grads), never accumulated into from zero -> the update leaves them alone instead of zeroing 4 bytes per parameter
        self._keep = np.array([1.0 + float(KEEP_GRAD and shadow_only(p)) for p in ps], dtype=np.float32)
        self._gidx_np = np.asarray(gidx, dtype=np.int64)
        ends = offs[1:] + [n]          # (alignment gaps count as the tail of the parameter in front of them: zeros that stay zeros)
        self._ends = torch.tensor(ends, dtype=torch.int32, device=dev)
        # ring of pinned staging buffers: the async H2D copy of step k must have run before its slot is rewritten
        self._hyp_ring = [torch.zeros(len(ps), 4, dtype=torch.float32).pin_memory() for _ in range(4)]
        self._hyp_events = [None] * 4
        self._hyp_slot = 0
        self._hyp = torch.zeros(len(ps), 4, dtype=torch.float32, device=dev)
        self._gnorm = torch.zeros(1, dtype=torch.float32, device=dev)
        self._ws = torch.empty(L.workspace_bytes(L.WS_SUMSQ) // 4, dtype=torch.float32, device=dev)
        self._sync_shadow_views()
        self._publish_grad_slots()
        self._built = True

    def _publish_grad_slots(self):
        """Let the grouped weight-gradient launch (wgrad.py) write straight into the flat gradient arena."""
        import weakref
        me = weakref.ref(self)
        self._index_of = {id(p): i for i, p in enumerate(self._params)}
        for p, o, k in zip(self._params, self._offs, self._keep):
            p._hamt_grad_slot = self._flat_g[o:o + p.numel()].view(p.shape)
            p._hamt_opt = me
            p._hamt_slot_zeroed = bool(k == 1.0)      # producers that ADD into the slot from zero (ops.GatherRowsFn, LayerNorm partials) need this

    def _sync_shadow_views(self):
        """Publish the bf16 shadow of every >=2-D parameter as the GEMM weight operand (see ops.weight_operand)."""
        ver = self._flat_p._version
        for p, o in zip(self._params, self._offs):
            if p.dim() >= 2:
                p._hamt_arena16 = (self._flat_p16[o:o + p.numel()].view(p.shape), self._flat_p, ver, p._version)

    def update_bytes(self, active=None) -> float:
        """algorithmic HBM bytes of one update of every parameter (or of those flagged in `active`, one bool per parameter): read
        p, g, m, v + write p, m, v + bf16 shadow = 30 per element, + 4 where the gradient slot is zeroed as well"""
        self.materialize()
        sizes = np.diff(np.concatenate([[0], self._ends.cpu().numpy()])).astype(np.float64)
        per = sizes * np.where(self._keep == 2.0, 30.0, 34.0)
        if active is not None:
            per = per * np.asarray(active, dtype=np.float64)
        return float(per.sum())

    def refresh_shadow(self):
        """Re-derive the bf16 shadow arena from the fp32 masters (after loading / broadcasting parameters in place)."""
        self.wait_update()
        L.check(L.load().hamt_cast_f32_bf16(self._n, _p(self._flat_p), _p(self._flat_p16), _stream()), "hamt_cast_f32_bf16")
        torch.autograd.graph.increment_version(self._flat_p)
        self._sync_shadow_views()

    def materialize(self):
        """Build the flat arenas now (re-homes p.data); call before wrapping the model in DDP / capturing a graph."""
        if not self._built:
            self._build()
        return self

    def _pack_grads(self):
        """Copy the autograd-produced gradients into the flat arena (one fused multi-tensor copy)."""
        self.materialize()
        self.wait_update()
        src, dst, active = [], [], []
        for p, o in zip(self._params, self._offs):
            a = p.grad is not None
            active.append(a)
            if a and p.grad.data_ptr() != self._flat_g.data_ptr() + 4 * o:   # else: already written in place (wgrad.py)
                src.append(p.grad.reshape(-1))
                dst.append(self._flat_g[o:o + p.numel()])
        if self._fused is not None:
            # the tile sums describe gradients that sit in their slots as the weight-gradient launch left them: a fused parameter
            # whose .grad was replaced (averaged over micro-batches / ranks into a new tensor) or dropped invalidates them
            base = self._flat_g.data_ptr()
            for i in self._fused[1]:
                p = self._params[i]
                if p.grad is None or p.grad.data_ptr() != base + 4 * self._offs[i]:
                    self._fused = None
                    break
        if src:
            torch._foreach_copy_(dst, src)
        self._active = active
        self._packed = True
        self._table_ready = False

    def global_grad_sumsq(self) -> torch.Tensor:
        """device scalar sum(g^2) over every parameter that has a gradient (the active rows of the device table)."""
        if self._fused is not None and self._fused[2] != self._flat_g._version:
            self._fused = None              # the gradients were touched after the pass that produced the tile sums
            self._table_ready = False
        if not self._packed:
            self._pack_grads()
        self.wait_update()
        if self._ov is not None:
            L.check(L.load().hamt_sumsq(self._n, _p(self._flat_g), _p(self._gnorm), 0, _p(self._ws), _stream()), "hamt_sumsq")
            return self._gnorm
        if not torch.cuda.is_current_stream_capturing():
            self._ensure_table()            # (a captured step: the caller refreshes the table before every replay)
        # two launches: the GEMM-weight region (mostly skipped when the tile sums exist: what is left -- weights written twice in
        # the pass, ineligible shapes -- is scattered) and the fp32-read region (dense): each gets its own grid, else the few
        # blocks that own the dense tail of ONE grid do all the reading (measured 102 us for 100 MB)
        n_a = self._n_shadow_only
        lib = L.load()
        if 0 < n_a < self._n:
            L.check(lib.hamt_sumsq_table(0, n_a, _p(self._flat_g), _p(self._ends), _p(self._hyp), len(self._params),
                                         _p(self._gnorm), L.SUMSQ_SPARSE if self._fused is not None else 0, _p(self._ws), _stream()), "hamt_sumsq_table")
            L.check(lib.hamt_sumsq_table(n_a, self._n - n_a, _p(self._flat_g[n_a:]), _p(self._ends), _p(self._hyp), len(self._params),
                                         _p(self._gnorm), 1, _p(self._ws), _stream()), "hamt_sumsq_table")
        else:
            L.check(lib.hamt_sumsq_table(0, self._n, _p(self._flat_g), _p(self._ends), _p(self._hyp), len(self._params),
                                         _p(self._gnorm), 0, _p(self._ws), _stream()), "hamt_sumsq_table")
        if self._fused is not None:         # the table (flag 3) skipped these: their tiles' sums of squares
            for ss in self._fused[0]:
                L.check(L.load().hamt_sumsq_partials(ss.numel(), _p(ss), _p(self._gnorm), 1, _stream()), "hamt_sumsq_partials")
        return self._gnorm

    # ---------------------------------------------------------------- step = host part + launches
    def note_fused_sumsq(self, ss: torch.Tensor, params, append: bool = False):
        """wgrad.py: the grouped weight-gradient launch of this pass left the sum of squares of these parameters' gradients in
        `ss` (one float per output tile): the norm kernel skips them (table flag 3) and adds sum(ss) instead.  append: a second
        launch of the same pass adds its tiles to the first one's."""
        if self._ov is not None:        # (update at the head of the next replay: the table describes the previous step, see attach())
            return
        # valid for as long as nothing else writes the gradient arena: any torch op on a gradient (averaging over micro-batches or
        # ranks, scaling, a copy) bumps the arena's version counter and global_grad_sumsq then reduces everything from memory
        idx = np.asarray([self._index_of[id(p)] for p in params], dtype=np.int64)
        if append and self._fused is not None:
            self._fused = (self._fused[0] + [ss], np.concatenate([self._fused[1], idx]), self._flat_g._version)
        else:
            self._fused = ([ss], idx, self._flat_g._version)

    def clear_fused_sumsq(self):
        self._fused = None

    def table_flags(self, active=None) -> np.ndarray:
        """per parameter: 0 no gradient this step, 1 update + zero the gradient slot, 2 update and leave the slot to its producer,
        3 like 2 and the gradient's sum of squares is already known (note_fused_sumsq)"""
        if active is None:
            active = self._active if self._active is not None else [p.grad is not None for p in self._params]
        act = np.asarray(active)
        if act.dtype != np.bool_:
            return act.astype(np.float32)          # already flags (a captured step's own)
        f = act.astype(np.float32) * self._keep
        if self._fused is not None:
            idx = self._fused[1]
            f[idx] = np.where(f[idx] == 2.0, 3.0, f[idx])
        return f

    def host_table(self, active: Optional[List[bool]] = None, advance: bool = True) -> np.ndarray:
        """Host side of a step: advance the per-parameter step counts (unless `advance` is False: the table of the step that was
        already counted, e.g. rebuilt for new learning rates) and return the [nparams, 4] table {lr, bias-corrected step size,
        weight decay, active (2: leave the gradient slot unzeroed)} for `upload_table`.  `active` defaults to "has a gradient now"."""
        self.materialize()
        flags = self.table_flags(active)
        act = flags != 0
        if advance:
            self._steps[act] += 1
            self._counted = act.copy()
        t = np.maximum(self._steps, 1).astype(np.float64)
        b1, b2 = self.param_groups[0]["betas"]
        lr = np.array([g["lr"] for g in self.param_groups], dtype=np.float64)[self._gidx_np]
        wd = np.array([g["weight_decay"] for g in self.param_groups], dtype=np.float64)[self._gidx_np]
        cb = np.array([bool(g["correct_bias"]) for g in self.param_groups])[self._gidx_np]
        ss = np.where(cb, lr * np.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t), lr)
        h = np.empty((len(self._params), 4), dtype=np.float32)
        h[:, 0], h[:, 1], h[:, 2], h[:, 3] = lr, ss, wd, flags
        self._table_active = flags
        return h

    def upload_table(self, table: Optional[np.ndarray]):
        """Async copy of a host_table() result (None: nothing active -- the update kernel then touches no parameter) into the
        device table, from a ring of pinned staging buffers, on the current stream."""
        self.materialize()
        k = self._hyp_slot
        self._hyp_slot = (k + 1) % len(self._hyp_ring)
        if self._hyp_events[k] is not None:
            self._hyp_events[k].synchronize()   # only blocks when the GPU is >= 4 steps behind the host
        h = self._hyp_ring[k].numpy()           # pinned memory, shared with the tensor
        if table is None:
            h[:] = 0.0
        else:
            h[:] = table
        self._hyp.copy_(self._hyp_ring[k], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._hyp_events[k] = ev

    def prepare_step(self, active: Optional[List[bool]] = None):
        """Host side of a step: advance the per-parameter step counts, refresh the device hyper-parameter table
        (async copy from pinned memory on the current stream).  `active` defaults to "has a gradient now"."""
        self.upload_table(self.host_table(active))
        self._table_ready = True
        self._table_lrs = [g["lr"] for g in self.param_groups]

    def _ensure_table(self):
        """The eager loop is clip_grad_norm_ -> step(): the norm already needs the table (which parameters are active), the
        update needs it too -- built once per step, rebuilt (without counting the step again) if the learning rates changed
        in between."""
        if not self._table_ready:
            self.prepare_step()
        elif self._table_lrs != [g["lr"] for g in self.param_groups]:
            self.upload_table(self.host_table(self._table_active, advance=False))
            self._table_lrs = [g["lr"] for g in self.param_groups]

    def launch_step(self, zero_grad_arena: bool = True):
        """Device side of a step (static launch sequence; capturable)."""
        b1, b2 = self.param_groups[0]["betas"]
        gn, max_norm = self._pending_clip if self._pending_clip is not None else (None, 0.0)
        self._counted = None        # the counted step is applied by this launch
        if zero_grad_arena:
            self._flat_g._hamt_dirty = False
        L.check(L.load().hamt_adamw_table(self._n, _p(self._flat_p), _p(self._flat_g), _p(self._flat_m), _p(self._flat_v),
                                          _p(self._flat_p16), _p(self._ends), _p(self._hyp), len(self._params), _p(gn),
                                          float(max_norm), b1, b2, self.param_groups[0]["eps"], int(zero_grad_arena), _stream()),
                "hamt_adamw_table")

    # ---------------------------------------------------------------- update next to the following forward pass
    def attach(self, model: torch.nn.Module, chunk_elems: int = 8 << 20):
        """Let `step()` (and graph.GraphedTrainStep) run the update on a stream of its own, chunk by chunk in the order the
        forward pass reads the parameters, while the NEXT forward pass of `model` already runs: the update streams 34 bytes
        per parameter through HBM and no matrix core, the forward pass is the opposite.  Forward pre-hooks on the modules of
        `model` make the stream a module runs on wait for the chunk(s) holding the parameters of that module's subtree --
        except for classes that declare `_hamt_container = True` (their forward reads parameters only through child modules'
        __call__, or behind an explicit streams.gate(...)): without such declarations the root waits for everything and
        nothing overlaps, which is slow but never wrong.  A forward hook on `model` itself waits for the whole update, so everything behind a forward pass (backward, gradient norm,
        the next update) is ordered as before.  Anything else that reads parameters or the arenas directly must call
        `wait_update()` first (state_dict / load_state_dict / refresh_shadow / gradient packing here do)."""
        self.materialize()
        self._ov = _Overlap(self, model, chunk_elems)
        # (graph.GraphedTrainStep replays the update of step t at the head of step t+1: the device table then describes the
        # PREVIOUS step while this step's norm is reduced -> zero every slot and reduce the whole arena, as before)
        self._keep[:] = 1.0
        self._publish_grad_slots()
        return self

    def detach(self):
        if self._ov is not None:
            self.wait_update()
            self._ov.remove()
            self._ov = None

    def _gates(self):
        """the overlap objects whose events gate parameter reads: the update on its own stream (attach) and / or the parameters'
        all-gathers of the sharded exchange on the communication stream (parallel.ShardedGradSync.attach_gather)"""
        return [g for g in (self._ov, self._gather_ov) if g is not None]

    def wait_update(self, gathers: bool = True):
        """Order the current stream behind an update still running on the update stream -- and (`gathers`) behind the parameter
        all-gathers of the sharded exchange still running on the communication stream (no-op otherwise)."""
        for ov in ((self._ov, self._gather_ov) if gathers else (self._ov,)):
            if ov is not None and ov.pending:
                torch.cuda.current_stream().wait_event(ov.events[-1])
                ov.pending = False
        if not any(g.pending for g in self._gates()):
            streams.pending_updates.discard(self)

    def launch_step_overlapped(self, gnorm_sq=None, max_norm: float = 0.0):
        """launch_step on the update stream, one launch + one event per chunk (capturable: a fork of the capturing stream that
        the forward hooks join again)."""
        ov = self._ov
        b1, b2 = self.param_groups[0]["betas"]
        if gnorm_sq is None and self._pending_clip is not None:
            gnorm_sq, max_norm = self._pending_clip
        self._flat_g._hamt_dirty = False
        self._counted = None
        lib = L.load()
        ov.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(ov.stream):
            for i, (f, c) in enumerate(ov.chunks):
                L.check(lib.hamt_adamw_table_range(f, c, _p(self._flat_p[f:f + c]), _p(self._flat_g[f:f + c]), _p(self._flat_m[f:f + c]),
                                                   _p(self._flat_v[f:f + c]), _p(self._flat_p16[f:f + c]), _p(self._ends), _p(self._hyp),
                                                   len(self._params), _p(gnorm_sq), float(max_norm), b1, b2, self.param_groups[0]["eps"], 1,
                                                   _stream()), "hamt_adamw_table_range")
                ov.events[i].record(ov.stream)
        ov.pending = True
        ov.waited.clear()
        streams.pending_updates.add(self)

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for w in list(self._dp_wrappers):
            w._on_update(True)
        sync = self._sharded_sync
        if sync is not None and sync._unconsumed:
            # the gradients of this step were reduce-scattered (parallel.ShardedGradSync, e.g. under parallel.ArenaDataParallel): the
            # update is the owned-slice AdamW + the parameter all-gathers; the clip -- if clip_grad_norm_ ran -- is fused into it
            self._ensure_table()
            max_norm = self._pending_clip[1] if self._pending_clip is not None else 0.0
            sync.update(max_norm, norm_reduced=sync._norm_reduced)
            return loss
        if not self._packed:
            self._pack_grads()
        self._ensure_table()
        if self._ov is not None:
            self.launch_step_overlapped()
        else:
            self.launch_step()
        self.mark_updated()
        return loss

    def mark_updated(self):
        """Parameters changed outside autograd's view: bump the version counter (cached transposed shadows are
        rebuilt lazily) and re-publish the bf16 shadow written by the update kernel."""
        torch.autograd.graph.increment_version(self._flat_p)
        self._sync_shadow_views()
        self._pending_clip = None
        self._packed = False
        self._active = None
        self._table_ready = False
        self._fused = None

    def zero_grad(self, set_to_none: bool = True):
        super().zero_grad(set_to_none=set_to_none)
        for w in list(self._dp_wrappers):
            w._on_update(False)
        self._packed = False
        self._fused = None
        if self._built and self._counted is not None:
            # a step was COUNTED (clip_grad_norm_ / prepare_step built the table: which parameters have a gradient, step counts advanced) and
            # no update launch consumed it -- the loop dropped this pass (NaN guard, early `continue`).  The reference counts a parameter's
            # step inside step() only (optim/adamw.py:76-84): take the count back, and let the next pass build its own table.  The flag is
            # `_counted` (set by host_table(advance=True), cleared by launch_step / launch_step_overlapped / the sharded update / a
            # captured update's replay), not `_table_ready`: prepare_step() + launch_step() called directly leave `_table_ready` set
            # after a real update
            self._steps[self._counted] -= 1
            self._counted = None
            self._pending_clip = None
        self._table_ready = False
        self._active = None
        from .. import wgrad
        sync = getattr(self, "_sharded_sync", None)
        if sync is not None and sync._unconsumed:
            # an exchange ran and no update consumed it (a skipped step): the arena holds rank means in the owned chunks only -- drop it
            # all, and let the next pass exchange again instead of raising
            sync.discard()
            self._pending_clip = None
            if self._built:
                self._flat_g._hamt_dirty = True
        if self._built and getattr(self._flat_g, "_hamt_dirty", False):
            self.wait_update(gathers=False)      # (an update on its own stream reads the gradient arena; the parameters' all-gathers do not)
            # something accumulated into arena slots it assumed zero and no update (which zeroes the arena) has run since
            self._flat_g.zero_()
            self._flat_g._hamt_dirty = False
        dev = self.param_groups[0]["params"][0].device
        if dev.type == "cuda" and torch._C._current_graph_task_id() < 0 and wgrad.pending(dev):
            wgrad.reset(dev)                     # leftovers of a backward pass that raised: never carry them into the next step

    # ---------------------------------------------------------------- checkpointing (utils/save.py:42-45 saves optimizer.state_dict())
    def state_dict(self):
        """The reference optimizer's layout (optim/adamw.py:76-84): per stepped parameter {'step', 'exp_avg', 'exp_avg_sq'}
        (copies of the arena slots), plus param_groups -- what ModelSaver writes to train_state_*.pt."""
        if not self._built:
            return super().state_dict()
        if getattr(self, "_shard_stale", False):
            raise RuntimeError("AdamW.state_dict(): the optimizer state is sharded over the ranks (parallel.ShardedGradSync) and this rank's "
                               "copy of the other ranks' exp_avg / exp_avg_sq / fp32 masters is out of date: call grad_sync.gather_state() "
                               "on EVERY rank first")
        self.wait_update()
        self.state.clear()
        for i, (p, o) in enumerate(zip(self._params, self._offs)):
            if self._steps[i] > 0:
                n = p.numel()
                self.state[p] = {"step": int(self._steps[i]), "exp_avg": self._flat_m[o:o + n].view(p.shape).clone(),
                                 "exp_avg_sq": self._flat_v[o:o + n].view(p.shape).clone()}
        try:
            return super().state_dict()
        finally:
            self.state.clear()

    @torch.no_grad()
    def load_state_dict(self, state_dict):
        """Restore moments and per-parameter step counts into the arenas (and re-derive the bf16 weight shadow: resuming
        normally follows a model.load_state_dict)."""
        super().load_state_dict(state_dict)
        self.materialize()
        self.wait_update()
        self._flat_m.zero_()
        self._flat_v.zero_()
        self._steps[:] = 0
        for i, (p, o) in enumerate(zip(self._params, self._offs)):
            st = self.state.get(p)
            if st:
                n = p.numel()
                self._flat_m[o:o + n].copy_(st["exp_avg"].reshape(-1))
                self._flat_v[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
                self._steps[i] = int(st["step"])
        self.state.clear()
        self.refresh_shadow()

    @property
    def active_mask(self):
        return self._active


def clip_grad_norm_(parameters: Iterable[torch.Tensor], max_norm: float, optimizer: AdamW = None) -> torch.Tensor:
    """torch.nn.utils.clip_grad_norm_ (main_r2r.py:271-273) on the GPU: returns the total L2 norm (device tensor,
    no host sync).  With `optimizer` (our AdamW; found through the parameters when the call names none, as the reference's
    does, and the list is exactly that optimizer's parameter set) the scaling min(1, max_norm/(norm+1e-6)) is DEFERRED: it is
    applied inside the next ``optimizer.step()`` -- `.grad` still holds the unscaled gradients after this call (torch scales
    them in place; code that reads `.grad` between the clip and the step sees the difference, the update does not).  Without
    an optimizer the gradients are scaled in place."""
    lib = L.load()
    if optimizer is None:
        # the reference's call has no optimizer argument (main_r2r.py:271-273): parameters that live in an AdamW arena name it
        parameters = list(parameters)
        for p in parameters:
            r = getattr(p, "_hamt_opt", None)
            if r is not None:
                optimizer = r()
                break
        # the fused path clips EXACTLY the optimizer's parameters: the caller's list must be that set (by identity, not by count -- a list of
        # the same length holding another model's tensors, or one parameter twice, takes the generic path below)
        if optimizer is not None and not (optimizer._built and len(parameters) == len(optimizer._params)
                                          and all(id(p) in optimizer._index_of for p in parameters)
                                          and len({id(p) for p in parameters}) == len(parameters)):
            optimizer = None
    if optimizer is not None:
        sync = optimizer._sharded_sync
        if sync is not None and sync._unconsumed:      # reduce-scattered gradients: the norm over the owned chunks, one 4-byte all-reduce
            gn = sync.global_norm()
            optimizer._pending_clip = (sync._gsq, float(max_norm))
            return gn
        gsq = optimizer.global_grad_sumsq()
        optimizer._pending_clip = (gsq, float(max_norm))
        return gsq.sqrt()[0]
    grads = [p.grad for p in parameters if p.grad is not None]
    if not grads:
        return torch.zeros(())
    dev = grads[0].device
    gsq = torch.zeros(1, dtype=torch.float32, device=dev)
    ws = torch.empty(1024, dtype=torch.float32, device=dev)
    flat = [g.contiguous() for g in grads]
    for i, g in enumerate(flat):
        if g.data_ptr() % 16:
            g = g.clone()
            flat[i] = g
        L.check(lib.hamt_sumsq(g.numel(), _p(g), _p(gsq), int(i > 0), _p(ws), _stream()), "hamt_sumsq")
    for g, orig in zip(flat, grads):
        L.check(lib.hamt_clip_scale(g.numel(), _p(g), _p(gsq), float(max_norm), _stream()), "hamt_clip_scale")
        if g.data_ptr() != orig.data_ptr():
            orig.copy_(g)
    return gsq.sqrt()[0]
