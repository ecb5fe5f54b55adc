"""Whole-step hipGraph capture: forward + backward + gradient packing + norm + AdamW + dropout-epoch bump of one
(task, batch-shape) key are captured once and replayed, removing the ~2000 per-step host launches.

Requirements the rest of the package is written for: every launch goes to torch's current stream, nothing inside a
step synchronises (index lists for MLM/MRC/ITM come with the batch), dropout masks depend on a device-resident epoch,
the optimizer's launch sequence is static (per-parameter hyper-parameters live in a device table the host refreshes
before each replay), lazily cached weight shadows are invalidated before capture so their rebuild is part of the graph.
"""
from __future__ import annotations

import copy
import os
import time

import torch

from . import ops, streams


def _new_graph():
    return torch.cuda.CUDAGraph()


def _finish_graph(g):
    """what to replay for a captured graph (round 4 also cut the captured graph into its linear chains and replayed those on separate
    streams: tools/experiments/graph_split.hip, DESIGN_HISTORY.md -- correct, and no faster)"""
    return g


class GraphedTrainStep:
    """`grad_sync` (multi-GPU): a callable run between backward and the optimizer, e.g. parallel.allreduce_grads.  The
    step is then captured as TWO graphs -- forward/backward/packing per (task, shape) key, and one shared
    norm + AdamW graph -- with the collective launched eagerly between the two replays on the same stream, so RCCL
    never has to be captured."""

    def __init__(self, model, optimizer, max_grad_norm: float = 5.0, grad_sync=None, loss_fn=None, overlap_update=None):
        """`loss_fn(model, batch, task) -> scalar loss` replaces the default `model(batch, task, True).mean()`: e.g. a whole
        finetune rollout (language once, history / visual per step, a loss per step -- agent_cmt.py:248-529) whose ONE
        backward then sits in the same captured graph (`task` is only a label for such a function).

        `overlap_update` (default off; HAMT_OVERLAP_UPDATE=1 turns it on for a single GPU): the parameter update of step t is
        the FIRST thing the replay of step t+1 launches, on a stream of its own next to that step's forward pass
        (optim.AdamW.attach), instead of the last thing of step t.  The parameters then lag the returned loss by one update
        until `finish()` is called (before evaluating, saving or reading parameters; a capture of a new key does it itself).
        Measured on MI355X (profiles/r02_timeline_b16_overlap.txt): no gain -- the update saturates HBM, and the small
        latency-bound kernels of the forward pass running next to it take 3-8x as long (B = 16: 5.89 vs 5.71 ms per step,
        B = 64: 10.87 vs 10.49), so it stays an option, not the default."""
        self.model, self.opt, self.max_norm = model, optimizer, float(max_grad_norm)
        self.grad_sync = grad_sync
        if overlap_update is None:
            overlap_update = grad_sync is None and os.environ.get("HAMT_OVERLAP_UPDATE", "0") == "1"
        self.lag = bool(overlap_update) and grad_sync is None
        self._pending_table = None            # host table of the update the next replay (or finish()) applies
        self.loss_fn = loss_fn or (lambda m, b, t: m(b, t, True).mean())
        self._custom_loss = loss_fn is not None
        # with a process group alive, RCCL's watchdog thread polls its events while we capture: only this thread's
        # calls may be policed by the capture ("global" mode aborts the watchdog with hipErrorCapturedEvent)
        self.capture_mode = "thread_local" if grad_sync is not None else "global"
        self.update_graph = None
        self.graphs = {}
        self.pool = None
        # ONE dedicated stream for warm-up and for every capture: autograd's per-parameter AccumulateGrad nodes remember
        # the stream they were created on; if a later capture ran on a different stream their accumulation kernels
        # would execute outside the capture (run once, never replayed).
        # Default priority (a high-priority capture stream -- the text-side critical path -- was measured: 16.4 ms per B = 64 step
        # instead of 9.9; a non-default priority on EITHER compute stream stops the two from overlapping inside a replay).
        # (one capture stream per process: see streams.role_stream -- it must never coincide with the second compute stream or an update stream)
        self.stream = streams.role_stream(torch.cuda.current_device(), "capture")
        # (the warm-up step before each capture runs on that stream while the parameters' AccumulateGrad nodes may date
        # from an earlier pass on the caller's stream: intended, and synchronised by wait_stream on both sides)
        if hasattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch"):
            torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
        self.opt.materialize()
        if self.lag and self.opt._ov is None:
            self.opt.attach(model)

    def finish(self):
        """Apply the update the last replayed step left pending (overlap_update): call before reading parameters."""
        if self._pending_table is not None:
            self.opt.upload_table(self._pending_table)
            self._pending_table = None
            self.opt._pending_clip = (self.opt._gnorm, self.max_norm)
            self.opt.launch_step()
            self.opt.mark_updated()

    def _eager(self, batch, task):
        loss = self.loss_fn(self.model, batch, task)
        loss.backward()
        if self.grad_sync is not None:
            self.grad_sync(self.opt)
        if getattr(self.grad_sync, "sharded", False):      # reduce-scattered gradients: norm, AdamW over the owned slices, all-gather
            self.opt.prepare_step()
            self.grad_sync.update(self.max_norm)
        else:
            from .optim import clip_grad_norm_
            clip_grad_norm_(self.model.parameters(), self.max_norm, optimizer=self.opt)
            self.opt.step()
        self.opt.zero_grad()
        ops.advance_rng_epoch(loss.device)
        return loss

    def _update(self, dev):
        gsq = self.opt.global_grad_sumsq()
        self.opt._pending_clip = (gsq, self.max_norm)
        self.opt.launch_step()
        ops.advance_rng_epoch(dev)

    @staticmethod
    def key_for(task, batch):
        """A capture key that distinguishes everything a captured step bakes in: the task and every tensor's shape / dtype
        (incl. the data-dependent lengths of `txt_label_idx` / `hist_mrc_idx`)."""
        return (task,) + tuple((k, tuple(v.shape), str(v.dtype)) for k, v in sorted(batch.items()) if torch.is_tensor(v))

    def static_batch(self, key):
        """The captured step's own input tensors: a loader may write the next batch straight into them (data.collate's
        `out=`), in which case step() has nothing to copy."""
        return self.graphs[key][4]

    @staticmethod
    def _check_capturable(batch, task):
        need = {"mlm": "txt_label_idx", "mrc": "hist_mrc_idx"}
        for t, k in need.items():
            if task.startswith(t) and batch.get(k) is None:
                raise ops.L.HamtError(f"GraphedTrainStep: task '{task}' needs batch['{k}'] (the masked positions as an index list, "
                                      "as data.collate emits it): without it the model derives them with nonzero(), a host "
                                      "synchronisation and a data-dependent shape that cannot be captured")

    def _capture(self, key, batch, task):
        if not self._custom_loss:
            self._check_capturable(batch, task)
        self.finish()
        src = batch
        batch = copy.copy(src)                 # the graph's static inputs: replays read THESE tensors (step() refills them)
        for k, v in src.items():
            if torch.is_tensor(v):
                batch[k] = v.clone()
        dev = next(self.model.parameters()).device
        cur = torch.cuda.current_stream()
        side = self.stream
        side.wait_stream(cur)
        with torch.cuda.stream(side):          # one real step on the capture stream first (allocator / cache warm-up)
            loss = self._eager(batch, task).detach()      # .detach(): do not keep this step's autograd graph alive
        cur.wait_stream(side)
        torch.cuda.synchronize()
        self.opt.wait_update()                 # (host flag only: the warm-up step's update has finished)
        if self.grad_sync is not None:
            # RCCL's watchdog thread polls the events of collectives it has not reaped yet, and an event query from any
            # thread while this process captures aborts it (hipErrorCapturedEvent, in "thread_local" mode too on
            # ROCm 7.0 / torch 2.10).  Every collective has finished (synchronize above); give the watchdog (100 ms
            # polling period) time to retire them so that it has nothing to query during the capture.
            time.sleep(0.5)
        ops.invalidate_weight_caches()
        self.opt.zero_grad(set_to_none=True)
        g = _new_graph()
        if self.pool is None:
            self.pool = torch.cuda.graph_pool_handle()
        overl = getattr(self.grad_sync, "overlapped", False)
        if overl:
            self.grad_sync.mode = "plan"       # the end-of-backward flush only builds the plan during the capture
        with torch.cuda.graph(g, pool=self.pool, stream=side, capture_error_mode=self.capture_mode):
            if self.lag:                       # the update of the step replayed before this one (device table: which parameters)
                self.opt.launch_step_overlapped(self.opt._gnorm, self.max_norm)
            loss_c = self.loss_fn(self.model, batch, task)
            loss_c.backward()
            if self.lag:
                self.opt.wait_update()         # (a loss_fn that never ran the attached model's forward: join the update stream)
                self.opt.global_grad_sumsq()   # -> opt._gnorm, read by the update at the head of the next replay
                ops.advance_rng_epoch(dev)
            elif self.grad_sync is None:
                self._update(dev)
            else:
                self.opt._pack_grads()
            loss_c = loss_c.detach()           # drop the captured step's autograd graph (its buffers live in the pool)
        plan = None
        if overl:
            self.grad_sync.mode = "eager"
            plan = self.grad_sync.take_plan()  # the pass's weight-gradient GEMMs: launched between the two graphs
            if plan is not None and os.environ.get("HAMT_NO_GROUP_GRAPHS") is None:
                # each launch group (its table-write kernels + the grouped kernels) as a small graph of its own: replayed on the
                # group's lane stream between the collectives, instead of 3-10 host launches per group in the eager section
                from . import wgrad
                plan.graphs = []
                for gi, (_descs, n_) in enumerate(plan.groups):
                    if not n_:
                        plan.graphs.append(None)
                        continue
                    # the group's launch table is a function of the plan alone (device pointers into the captured step's static
                    # buffers and the arenas): written ONCE, here; the replayed graph is the grouped kernels only.  (With the
                    # table-write kernel inside the graph, the second lane's one-workgroup write sat ~275 us behind the first
                    # lane's chip-filling tiles before it was dispatched: DESIGN 6c)
                    # The table is re-homed first: `plan.tables` were allocated DURING the capture, i.e. in the graph's private pool, where
                    # a block may have been the temporary of an earlier node -- every replay of the step graph scribbles over it (harmless
                    # while each group graph rewrote its table, an aperture violation in the grouped kernel once it did not)
                    with torch.cuda.stream(side):
                        plan.tables[gi] = torch.empty_like(plan.tables[gi])
                        wgrad.launch_group(plan, gi, phase=1)
                    side.synchronize()
                    gg = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gg, pool=self.pool, stream=side, capture_error_mode=self.capture_mode):
                        wgrad.launch_group(plan, gi, phase=2)
                    plan.graphs.append(gg)
            if getattr(self.grad_sync, "sharded", False) and os.environ.get("HAMT_NO_GROUP_GRAPHS") is None:
                self.grad_sync.capture_update(self.max_norm, self.pool, side, self.capture_mode)
        if self.grad_sync is not None and not getattr(self.grad_sync, "sharded", False):
            if self.update_graph is None:      # norm + AdamW over the arena: the same launches for every key
                self.update_graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.update_graph, pool=self.pool, stream=side, capture_error_mode=self.capture_mode):
                    self._update(dev)
        active = self.opt.table_flags()        # per-parameter flags of THIS key's step (incl. which norms the captured wgrad launch supplies)
        self.opt._fused = None
        self.opt._pending_clip = None
        self.opt._packed = False
        for p in self.opt._params:             # the captured gradient buffers stay alive inside the graph's pool
            p.grad = None
        self.graphs[key] = (_finish_graph(g), loss_c, active, plan, batch)
        return loss

    @staticmethod
    def _refill(static, batch, key):
        """Copy `batch` into the captured step's static inputs (same keys, shapes and dtypes, or a loud error: a captured
        graph replays the launches of ITS shapes whatever the caller passes)."""
        if static is batch:
            return
        tk = {k for k, v in batch.items() if torch.is_tensor(v)}
        sk = {k for k, v in static.items() if torch.is_tensor(v)}
        if tk != sk:
            raise ops.L.HamtError(f"GraphedTrainStep.step: batch keys differ from the captured batch of key {key!r}: "
                                  f"missing {sorted(sk - tk)}, unexpected {sorted(tk - sk)}")
        for k in tk:
            v, s_ = batch[k], static[k]
            if v is s_ or (v.data_ptr() == s_.data_ptr() and v.shape == s_.shape):
                continue
            if v.shape != s_.shape or v.dtype != s_.dtype:
                raise ops.L.HamtError(f"GraphedTrainStep.step: batch['{k}'] is {tuple(v.shape)} {v.dtype}, the graph captured under "
                                      f"key {key!r} has {tuple(s_.shape)} {s_.dtype}; use a key that separates them "
                                      "(GraphedTrainStep.key_for) so that the new shape is captured on its own")
            s_.copy_(v, non_blocking=True)

    def step(self, key, batch, task):
        """One optimisation step on `batch` (param_groups' lr must be set by the caller).  Returns the (static) loss tensor.

        Contract: the first call with a `key` runs the step eagerly once and captures it; later calls with that key COPY
        `batch` into the captured step's static input tensors (on the current stream) and replay -- the batch must have the
        captured keys, shapes and dtypes (HamtError otherwise; `key_for(task, batch)` builds a key that guarantees it).
        Tensors that already ARE the static inputs (`static_batch(key)`, e.g. filled by the loader) are not copied.  MLM /
        MRC batches must carry their masked-position index lists (`txt_label_idx` / `hist_mrc_idx`)."""
        ent = self.graphs.get(key)
        if ent is None:
            return self._capture(key, batch, task)
        g, loss_c, active, plan, static = ent
        self._refill(static, batch, key)
        if self.lag:
            self.opt.upload_table(self._pending_table)        # None: no update pending, the leading update touches nothing
            self._pending_table = self.opt.host_table(active)   # this step's update (today's learning rates), applied by the next replay
            self.opt._counted = None                            # (... or by finish(): never taken back by an eager zero_grad())
            g.replay()
            return loss_c
        self.opt.prepare_step(active)
        self.opt._table_ready = False          # (consumed by the captured update below: a later eager zero_grad() must not take the step count back)
        self.opt._counted = None
        g.replay()
        if self.grad_sync is not None:
            sharded = getattr(self.grad_sync, "sharded", False)
            if plan is not None:
                if sharded:
                    self.grad_sync.run(plan, sumsq=True)   # wgrad groups + overlapped reduce-scatters, norm partials per range
                else:
                    self.grad_sync.run(plan)   # wgrad groups + overlapped all-reduces
            else:
                self.opt._packed = True        # the replay packed the gradients; only the collective is left
                self.grad_sync(self.opt)
                self.opt._packed = False
            if sharded:
                self.grad_sync.update(self.max_norm, have_sumsq=plan is not None)   # 4-byte all-reduce, AdamW (captured), all-gathers
                ops.advance_rng_epoch(loss_c.device)
            else:
                self.update_graph.replay()
        return loss_c


class GraphedInference:
    """hipGraph capture of a no-grad callable over tensors of fixed shapes (rollout steps of the finetune model: one
    graph per history length for `visual`, one for `history`).  Inputs are copied into static buffers before every replay;
    the returned tensors are the graph's static outputs (valid until the next call with the same key)."""

    def __init__(self, fn):
        self.fn = fn
        self.graphs = {}
        self.pool = None
        self.stream = streams.role_stream(torch.cuda.current_device(), "capture")

    @torch.no_grad()
    def __call__(self, key, *tensors):
        ent = self.graphs.get(key)
        if ent is None:
            static = [t.clone() if torch.is_tensor(t) else t for t in tensors]
            cur = torch.cuda.current_stream()
            self.stream.wait_stream(cur)
            with torch.cuda.stream(self.stream):
                self.fn(*static)                      # warm-up on the capture stream (allocator, lazy caches)
            cur.wait_stream(self.stream)
            torch.cuda.synchronize()
            g = _new_graph()
            if self.pool is None:
                self.pool = torch.cuda.graph_pool_handle()
            with torch.cuda.graph(g, pool=self.pool, stream=self.stream):
                out = self.fn(*static)
            ent = self.graphs[key] = (_finish_graph(g), static, out)
        g, static, out = ent
        for s_, t in zip(static, tensors):
            if torch.is_tensor(t):
                s_.copy_(t, non_blocking=True)
        g.replay()
        return out
