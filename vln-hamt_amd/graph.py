"""Whole-step hipGraph capture: forward + backward + gradient packing + norm + AdamW + dropout-epoch bump of one
(task, batch-shape) key are captured once and replayed, removing the ~2000 per-step host launches.

Requirements the rest of the package is written for: every launch goes to torch's current stream, nothing inside a
step synchronises (index lists for MLM/MRC/ITM come with the batch), dropout masks depend on a device-resident epoch,
the optimizer's launch sequence is static (per-parameter hyper-parameters live in a device table the host refreshes
before each replay), lazily cached weight shadows are invalidated before capture so their rebuild is part of the graph.
"""
from __future__ import annotations

import torch

from . import ops


class GraphedTrainStep:
    def __init__(self, model, optimizer, max_grad_norm: float = 5.0):
        self.model, self.opt, self.max_norm = model, optimizer, float(max_grad_norm)
        self.graphs = {}
        self.pool = None
        # ONE dedicated stream for warm-up and for every capture: autograd's per-parameter AccumulateGrad nodes remember
        # the stream they were created on; if a later capture ran on a different stream their accumulation kernels
        # would execute outside the capture (run once, never replayed).
        self.stream = torch.cuda.Stream()
        self.opt.materialize()

    def _eager(self, batch, task):
        loss = self.model(batch, task, True).mean()
        loss.backward()
        from .optim import clip_grad_norm_
        clip_grad_norm_(self.model.parameters(), self.max_norm, optimizer=self.opt)
        self.opt.step()
        self.opt.zero_grad()
        ops.advance_rng_epoch(loss.device)
        return loss

    def _capture(self, key, batch, task):
        dev = next(self.model.parameters()).device
        cur = torch.cuda.current_stream()
        side = self.stream
        side.wait_stream(cur)
        with torch.cuda.stream(side):          # one real step on the capture stream first (allocator / cache warm-up)
            loss = self._eager(batch, task).detach()      # .detach(): do not keep this step's autograd graph alive
        cur.wait_stream(side)
        torch.cuda.synchronize()
        ops.invalidate_weight_caches()
        self.opt.zero_grad(set_to_none=True)
        g = torch.cuda.CUDAGraph()
        if self.pool is None:
            self.pool = torch.cuda.graph_pool_handle()
        with torch.cuda.graph(g, pool=self.pool, stream=side):
            loss_c = self.model(batch, task, True).mean()
            loss_c.backward()
            gsq = self.opt.global_grad_sumsq()
            self.opt._pending_clip = (gsq, self.max_norm)
            self.opt.launch_step()
            ops.advance_rng_epoch(dev)
            loss_c = loss_c.detach()           # drop the captured step's autograd graph (its buffers live in the pool)
        active = list(self.opt.active_mask)
        self.opt._pending_clip = None
        self.opt._packed = False
        for p in self.opt._params:             # the captured gradient buffers stay alive inside the graph's pool
            p.grad = None
        self.graphs[key] = (g, loss_c, active)
        return loss

    def step(self, key, batch, task):
        """One optimisation step (param_groups' lr must be set by the caller).  Returns the (static) loss tensor."""
        ent = self.graphs.get(key)
        if ent is None:
            return self._capture(key, batch, task)
        g, loss_c, active = ent
        self.opt.prepare_step(active)
        g.replay()
        return loss_c
