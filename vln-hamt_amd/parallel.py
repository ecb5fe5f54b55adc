"""Data-parallel plumbing: one process per GPU, torch.distributed over RCCL (backend "nccl" on ROCm).

The HAMT path shards by *samples* only: every rank runs the same task on its own minibatch and
the only exchange is the gradient average.  The reference gets it from torch DDP (pretrain_src/utils/misc.py:52-65,
find_unused_parameters=True because each proxy task leaves the other tasks' heads without gradient).  Here the
gradients already sit in ONE flat fp32 arena (optim.AdamW; the GEMM weight gradients are written there directly by
the grouped end-of-pass launch, wgrad.py, which bypasses the per-parameter autograd hooks DDP relies on), so the
exchange is a few large RCCL all-reduces over that arena (`allreduce_grads`): ring collectives over xGMI are per-link
bound, and 128 MiB messages run at the link rate without per-bucket bookkeeping or unused-parameter detection.
Unused heads have all-zero slots on every rank, and stay "inactive" for AdamW exactly like `grad is None` in the
reference.  `wrap_ddp` remains for models whose gradients all flow through autograd (the CPU oracle in the gloo
test, fp32 mode).  The reference also broadcasts the sampled task id every step (data/loader.py:56-59); here every
rank derives it from a shared-seed host RNG instead, so no collective is needed for it.
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.distributed as dist

from .synth import MIX_RATIO, TASKS


def dist_env():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init_distributed(backend: str | None = None):
    """Initialise the default process group from torchrun's env (MASTER_ADDR defaults to 127.0.0.1)."""
    rank, local_rank, world = dist_env()
    force = os.environ.get("HAMT_FORCE_DIST") is not None     # exercise the multi-rank code path with one rank
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


class TaskSchedule:
    """Per-step task choice with the reference's mix ratio (pretrain_r2r.json:43-58; MetaLoader samples with
    torch.multinomial and broadcasts, loader.py:56-59).  Seeded identically on every rank => same task everywhere."""

    def __init__(self, tasks=TASKS, ratios=None, seed: int = 0, cyclic: bool = True):
        ratios = ratios or MIX_RATIO
        self.tasks = list(tasks)
        self.cyclic = cyclic
        if cyclic:      # deterministic interleaving of a ratio-exact cycle (used by the bench: fixed work per cycle)
            pool = {t: ratios[t] for t in self.tasks}
            cyc, total = [], sum(pool.values())
            credit = {t: 0.0 for t in self.tasks}
            for _ in range(total):
                for t in self.tasks:
                    credit[t] += pool[t] / total
                t = max(self.tasks, key=lambda k: credit[k])
                credit[t] -= 1.0
                cyc.append(t)
            self.cycle = cyc
        else:
            self.rng = np.random.Generator(np.random.PCG64(seed))
            w = np.array([ratios[t] for t in self.tasks], dtype=np.float64)
            self.p = w / w.sum()

    def task_at(self, step: int) -> str:
        if self.cyclic:
            return self.cycle[step % len(self.cycle)]
        return self.tasks[int(self.rng.choice(len(self.tasks), p=self.p))]


def wrap_ddp(model, local_rank: int):
    """DistributedDataParallel exactly as the reference wraps it (utils/misc.py:57-58)."""
    from torch.nn.parallel import DistributedDataParallel as DDP
    if next(model.parameters()).is_cuda:
        return DDP(model, device_ids=[local_rank], output_device=local_rank, find_unused_parameters=True)
    return DDP(model, find_unused_parameters=True)


def allreduce_mean_(flat: torch.Tensor, chunk_elems: int = 32 << 20) -> torch.Tensor:
    """In-place average of a flat tensor over all ranks, in `chunk_elems`-element (128 MiB fp32) all-reduces."""
    if not (dist.is_available() and dist.is_initialized()):
        return flat
    world = dist.get_world_size()
    avg = dist.get_backend() == "nccl"          # RCCL reduces with the 1/world scale fused; gloo has no AVG
    for o in range(0, flat.numel(), chunk_elems):
        c = flat[o:o + chunk_elems]
        if avg:
            dist.all_reduce(c, op=dist.ReduceOp.AVG)
        else:
            dist.all_reduce(c, op=dist.ReduceOp.SUM)
            c.mul_(1.0 / world)
    return flat


def allreduce_grads(optimizer) -> None:
    """Average this step's gradients over the ranks: pack what autograd produced into the optimizer's flat arena
    (the grouped weight gradients are already there) and all-reduce the arena.  Call between backward and clip/step."""
    if not optimizer._packed:
        optimizer._pack_grads()
    allreduce_mean_(optimizer._flat_g)


class OverlappedGradSync:
    """Gradient exchange overlapped with the weight-gradient GEMMs (the only part of backward that is off the critical
    path, and here deliberately run last, see wgrad.py):

        backward (dgrad chain, queue filled)  ->  pack autograd grads into the arena
        -> for g in launch groups (arena order):   grouped wgrad g on the compute stream
                                                    all-reduce(arena range g) on the communication stream
        -> tail range (biases / LayerNorm / whatever the last group touched) -> join -> clip + AdamW

    so that RCCL moves range g over xGMI while group g+1 multiplies.  Works eagerly (the queue's end-of-backward
    callback runs the whole sequence) and with graph.GraphedTrainStep (forward/backward/pack graph, then this sequence
    launched eagerly from a stored plan, then the update graph).  Use as the `grad_sync` callable."""

    overlapped = True

    def __init__(self, optimizer, n_groups: int = 4):
        from . import wgrad
        self.opt, self.n_groups = optimizer, n_groups
        self.comm = torch.cuda.Stream()
        self.mode = "eager"                # "eager": run at flush; "plan": only build the plan (graph capture)
        self.plan = None
        self.done = False
        wgrad.set_handler(self._on_flush)

    def close(self):
        from . import wgrad
        wgrad.set_handler(None)

    def _on_flush(self, items):
        from . import wgrad
        plan = wgrad.build_plan(items, self.opt, self.n_groups)
        if plan is None:                   # some parameter lives outside the arena: plain semantics
            wgrad.set_handler(None)
            try:
                wgrad._items.extend(items)
                wgrad.flush()
            finally:
                wgrad.set_handler(self._on_flush)
            return
        if self.mode == "plan":
            self.plan = plan
            return
        self.opt._pack_grads()
        self.run(plan)
        self.done = True

    def take_plan(self):
        p, self.plan = self.plan, None
        return p

    def run(self, plan):
        """Launch the plan's groups on the current stream with the arena all-reduces on the communication stream."""
        from . import wgrad
        flat = self.opt._flat_g
        main = torch.cuda.current_stream()
        pending = sorted(plan.ranges, key=lambda r: r[2])
        k = 0

        def reduce_ready(after):
            nonlocal k
            first = True
            while k < len(pending) and pending[k][2] <= after:
                if first:
                    self.comm.wait_stream(main)
                    first = False
                lo, hi, _ = pending[k]
                with torch.cuda.stream(self.comm):
                    allreduce_mean_(flat[lo:hi])
                k += 1

        reduce_ready(-1)
        for g in range(len(plan.groups)):
            wgrad.launch_group(plan, g)
            reduce_ready(g)
        main.wait_stream(self.comm)

    def __call__(self, optimizer):
        """After backward: nothing left to do when the flush already exchanged; else (no queued GEMM gradients in this
        pass, e.g. fp32 mode) the plain flat all-reduce."""
        if self.done:
            self.done = False
            return
        allreduce_grads(optimizer)


def broadcast_params(optimizer, src: int = 0) -> None:
    """Every rank starts from rank `src`'s parameters (what DDP does at construction)."""
    optimizer.materialize()
    if dist.is_available() and dist.is_initialized():
        dist.broadcast(optimizer._flat_p, src)
        optimizer.refresh_shadow()


def max_over_ranks(value: float, device) -> float:
    if not (dist.is_available() and dist.is_initialized()):
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device) -> float:
    if not (dist.is_available() and dist.is_initialized()):
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
