"""Data-parallel plumbing: one process per GPU, torch.distributed over RCCL (backend "nccl" on ROCm).

The HAMT path shards by *samples* only: every rank runs the same task on its own minibatch and
the only exchange is the gradient all-reduce, which torch DDP buckets and overlaps with backward
(reference: pretrain_src/utils/misc.py:52-65 wraps the model the same way, find_unused_parameters=True because
each proxy task leaves the other tasks' heads without gradient).  The reference also broadcasts the sampled task
id every step (data/loader.py:56-59); here every rank derives it from a shared-seed host RNG instead, so no
collective is needed for it.
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
    if world > 1 and not dist.is_initialized():
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
