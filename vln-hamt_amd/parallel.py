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
            # HAMT_DIST_BACKEND=gloo: functional testing of the multi-rank path with several ranks on ONE GPU (gloo stages
            # CUDA tensors through the host; RCCL refuses two ranks per device)
            backend = os.environ.get("HAMT_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
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
            self.seed = int(seed)
            w = np.array([ratios[t] for t in self.tasks], dtype=np.float64)
            self.p = w / w.sum()

    def task_at(self, step: int) -> str:
        if self.cyclic:
            return self.cycle[step % len(self.cycle)]
        # a pure function of (seed, step): ranks agree whatever the order / number of calls each of them makes
        rng = np.random.Generator(np.random.PCG64([self.seed, int(step)]))
        return self.tasks[int(rng.choice(len(self.tasks), p=self.p))]


def wrap_ddp(model, local_rank: int):
    """DistributedDataParallel exactly as the reference wraps it (utils/misc.py:57-58)."""
    from torch.nn.parallel import DistributedDataParallel as DDP
    if next(model.parameters()).is_cuda:
        return DDP(model, device_ids=[local_rank], output_device=local_rank, find_unused_parameters=True)
    return DDP(model, find_unused_parameters=True)


# Wire format of the gradient exchange: "fp32" (DDP's default arithmetic) or "bf16" (the stock DDP bf16_compress_hook's:
# divide by the world size, round to bf16, all-reduce(SUM) in bf16, widen back) -- half the bytes per xGMI link.
# `default_wire(prec)` picks bf16 for the bf16 compute mode (whose weight gradients are products of bf16 images anyway)
# unless HAMT_GRAD_WIRE says otherwise.
_staging: dict = {}


def default_wire(prec: str = "bf16") -> str:
    w = os.environ.get("HAMT_GRAD_WIRE")
    if w is not None:
        assert w in ("fp32", "bf16"), w
        return w
    return "bf16" if prec == "bf16" else "fp32"


def _stage_for(flat: torch.Tensor) -> torch.Tensor:
    """bf16 mirror of the storage `flat` views (same element offsets), allocated once per arena"""
    st = flat.untyped_storage()
    key = (st.data_ptr(), st.nbytes(), str(flat.device))
    buf = _staging.get(key)
    if buf is None:
        buf = _staging[key] = torch.empty(st.nbytes() // 4, dtype=torch.bfloat16, device=flat.device)
    o = flat.storage_offset()
    return buf[o:o + flat.numel()]


def allreduce_mean_(flat: torch.Tensor, chunk_elems: int = 32 << 20, wire: str = "fp32") -> torch.Tensor:
    """In-place average of a flat fp32 tensor over all ranks, in `chunk_elems`-element all-reduces (128 MiB fp32 / 64 MiB
    bf16 messages)."""
    if not (dist.is_available() and dist.is_initialized()):
        return flat
    world = dist.get_world_size()
    if wire == "bf16":
        assert flat.dtype == torch.float32 and flat.is_contiguous()
        stage = _stage_for(flat)
        if flat.is_cuda:
            import ctypes as C
            from . import _lib as L
            from .ops import _p, _stream
            lib = L.load()
            L.check(lib.hamt_wire_pack_bf16(flat.numel(), _p(flat), _p(stage), 1.0 / world, _stream()), "hamt_wire_pack_bf16")
            for o in range(0, flat.numel(), chunk_elems):
                dist.all_reduce(stage[o:o + chunk_elems], op=dist.ReduceOp.SUM)
            L.check(lib.hamt_wire_unpack_bf16(flat.numel(), _p(stage), _p(flat), _stream()), "hamt_wire_unpack_bf16")
        else:                                   # host tensors (gloo tests of the protocol): same arithmetic with torch ops
            stage.copy_(flat * (1.0 / world))
            dist.all_reduce(stage, op=dist.ReduceOp.SUM)
            flat.copy_(stage)
        return flat
    avg = dist.get_backend() == "nccl"          # RCCL reduces with the 1/world scale fused; gloo has no AVG
    for o in range(0, flat.numel(), chunk_elems):
        c = flat[o:o + chunk_elems]
        if avg:
            dist.all_reduce(c, op=dist.ReduceOp.AVG)
        else:
            dist.all_reduce(c, op=dist.ReduceOp.SUM)
            c.mul_(1.0 / world)
    return flat


def allreduce_grads(optimizer, wire: str | None = None) -> None:
    """Average this step's gradients over the ranks: pack what autograd produced into the optimizer's flat arena
    (the grouped weight gradients are already there) and all-reduce the arena.  Call between backward and clip/step."""
    if not optimizer._packed:
        optimizer._pack_grads()
    allreduce_mean_(optimizer._flat_g, wire=wire or default_wire("fp32"))


class OverlappedGradSync:
    """Gradient exchange overlapped with the weight-gradient GEMMs (the only part of backward that is off the critical
    path, and here deliberately run last, see wgrad.py):

        backward (dgrad chain, queue filled)  ->  pack autograd grads into the arena
        -> for g in launch groups (arena order):   grouped wgrad g on the compute stream
                                                    all-reduce(arena range g) on the communication stream
        -> tail range (biases / LayerNorm / whatever the last group touched) -> join -> clip + AdamW

    so that RCCL moves range g over xGMI while group g+1 multiplies.  Works eagerly (the queue's end-of-backward
    callback runs the whole sequence) and with graph.GraphedTrainStep (forward/backward/pack graph, then this sequence
    launched eagerly from a stored plan, then the update graph).  Use as the `grad_sync` callable.

    Tried and dropped: capturing the groups into the step graph with progress flags (a one-lane kernel
    on the communication stream polling a word the graph sets after each group) and a single graph whose optimizer part
    polls an "exchange done" word.  Both work when the polling kernel and the kernels it waits for sit on different
    hardware queues, and stall until the poll times out when HIP maps the two streams onto the same queue (4 hardware
    queues for 6 streams here) -- not something to ship to an 8-GPU job."""

    overlapped = True

    def __init__(self, optimizer, n_groups: int = 4, wire: str = "fp32"):
        from . import wgrad
        self.opt, self.n_groups, self.wire = optimizer, n_groups, wire
        self.comm = torch.cuda.Stream()
        self.lanes = [torch.cuda.Stream() for _ in range(int(os.environ.get("HAMT_SYNC_LANES", 2)))]
        self.alternate = os.environ.get("HAMT_SYNC_ONE_LANE") is None
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
            ps = wgrad._Pass()
            ps.items = list(items)
            wgrad._flush_pass(ps, None)
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
        """Launch the plan's groups with the arena all-reduces on the communication stream.  Consecutive groups go to two
        alternating compute streams: group g+1's tiles fill the CUs that group g's last round leaves idle (a quarter of
        the tiles per launch is ~1.5 rounds of the chip: run back to back the four launches cost 40 % more than one),
        while range g's all-reduce still starts as soon as group g itself has finished."""
        from . import wgrad
        flat = self.opt._flat_g
        main = torch.cuda.current_stream()
        pending = sorted(plan.ranges, key=lambda r: r[2])
        k = 0

        done = []

        def reduce_ready(after):
            nonlocal k
            while k < len(pending) and pending[k][2] <= after:
                lo, hi, _, touched = pending[k]
                if touched:
                    for g in touched:          # every group that writes into the range (they may sit on both lanes)
                        self.comm.wait_event(done[g])
                else:
                    self.comm.wait_stream(main)
                with torch.cuda.stream(self.comm):
                    allreduce_mean_(flat[lo:hi], wire=self.wire)
                k += 1

        reduce_ready(-1)
        lanes = self.lanes if self.alternate else [main]
        for ln in lanes:
            if ln is not main:
                ln.wait_stream(main)
        for g in range(len(plan.groups)):
            ln = lanes[g % len(lanes)]
            for d in plan.deps[g]:                 # a buffer written by group d and accumulated into by group g
                ln.wait_event(done[d])
            with torch.cuda.stream(ln):
                wgrad.launch_group(plan, g)
            done.append(ln.record_event())
            reduce_ready(g)
        for ln in lanes:
            if ln is not main:
                main.wait_stream(ln)
        main.wait_stream(self.comm)

    def __call__(self, optimizer):
        """After backward: nothing left to do when the flush already exchanged; else (no queued GEMM gradients in this
        pass, e.g. fp32 mode) the plain flat all-reduce."""
        if self.done:
            self.done = False
            return
        allreduce_grads(optimizer, self.wire)


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
