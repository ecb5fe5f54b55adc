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
        if os.environ["MASTER_ADDR"] in ("127.0.0.1", "localhost"):      # one node: gloo must not try to resolve the host name (it may not resolve)
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
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
DRY = [False]      # measurement switch (bench.py): skip the collectives themselves (every rank keeps its own values) to time a step without them


def default_wire(prec: str = "bf16") -> str:
    w = os.environ.get("HAMT_GRAD_WIRE")
    if w is not None:
        assert w in ("fp32", "bf16"), w
        return w
    return "bf16" if prec == "bf16" else "fp32"


def _stage_for(flat: torch.Tensor, cache: dict | None = None) -> torch.Tensor:
    """bf16 mirror of the storage `flat` views (same element offsets), allocated once per arena.  `cache`: where the mirror is
    kept -- the gradient-sync object's own dict (OverlappedGradSync passes one); the module-level dict only serves the plain
    allreduce_grads() helper."""
    cache = _staging if cache is None else cache
    st = flat.untyped_storage()
    key = (st.data_ptr(), st.nbytes(), str(flat.device))
    buf = cache.get(key)
    if buf is None:
        buf = cache[key] = torch.empty(st.nbytes() // 4, dtype=torch.bfloat16, device=flat.device)
    o = flat.storage_offset()
    return buf[o:o + flat.numel()]


def allreduce_mean_(flat: torch.Tensor, chunk_elems: int = 32 << 20, wire: str = "fp32", stage_cache: dict | None = None) -> torch.Tensor:
    """In-place average of a flat fp32 tensor over all ranks, in `chunk_elems`-element all-reduces (128 MiB fp32 / 64 MiB
    bf16 messages)."""
    if not (dist.is_available() and dist.is_initialized()) or DRY[0]:
        return flat
    world = dist.get_world_size()
    if wire == "bf16":
        assert flat.dtype == torch.float32 and flat.is_contiguous()
        stage = _stage_for(flat, stage_cache)
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
    # the per-tile sums of squares the weight-gradient launch left (optim.AdamW.note_fused_sumsq) describe THIS rank's gradients
    # before the average; the collective rewrites the arena through raw pointers, which no version counter sees
    optimizer.clear_fused_sumsq()
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
        from . import streams
        dev = optimizer._flat_p.device if getattr(optimizer, "_flat_p", None) is not None else torch.cuda.current_device()
        self.comm = streams.role_stream(dev, "comm")      # (one stream per role and process: streams.role_stream)
        self.lanes = [streams.role_stream(dev, f"lane{i}") for i in range(int(os.environ.get("HAMT_SYNC_LANES", 2)))]
        self.alternate = os.environ.get("HAMT_SYNC_ONE_LANE") is None
        self._stage_cache: dict = {}        # this object's bf16 wire mirrors (no module-level state)
        self._ar_ranges = None             # static all-reduce ranges (see _static_ranges)
        self.mode = "eager"                # "eager": run at flush; "plan": only build the plan (graph capture)
        self.plan = None
        self.done = False
        wgrad.set_handler(self._on_flush)

    def close(self):
        from . import wgrad
        wgrad.set_handler(None)

    def _on_flush(self, items):
        from . import wgrad
        plan = wgrad.build_plan(items, self.opt, self.n_groups, wire=self._wire_spec(), cuts=self._group_cuts())
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

    def _group_cuts(self):
        """interior boundaries of the static exchange ranges inside the GEMM-weight region: the weight-gradient launch groups are cut
        THERE (wgrad.build_plan `cuts`), so that range k is final -- and its collective under way -- when group k has run.
        HAMT_GROUPS_EQUAL_WORK=1: `n_groups` equal-work groups as before round 6 (measurements)."""
        if os.environ.get("HAMT_GROUPS_EQUAL_WORK"):
            return None
        n_a = self.opt._n_shadow_only
        return [hi for (lo, hi) in self._static_ranges() if hi < n_a]

    def _wire_spec(self):
        """None, or (bf16 staging arena, scale, first element NOT covered) for weight-gradient launches that write the wire value
        themselves (wgrad.build_plan)."""
        return None

    @staticmethod
    def _launch_group(plan, g):
        """group g's launches on the current stream: the captured graph when graph.GraphedTrainStep made one, else eagerly"""
        from . import wgrad
        if plan.graphs is not None and plan.graphs[g] is not None:
            plan.graphs[g].replay()
        else:
            wgrad.launch_group(plan, g)

    def _static_ranges(self):
        """The arena ranges of the all-reduce exchange: a function of the arena layout only (`shard_cuts` with world = 1), like the
        sharded exchange's.  A plan's own range bounds follow the step's launch-group cuts, i.e. whatever parameters THIS rank
        queued -- collectives sized by them would not match between ranks."""
        if self._ar_ranges is None:
            o = self.opt.materialize()
            cuts = shard_cuts(o._n, o._n_shadow_only, 1, int(os.environ.get("HAMT_SHARD_PARTS", 4)))
            self._ar_ranges = [(lo, hi) for lo, hi in zip(cuts[:-1], cuts[1:]) if hi > lo]
        return self._ar_ranges

    def _exchange(self, lo, hi, plan, first):
        """one static range on the communication stream (the caller has made it wait for the range's producers)"""
        if self.log is not None:
            self.log.append((lo, hi))
        allreduce_mean_(self.opt._flat_g[lo:hi], wire=self.wire, stage_cache=self._stage_cache)

    log = None      # tests: a list that receives the (lo, hi) of every range exchange in issue order

    def run(self, plan, sumsq=False):
        """Launch the plan's groups with the arena exchanges on the communication stream.  Consecutive groups go to two
        alternating compute streams: group g+1's tiles fill the CUs that group g's last round leaves idle (a quarter of
        the tiles per launch is ~1.5 rounds of the chip: run back to back the four launches cost 40 % more than one),
        while a range's exchange still starts as soon as every group that writes into it has finished.

        The SEQUENCE of collectives is rank-invariant by construction: the static ranges (arena layout and world size only) in
        arena order, every step, whatever the task, the batch shape or the parameters this rank happened to queue.  The plan
        (rank-local: its launch groups follow what was queued) only decides which events a range's exchange waits for and after
        which launch the host enqueues it -- a range is enqueued once its predecessors in arena order are and every group that
        touches it has been launched.  (ordering the ranges by the plan's `after` let two ranks with different
        padded row counts reduce-scatter different ranges against each other.)"""
        from . import wgrad
        main = torch.cuda.current_stream()
        rng = getattr(plan, "_static_rng", None)
        if rng is None or rng[0] is not self:
            rng = plan._static_rng = (self, range_finality(self._static_ranges(), plan.ranges))
        pending = rng[1]                   # arena order
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
                    self._exchange(lo, hi, plan, (k == 0) if sumsq else None)
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
                self._launch_group(plan, g)
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


def shard_cuts(n: int, n_a: int, world: int, parts: int = 8):
    """STATIC range boundaries of the sharded exchange: the GEMM-weight region [0, n_a) (all-gathered as bf16) in `parts` ranges of
    equal size rounded down to multiples of 8 * world, the fp32-read region [n_a, n) as one range.  Returns sorted unique cut
    points including 0 and n; every range splits into `world` equal chunks of whole 8-element granules, rank r owns chunk r.
    The cuts depend on nothing but the arena layout and the world size: ownership of an element -- and with it the only valid
    copy of its exp_avg / exp_avg_sq / fp32 master -- never moves between steps, tasks or batch shapes (a first version cut at
    each step's weight-gradient group boundaries, which differ from task to task: elements changed owner and the new owner
    applied AdamW to stale moments)."""
    q = 8 * world
    assert n % q == 0 and n_a % q == 0, (n, n_a, world)
    cuts = {0, n, n_a}
    for k in range(1, max(1, parts)):
        cuts.add((n_a * k // parts) // q * q)
    return sorted(c for c in cuts if 0 <= c <= n)


def range_finality(static_ranges, plan_ranges):
    """[(lo, hi, after_group, groups)] for the static ranges of the sharded exchange under one step's weight-gradient plan
    (`wgrad.Plan.ranges`, same tuple layout): a static range is final once every plan range that overlaps it is."""
    out = []
    for lo, hi in static_ranges:
        touched, after = set(), -1
        for (plo, phi, paf, pt) in plan_ranges:
            if plo < hi and phi > lo:
                touched |= set(pt)
                after = max(after, paf)
        out.append((lo, hi, after, touched))
    return out


def shardable(optimizer, world: int) -> bool:
    """Can the arenas be split into 8-element granules over `world` ranks?  (Region ends are multiples of optim.adamw.REGION_ALIGN =
    512 elements: true whenever 8 * world divides 512, i.e. world in {1, 2, 4, 8, 16, 32, 64}.)"""
    o = optimizer.materialize()
    return o._n % (8 * world) == 0 and o._n_shadow_only % (8 * world) == 0


def make_grad_sync(optimizer, prec: str = "bf16", n_groups: int = 4, wire: str | None = None, sharded: bool | None = None):
    """The gradient exchange for this process group: ShardedGradSync (reduce-scatter, owned-slice AdamW, all-gather) in bf16 mode
    when the arenas split evenly over the ranks, else OverlappedGradSync (all-reduce, full AdamW on every rank)."""
    wire = wire or default_wire(prec)
    world = dist.get_world_size() if dist.is_initialized() else 1
    if sharded is None:
        sharded = os.environ.get("HAMT_SHARDED", "1") != "0" and prec == "bf16"
    if sharded and not shardable(optimizer, world):
        import warnings
        warnings.warn(f"vln_hamt_amd.parallel: 8 x world size ({world}) does not divide the arena regions (multiples of 512 elements); "
                      "using the all-reduce exchange (OverlappedGradSync) instead of the sharded one")
        sharded = False
    return (ShardedGradSync if sharded else OverlappedGradSync)(optimizer, n_groups=n_groups, wire=wire)


class ShardedGradSync(OverlappedGradSync):
    """ZeRO-1 style step for the flat arenas: instead of all-reducing the gradient arena and running the
    34-byte-per-parameter AdamW pass over all 174.8 M parameters on EVERY rank,

        grouped wgrad g  ||  reduce-scatter(arena range g-1)        (as before, range by range behind the weight-gradient GEMMs)
        -> each rank: sum of squares of the slices it owns -> ONE 8-byte all-reduce = the global norm
        -> clip + AdamW over the owned slices only (1/world of the update traffic)
        -> all-gather: the bf16 shadow for the GEMM-weight region (what forward / backward read), fp32 for the rest

    Rank r owns chunk r of every range of a STATIC partition of the arena (`shard_cuts`: fixed at construction, the same for every
    task / batch shape / step, so exp_avg, exp_avg_sq and the fp32 master of an element live on one rank for the whole run; a
    range is exchanged as soon as every weight-gradient launch group of the step that writes into it is done).  Bytes per GPU on
    the wire with the bf16 format: (world-1)/world x (2 B/param reduce-scatter + 2 B/param all-gather of the weights + 4 B/param
    of the ~14 % fp32-read parameters) -- about the bf16 all-reduce's -- while the update's HBM traffic drops by `world`.
    Gradient slots of parameters without a gradient hold zeros on every rank (DDP find_unused_parameters=True semantics,
    utils/misc.py:57-58): a range is zeroed right after its reduce-scatter and only the owned, reduced chunk is written back.
    fp32 masters of GEMM weights and the moments of everything this rank does not own are NOT kept current (nobody reads them):
    `gather_state()` on EVERY rank before `state_dict()` / checkpointing (the optimizer's state_dict refuses otherwise)."""

    sharded = True

    def __init__(self, optimizer, n_groups: int = 4, wire: str = "fp32"):
        super().__init__(optimizer, n_groups, wire)
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.gloo = dist.is_initialized() and dist.get_backend() != "nccl"
        o = optimizer.materialize()
        if not shardable(o, self.world):
            raise ValueError(f"ShardedGradSync: 8 x world size ({self.world}) must divide the arena regions ({o._n_shadow_only}, {o._n} elements); "
                             "use parallel.make_grad_sync, which falls back to OverlappedGradSync")
        # (parts: 2 / 4 / 8 measured 10.99 / 10.33 / 10.57 ms per step through a one-rank group, 9.77 plain: fewer ranges = fewer
        # serial copy / widen / norm hops behind the last weight-gradient group, more = an earlier start under the first groups)
        cuts = shard_cuts(o._n, o._n_shadow_only, self.world, int(os.environ.get("HAMT_SHARD_PARTS", 4)))
        self._ranges = [(lo, hi) for lo, hi in zip(cuts[:-1], cuts[1:]) if hi > lo]      # static: see shard_cuts
        o._sharded_sync = self
        o._shard_stale = False             # True: moments / masters of foreign chunks are out of date (gather_state() clears)
        # bf16 staging arena of the wire format (same element offsets as the gradient arena).  Zero-initialised: slots of parameters
        # without a gradient are exchanged as they are (the update skips such parameters), so they must at least be finite
        self._stage = torch.zeros(o._n, dtype=torch.bfloat16, device=o._flat_g.device) if wire == "bf16" else None
        self.direct_wire = wire == "bf16" and os.environ.get("HAMT_NO_DIRECT_WIRE") is None
        self._own16 = torch.empty(o._n // self.world + 8, dtype=torch.bfloat16, device=o._flat_g.device) if wire == "bf16" else None
        self._own32 = torch.empty(o._n // self.world + 8, dtype=torch.float32, device=o._flat_g.device)
        self._gsq = torch.zeros(1, dtype=torch.float32, device=o._flat_g.device)
        self._upd = None                   # (norm graph, update graph, max_norm) once graph.GraphedTrainStep captured them
        self._unconsumed = False           # an exchange has run and update() has not consumed it yet
        self._norm_reduced = False         # global_norm() ran for the pending exchange

    def close(self):
        super().close()
        if getattr(self.opt, "_sharded_sync", None) is self:
            self.opt._sharded_sync = None
        self.detach_gather()

    def attach_gather(self, model: torch.nn.Module):
        """Gate `model`'s next forward pass range by range behind the parameters' all-gathers of update() instead of making the
        stream wait for all of them: forward pre-hooks on every module (optim.adamw._Overlap: the machinery of AdamW.attach, with
        the static exchange ranges as its chunks and the communication stream as its stream).  Eager loops only; under graph replay
        the all-gathers sit between the replays on the launching stream, as before."""
        from .optim.adamw import _Overlap
        o = self.opt
        self.detach_gather()
        n_a = o._n_shadow_only
        tail = [(lo, hi - lo) for lo, hi in self._ranges if hi > n_a]
        chunks = tail + [(lo, hi - lo) for lo, hi in self._ranges if hi <= n_a]
        o._gather_ov = _Overlap(o, model, 0, chunks=chunks, stream=self.comm)
        return self

    def detach_gather(self):
        gov = getattr(self.opt, "_gather_ov", None)
        if gov is not None:
            self.opt.wait_update()
            gov.remove()
            self.opt._gather_ov = None

    def _wire_spec(self):
        if not self.direct_wire:
            return None
        return (self._stage, 1.0 / self.world, self.opt._n_shadow_only)

    # ---- collectives (gloo has neither reduce_scatter nor all_gather_into_tensor: same arithmetic through all_reduce / all_gather)
    def _reduce_scatter(self, out, inp):
        if not dist.is_initialized() or DRY[0]:
            c = inp.numel() // self.world
            out.copy_(inp[self.rank * c:(self.rank + 1) * c])
        elif self.gloo:
            dist.all_reduce(inp, op=dist.ReduceOp.SUM)
            c = inp.numel() // self.world
            out.copy_(inp[self.rank * c:(self.rank + 1) * c])
        else:
            dist.reduce_scatter_tensor(out, inp, op=dist.ReduceOp.SUM)

    def _all_gather_inplace(self, buf):
        """buf [world * c]: chunk `rank` is current; fill in the others"""
        if not dist.is_initialized() or DRY[0]:
            return
        c = buf.numel() // self.world
        if self.gloo:
            parts = [torch.empty_like(buf[:c]) for _ in range(self.world)]
            dist.all_gather(parts, buf[self.rank * c:(self.rank + 1) * c].contiguous())
            for r, t in enumerate(parts):
                if r != self.rank:
                    buf[r * c:(r + 1) * c].copy_(t)
        else:
            dist.all_gather_into_tensor(buf, buf[self.rank * c:(self.rank + 1) * c])

    def _exchange_range(self, lo, hi, plan=None, sumsq=None):
        """flat_g[lo:hi] -> mean over ranks of chunk `rank`, written back in place.  bf16 wire with a plan whose launches wrote
        their wire values themselves (`plan.pack_segs`, GEMM-weight region): only the listed fp32 segments are packed, and the
        rest of the range is NOT zeroed -- nothing reads a gradient slot of that region outside the owned chunk (the weight-gradient
        launches store, the update and the norm walk the owned segments).  Else: pack the range, zero it, write the owned chunk."""
        from . import _lib as L
        from .ops import _p, _stream
        g = self.opt._flat_g[lo:hi]
        c = (hi - lo) // self.world
        if self.wire == "bf16":
            st, own = self._stage[lo:hi], self._own16[:c]
            lib = L.load()
            direct = plan is not None and plan.pack_segs is not None and hi <= self.opt._n_shadow_only
            if direct:
                for (so, sc) in plan.pack_segs:
                    a, b = max(so, lo), min(so + sc, hi)
                    if b > a:
                        L.check(lib.hamt_wire_pack_bf16(b - a, _p(self.opt._flat_g[a:b]), _p(self._stage[a:b]), 1.0 / self.world, _stream()), "hamt_wire_pack_bf16")
            else:
                L.check(lib.hamt_wire_pack_bf16(hi - lo, _p(g), _p(st), 1.0 / self.world, _stream()), "hamt_wire_pack_bf16")
            if self.world == 1 and not self.gloo:
                own = st                       # one rank: the reduce-scatter is the identity -- no staging copy of the whole range
            else:
                self._reduce_scatter(own, st)
            if not direct:
                g.zero_()
            if sumsq is not None and c:      # widen the owned chunk and add its share of the global norm in the same pass
                o, f = self.opt, lo + self.rank * c
                L.check(lib.hamt_wire_unpack_sumsq(f, c, _p(own), _p(o._flat_g[f:f + c]), _p(o._ends), _p(o._hyp), len(o._params), _p(self._gsq),
                                                   int(not sumsq), _p(o._ws), _stream()), "hamt_wire_unpack_sumsq")
                return
            L.check(lib.hamt_wire_unpack_bf16(c, _p(own), _p(g[self.rank * c:(self.rank + 1) * c]), _stream()), "hamt_wire_unpack_bf16")
        else:
            own = self._own32[:c]
            self._reduce_scatter(own, g)
            g.zero_()
            torch.mul(own, 1.0 / self.world, out=g[self.rank * c:(self.rank + 1) * c])
        if sumsq is not None and c:
            # this range's share of the global gradient norm, right behind its exchange on the communication stream (sumsq = is this
            # the first range of the step?): hidden under the weight-gradient launches instead of a serial run of 2 x 9 small
            # launches in front of the update.  Needs the step's hyper-parameter table (which parameters are active) on the device
            # already: graph.GraphedTrainStep uploads it before the replay.
            o = self.opt
            f = lo + self.rank * c
            L.check(L.load().hamt_sumsq_table(f, c, _p(o._flat_g[f:f + c]), _p(o._ends), _p(o._hyp), len(o._params), _p(self._gsq), int(not sumsq),
                                              _p(o._ws), _stream()), "hamt_sumsq_table")

    def _static_ranges(self):
        return self._ranges

    def _exchange(self, lo, hi, plan, first):
        """(parent.run) first: None = no norm share; True / False = this is / is not the step's first range (see _exchange_range)"""
        if self.log is not None:
            self.log.append((lo, hi))
        self._exchange_range(lo, hi, plan, first)

    def _once_per_update(self):
        """The sharded exchange runs ONCE per optimizer step: afterwards a range holds the rank mean in the owned chunk only (zeros, or
        with the direct bf16 wire stale local values, elsewhere), so a second micro-batch's gradients (gradient_accumulation_steps >
        1, main_r2r.py:244-251) would be added to, and exchanged on top of, something that is no longer this rank's local sum.  The
        reference's R2R / RxR configurations accumulate over 1 step (pretrain_r2r.json:11); anything else must accumulate locally
        and exchange on the last micro-batch -- refuse rather than drop gradients silently.  One rank is exempt: its
        owned chunk is the whole range, so accumulation is exact there."""
        if DRY[0]:
            return                         # (measurement passes with the collectives switched off: bench.py's probes)
        if self._unconsumed and self.world > 1:
            from ._lib import HamtError
            raise HamtError("ShardedGradSync: a second gradient exchange before update() consumed the first (gradient accumulation over "
                            "several backward passes): not supported by the sharded exchange -- accumulate locally with the exchange "
                            "detached and attach it for the last micro-batch, or use OverlappedGradSync (HAMT_SHARDED=0)")
        self._unconsumed = True

    def discard(self):
        """Forget a pending exchange whose update will not run (a loop that skips optimizer.step() for this pass: NaN guard, early
        break -- and then calls optimizer.zero_grad(), which calls this): the next backward pass may exchange again, and a norm
        reduced for the skipped pass is not mistaken for the next one's."""
        self._unconsumed = False
        self._norm_reduced = False
        self.done = False

    def run(self, plan, sumsq=False):
        self._once_per_update()
        super().run(plan, sumsq)

    def __call__(self, optimizer):
        if self.done:
            self.done = False
            return
        # no queued GEMM gradients in this pass (fp32 mode / nothing deferred): exchange the arena range by range, same ownership
        self._once_per_update()
        if not optimizer._packed:
            optimizer._pack_grads()
        for lo, hi in self._ranges:
            self._exchange_range(lo, hi)

    def owned(self):
        """[(first, count)] arena segments this rank updates"""
        out = []
        for lo, hi in self._ranges:
            c = (hi - lo) // self.world
            if c:
                out.append((lo + self.rank * c, c))
        return out

    def _launch_sumsq(self):
        """sum of squares of the owned gradient chunks -> self._gsq (active parameters only: see optim.AdamW._keep)"""
        from . import _lib as L
        from .ops import _p, _stream
        o, lib = self.opt, L.load()
        for i, (f, c) in enumerate(self.owned()):
            L.check(lib.hamt_sumsq_table(f, c, _p(o._flat_g[f:f + c]), _p(o._ends), _p(o._hyp), len(o._params), _p(self._gsq), int(i > 0),
                                         _p(o._ws), _stream()), "hamt_sumsq_table")

    def _launch_adamw(self, max_norm: float):
        """clip + AdamW over the owned chunks"""
        from . import _lib as L
        from .ops import _p, _stream
        o, lib = self.opt, L.load()
        b1, b2 = o.param_groups[0]["betas"]
        for f, c in self.owned():
            L.check(lib.hamt_adamw_table_range(f, c, _p(o._flat_p[f:f + c]), _p(o._flat_g[f:f + c]), _p(o._flat_m[f:f + c]), _p(o._flat_v[f:f + c]),
                                               _p(o._flat_p16[f:f + c]), _p(o._ends), _p(o._hyp), len(o._params), _p(self._gsq), float(max_norm),
                                               b1, b2, o.param_groups[0]["eps"], 1, _stream()), "hamt_adamw_table_range")

    def capture_update(self, max_norm: float, pool, stream, mode):
        """graph.GraphedTrainStep: the two launch runs of `update` (norm partials; clip + AdamW) as captured graphs -- 2 x 9 launches
        the host no longer issues one by one between the collectives of every step.  Everything they read is static (arena
        pointers, owned segments, the device hyper-parameter table)."""
        if self._upd is not None and self._upd[2] == float(max_norm):
            return
        g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(g1, pool=pool, stream=stream, capture_error_mode=mode):
            self._launch_sumsq()
        with torch.cuda.graph(g2, pool=pool, stream=stream, capture_error_mode=mode):
            self._launch_adamw(max_norm)
        self._upd = (g1, g2, float(max_norm))

    @torch.no_grad()
    def global_norm(self) -> torch.Tensor:
        """The global gradient norm after the exchange (device scalar, the same on every rank): this rank's sum of squares over the
        chunks it owns, one 4-byte all-reduce.  What `clip_grad_norm_` returns in a loop that keeps the reference's lines
        (ArenaDataParallel); `update(..., norm_reduced=True)` then uses it as it is."""
        self.opt._ensure_table()
        self._launch_sumsq()
        if dist.is_initialized() and not DRY[0]:
            dist.all_reduce(self._gsq, op=dist.ReduceOp.SUM)
        self._norm_reduced = True
        return self._gsq.sqrt()[0]

    @torch.no_grad()
    def update(self, max_norm: float, have_sumsq: bool = False, norm_reduced: bool = False):
        """Global-norm clip + AdamW over the owned segments, then the all-gathers.  The host part (optimizer.prepare_step) must
        have run; replaces clip_grad_norm_ + optimizer.step() + zero_grad() of the unsharded loop.  have_sumsq: the exchange
        (`run(plan, sumsq=True)`) already left this rank's sum of squares in self._gsq; norm_reduced: `global_norm()` already
        reduced it over the ranks."""
        from . import _lib as L
        from .ops import _p, _stream
        o, lib = self.opt, L.load()
        o.wait_update()                    # (all-gathers of the previous update still on the communication stream: attach_gather)
        graphs = self._upd if (self._upd is not None and self._upd[2] == float(max_norm)) else None
        if have_sumsq or norm_reduced:
            pass
        elif graphs is not None:
            graphs[0].replay()
        else:
            self._launch_sumsq()
        if dist.is_initialized() and not DRY[0] and not norm_reduced:
            dist.all_reduce(self._gsq, op=dist.ReduceOp.SUM)          # 4 bytes: the global squared norm
        self._norm_reduced = False
        if graphs is not None:
            graphs[1].replay()
        else:
            self._launch_adamw(max_norm)
        n_a = o._n_shadow_only
        gov = o._gather_ov
        if gov is not None and not torch.cuda.is_current_stream_capturing() and os.environ.get("HAMT_NO_GATHER_OVERLAP") is None:
            # eager loop (parallel.ArenaDataParallel): the parameters' all-gathers run on the communication stream in the order a forward
            # pass reads them -- the fp32-read region (embedding tables, biases, LayerNorm: the first kernels), then the GEMM-weight
            # ranges in arena (= registration = use) order -- one event per range; the NEXT forward starts right away and every module
            # waits for the range(s) that hold its parameters (attach_gather: optim.AdamW's pre-hook gates), instead of the whole
            # forward waiting for ~350 MB of all-gather at world 8 (DDP overlaps the same way: utils/misc.py:57-58 + main_r2r.py:246)
            cur = torch.cuda.current_stream()
            self.comm.wait_stream(cur)
            with torch.cuda.stream(self.comm):
                for i, (f, c) in enumerate(gov.chunks):
                    if f + c <= n_a:
                        self._all_gather_inplace(o._flat_p16[f:f + c])
                    else:
                        self._all_gather_inplace(o._flat_p[f:f + c])
                        L.check(lib.hamt_cast_f32_bf16(o._n - n_a, _p(o._flat_p[n_a:]), _p(o._flat_p16[n_a:]), _stream()), "hamt_cast_f32_bf16")
                    gov.events[i].record(self.comm)
            gov.mark_pending()
        else:
            for lo, hi in self._ranges:
                if hi <= n_a:
                    self._all_gather_inplace(o._flat_p16[lo:hi])         # GEMM weights: only their bf16 image is ever read
                else:
                    self._all_gather_inplace(o._flat_p[lo:hi])
            if o._n > n_a:                                                # bf16 images of the >= 2-D fp32-read parameters (tied MLM decoder ...)
                L.check(lib.hamt_cast_f32_bf16(o._n - n_a, _p(o._flat_p[n_a:]), _p(o._flat_p16[n_a:]), _stream()), "hamt_cast_f32_bf16")
        o._counted = None                                             # the counted step is applied (optim.AdamW.zero_grad)
        o.mark_updated()
        self._unconsumed = False
        o._shard_stale = self.world > 1

    @torch.no_grad()
    def gather_state(self):
        """Collective (call on EVERY rank, e.g. right before rank 0 saves -- utils/save.py:42-45 writes model and
        optimizer.state_dict() from one rank): make the fp32 masters of the GEMM-weight region and exp_avg / exp_avg_sq of the
        whole arena current on every rank."""
        o = self.opt
        o.wait_update()
        for lo, hi in self._ranges:
            if hi <= o._n_shadow_only:
                self._all_gather_inplace(o._flat_p[lo:hi])
            self._all_gather_inplace(o._flat_m[lo:hi])
            self._all_gather_inplace(o._flat_v[lo:hi])
        o._shard_stale = False

    gather_masters = gather_state          # (round-2 name)


class ArenaDataParallel(torch.nn.Module):
    """What `wrap_model` returns in place of the reference's `DistributedDataParallel(model, device_ids=[local_rank],
    find_unused_parameters=True)` (pretrain_src/utils/misc.py:52-65), so that main_r2r.py keeps its lines

        model = wrap_model(model, device, opts.local_rank)
        ...
        loss = model(batch, task=task, compute_loss=True); loss.mean().backward()
        grad_norm = clip_grad_norm_(model.parameters(), opts.grad_norm)      # vln_hamt_amd.optim's (the import swap)
        optimizer.step(); optimizer.zero_grad()

    in fp32 AND bf16 mode.  torch DDP cannot serve the bf16 path: its reducer is fed by per-parameter AccumulateGrad hooks, and the
    GEMM weight gradients never pass one -- they are written into the optimizer's flat gradient arena by ONE grouped launch at the
    end of backward (wgrad.py).  This wrapper does what DDP does, at that granularity:

      * construction: parameters and buffers of rank 0 are broadcast (DDP's _sync_module_states);
      * every forward arms an end-of-backward callback on its output (once per backward pass however many forwards fed it:
        the finetune rollout calls the module ~2T + 2 times before one backward, agent_cmt.py:584-597);
      * the callback runs the arena exchange of `make_grad_sync` -- reduce-scatter (or all-reduce) range by range behind the
        launch groups of the deferred weight gradients, i.e. overlapped with the only part of backward that is off the critical
        path -- created lazily the first time, when the optimizer that owns the parameters' arena is known (main_r2r.py builds it
        AFTER wrap_model, :150-156);
      * parameters without a gradient this step (the other tasks' heads) are exchanged as zeros and skipped by the update:
        find_unused_parameters=True semantics;
      * with the sharded exchange, `clip_grad_norm_` returns the global norm from the owned chunks + one 4-byte all-reduce, and
        `optimizer.step()` runs the owned-slice AdamW and the parameter all-gathers (optim.AdamW.step sees `_sharded_sync`).

    `.module` is the wrapped model (utils/save.py:36 `model.module if hasattr(model, 'module') else model`)."""

    def __init__(self, module: torch.nn.Module, n_groups: int | None = None, wire: str | None = None, sharded: bool | None = None,
                 accumulation_steps: int = 1):
        super().__init__()
        self.module = module
        # gradient accumulation (main_r2r.py:242-250: `loss / gradient_accumulation_steps`, optimizer.step() every k-th pass): the first
        # k - 1 backward passes of an update only SUM into the local gradient arena (the weight-gradient launch accumulates, no
        # collective), the k-th exchanges the sums -- one exchange per update, which is also what the sharded exchange requires.
        # Settable after construction (`model.accumulation_steps = opts.gradient_accumulation_steps`).
        self.accumulation_steps = max(1, int(accumulation_steps))
        self._micro = 0                # backward passes since the last update (re-set by optimizer.step() / zero_grad(): `_on_update`)
        self._exchanged = False        # ... and whether one of them exchanged
        self._cfg = (n_groups if n_groups is not None else int(os.environ.get("HAMT_SYNC_GROUPS", 4)), wire, sharded)
        self.grad_sync = None
        self._armed: set = set()
        if dist.is_available() and dist.is_initialized():
            with torch.no_grad():
                ts = [t for t in list(module.parameters()) + list(module.buffers()) if t.is_floating_point()]
                for dt in {t.dtype for t in ts}:
                    same = [t for t in ts if t.dtype == dt]
                    flat = torch.cat([t.detach().reshape(-1) for t in same])
                    dist.broadcast(flat, 0)
                    o = 0
                    for t in same:
                        t.copy_(flat[o:o + t.numel()].view(t.shape))
                        o += t.numel()
                opt = self._optimizer()
                if opt is not None and opt._built:
                    opt.refresh_shadow()

    def _optimizer(self):
        for p in self.module.parameters():
            r = getattr(p, "_hamt_opt", None)
            if r is not None and r() is not None:
                return r()
        return None

    def _sync_for(self, opt):
        if self.grad_sync is None or self.grad_sync.opt is not opt:
            if self.grad_sync is not None:
                self.grad_sync.close()
            from .modeling import precision_of
            cfg = getattr(self.module, "config", None)
            prec = precision_of(cfg) if cfg is not None else "bf16"
            n_groups, wire, sharded = self._cfg
            self.grad_sync = make_grad_sync(opt, prec, n_groups=n_groups, wire=wire, sharded=sharded)
            if getattr(self.grad_sync, "sharded", False) and os.environ.get("HAMT_NO_GATHER_OVERLAP") is None:
                self.grad_sync.attach_gather(self.module)       # the next forward runs under the parameters' all-gathers
        return self.grad_sync

    def forward(self, *args, **kwargs):
        out = self.module(*args, **kwargs)
        if torch.is_grad_enabled() and dist.is_available() and dist.is_initialized():
            # every tensor output that can start a backward pass arms the exchange (tuple / list / dict outputs, nested: a loss built
            # from the second output alone must exchange too -- torch DDP covers all outputs; `_armed` keeps it to once per pass)
            def walk(o):
                if torch.is_tensor(o):
                    if o.requires_grad:
                        o.register_hook(self._arm)
                elif isinstance(o, (tuple, list)):
                    for x in o:
                        walk(x)
                elif isinstance(o, dict):
                    for x in o.values():
                        walk(x)
            walk(out)
        return out

    def _arm(self, _grad):
        tid = torch._C._current_graph_task_id()
        if tid not in self._armed:
            self._armed.add(tid)
            torch.autograd.Variable._execution_engine.queue_callback(lambda: self._post_backward(tid))
        return None

    @torch.no_grad()
    def _post_backward(self, tid):
        self._armed.discard(tid)
        opt = self._optimizer()
        if opt is None:
            raise RuntimeError("ArenaDataParallel: the wrapped model's parameters belong to no vln_hamt_amd.optim.AdamW yet -- build the optimizer "
                               "(optim.misc.build_optimizer) before the first backward pass, as main_r2r.py does")
        opt.materialize()
        sync = self._sync_for(opt)
        if self not in opt._dp_wrappers:
            opt._dp_wrappers.add(self)          # (weak) optimizer.step() / zero_grad() call _on_update
        from . import wgrad
        dev = opt._flat_p.device
        # this callback was queued when the pass STARTED, i.e. in front of the weight-gradient queue's own end-of-pass flush: run that
        # flush now (it hands the queued problems to the exchange, which launches them group by group with the collectives behind
        # them); the queue's own callback then finds nothing left
        self._micro += 1
        q = wgrad.queue(dev)
        if self._micro % self.accumulation_steps:
            # not the update's last micro-batch: the queued weight gradients are launched into (accumulated into) the local arena, no
            # exchange.  (The exchange object stays the queue's handler for the pass that does exchange.)
            h, q.handler = q.handler, None
            try:
                q.flush(tid)
            finally:
                q.handler = h
            return
        q.flush(tid)
        sync(opt)                           # nothing queued in this pass (fp32 mode): the plain range-by-range exchange
        self._exchanged = True

    def _on_update(self, stepping: bool):
        """Called by the optimizer at the update boundary -- optimizer.step() (stepping) and zero_grad(): the private count of backward
        passes starts again there, whatever the caller's loop did in between (a second backward in one iteration, a micro-batch
        skipped before backward, an exception mid-update: ADVICE r5 -- the count used to run on and put the exchange on the wrong
        pass).  A step() over gradients that k > 0 backward passes produced and NO pass exchanged would update every rank from its
        own local sums, silently: refuse."""
        micro, exchanged = self._micro, self._exchanged
        if not stepping or exchanged or micro == 0:
            self._micro, self._exchanged = 0, False
        if stepping and micro and not exchanged and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            raise RuntimeError(f"ArenaDataParallel: optimizer.step() after {micro} backward pass(es) of which none exchanged gradients "
                               f"(accumulation_steps = {self.accumulation_steps}: the exchange runs on every {self.accumulation_steps}-th pass since "
                               "the last update) -- the ranks would update from their local sums and diverge.  Step on the pass that "
                               "exchanges, or set model.accumulation_steps to the loop's gradient_accumulation_steps")

    def close(self):
        if self.grad_sync is not None:
            self.grad_sync.close()
            self.grad_sync = None


def wrap_model(model: torch.nn.Module, device: torch.device, local_rank: int, gradient_accumulation_steps: int = 1) -> torch.nn.Module:
    """pretrain_src/utils/misc.py:52-65 with the same signature: `model.to(device)`; distributed (local_rank != -1) -> the
    arena-exchange wrapper instead of torch DDP (see ArenaDataParallel); a single process is returned as it is (the reference's
    nn.DataParallel branch for several visible GPUs in ONE process is not built: one process per GPU).
    `gradient_accumulation_steps` (optional, the one argument the reference's call does not have): opts.gradient_accumulation_steps when
    it is > 1 -- the exchange then runs on every k-th backward pass only, on the locally accumulated sums (ArenaDataParallel)."""
    model.to(device)
    if local_rank != -1 and dist.is_available() and dist.is_initialized():
        return ArenaDataParallel(model, accumulation_steps=gradient_accumulation_steps)
    return model


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
