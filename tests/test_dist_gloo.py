"""CPU, world_size 2 over gloo: the data-parallel plumbing of the N>1 path (vln_hamt_amd/parallel.py).

The HIP kernels cannot run here, so the model under DDP is the CPU oracle (test infrastructure) with the product's
parameter names; what is checked is the part that is identical on the GPU box: same task on every rank without a
broadcast, different data per rank, DDP(find_unused_parameters=True) averaging == single-process gradients on the
concatenated batch (unused heads included), and the timing/accounting reductions bench.py uses.
"""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class OracleModule(torch.nn.Module):
    """nn.Module shell around the functional oracle so that DDP can wrap it."""

    def __init__(self, sd, cfg):
        super().__init__()
        self.names = [k for k in sd if k != "mlm_head.predictions.decoder.weight"]
        self.params = torch.nn.ParameterList([torch.nn.Parameter(sd[k].clone()) for k in self.names])
        self.cfg = cfg

    def forward(self, batch, task):
        from oracle.hamt_oracle import HamtOracle
        return HamtOracle(dict(zip(self.names, self.params)), self.cfg).forward(batch, task, True)


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from _util import tiny_cfg
    from oracle.hamt_oracle import make_state_dict, pretrain_param_shapes
    from vln_hamt_amd.parallel import (TaskSchedule, allreduce_mean_, barrier, init_distributed, max_over_ranks, sum_over_ranks,
                                       wrap_ddp)
    from vln_hamt_amd.synth import make_batch
    r, lr, w = init_distributed(backend="gloo")
    assert (r, w) == (rank, world) and dist.get_backend() == "gloo"
    # 1) same task on every rank for 40 steps without any collective
    sched = TaskSchedule(cyclic=False, seed=7)
    tasks = [sched.task_at(s) for s in range(40)]
    gathered = [None] * world
    dist.all_gather_object(gathered, tasks)
    assert all(g == tasks for g in gathered)
    assert TaskSchedule(cyclic=True).cycle.count("mlm") == 5 and len(TaskSchedule(cyclic=True).cycle) == 12
    # 2) DDP gradient averaging == single-process gradient of the concatenated batch (SAP; other heads unused)
    cfg = tiny_cfg()
    sd = make_state_dict(pretrain_param_shapes(cfg), seed=7)
    model = OracleModule(sd, cfg)
    ddp = wrap_ddp(model, lr)
    batch = make_batch("sap", 2, cfg, seed=100 + rank, txt_len=20, hist_len=4)      # per-rank data (seed + rank)
    ddp(batch, "sap").mean().backward()
    grads = {k: (p.grad.clone() if p.grad is not None else None) for k, p in zip(model.names, model.params)}
    if rank == 0:
        ref = OracleModule(sd, cfg)
        bs = [make_batch("sap", 2, cfg, seed=100 + i, txt_len=20, hist_len=4) for i in range(world)]
        cat = {k: torch.cat([b[k] for b in bs], 0) for k in bs[0] if torch.is_tensor(bs[0][k])}
        ref(cat, "sap").mean().backward()
        worst = 0.0
        gmax = max(float(p.grad.abs().max()) for p in ref.params if p.grad is not None)
        for k, p in zip(ref.names, ref.params):
            if p.grad is None:
                assert grads[k] is None or float(grads[k].abs().max()) == 0.0, k       # unused heads: no / zero grad
            else:   # relative to the global gradient scale (some gradients are exactly 0 in exact arithmetic)
                worst = max(worst, float((grads[k] - p.grad).abs().max()) / gmax)
        assert worst < 1e-5, worst
    # 2b) the product's exchange: per-rank gradients laid out in ONE flat arena (zeros for unused heads, 8-element
    #     aligned slots like optim.AdamW), chunked all-reduce average == DDP's result == the concatenated-batch gradient
    local = OracleModule(sd, cfg)
    local(batch, "sap").mean().backward()
    offs, n = [], 0
    for p in local.params:
        offs.append(n)
        n += (p.numel() + 7) // 8 * 8
    flat = torch.zeros(n)
    for p, o in zip(local.params, offs):
        if p.grad is not None:
            flat[o:o + p.numel()] = p.grad.reshape(-1)
    allreduce_mean_(flat, chunk_elems=100_003)            # several ragged chunks
    gscale = max(float(g.abs().max()) for g in grads.values() if g is not None)
    for k, p, o in zip(local.names, local.params, offs):
        got = flat[o:o + p.numel()].view(p.shape)
        want = grads[k] if grads[k] is not None else torch.zeros_like(got)
        assert float((got - want).abs().max()) <= 1e-6 * gscale, k
    # 2c) bf16 wire format (the DDP bf16_compress_hook arithmetic): same averages to bf16 resolution, on a view that
    #     starts at an odd arena offset
    flat_b = torch.zeros(n + 3)
    for p, o in zip(local.params, offs):
        if p.grad is not None:
            flat_b[3 + o:3 + o + p.numel()] = p.grad.reshape(-1)
    allreduce_mean_(flat_b[3:], wire="bf16")
    assert float(flat_b[:3].abs().max()) == 0.0
    err = float((flat_b[3:] - flat).abs().max())
    assert 0.0 < err <= 2.0 ** -7 * gscale, err
    # 3) reductions used by bench.py
    assert max_over_ranks(float(rank + 1), "cpu") == float(world)
    assert sum_over_ranks(2.0, "cpu") == 2.0 * world
    barrier()
    open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    dist.destroy_process_group()


def test_data_parallel_plumbing_world2(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(os.path.join(str(tmp_path), f"ok{r}")) for r in range(world))


def test_wgrad_plan_invariants_cpu():
    import numpy as np
    """Host logic of the overlapped exchange (wgrad.build_plan, no kernel runs): for a fake backward pass over an arena
    with shared parameters -- (1) the ranges tile the arena exactly, (2) every group that writes into a range is in that
    range's `groups` and `after` is the last of them, (3) a buffer written twice is written by strictly increasing groups
    with the dependency recorded and the accumulate flag set on the later write, (4) work is split in arena order."""
    from types import SimpleNamespace
    from vln_hamt_amd import wgrad
    rng = np.random.Generator(np.random.PCG64(3))
    shapes = [(768, 768)] * 6 + [(3072, 768), (768, 3072), (2304, 768), (128, 768)]
    n = sum(int(np.prod(s)) + s[0] for s in shapes) + 4096
    flat_g = torch.zeros(n)
    params, off = [], 512                       # leading 512 elements: "non-GEMM" parameters (LayerNorm etc.)
    for s in shapes:
        w = torch.nn.Parameter(torch.zeros(s)); b = torch.nn.Parameter(torch.zeros(s[0]))
        w._hamt_grad_slot = flat_g[off:off + w.numel()].view(s); off += w.numel()
        b._hamt_grad_slot = flat_g[off:off + s[0]]; off += s[0]
        params.append((w, b))
    K = 256
    mk = lambda w: (torch.zeros(K, w.shape[0], dtype=torch.bfloat16), torch.zeros(K, w.shape[1], dtype=torch.bfloat16))
    items = []
    order = list(rng.permutation(len(params))) + [0, 0, 3]      # parameter 0 used three times, parameter 3 twice
    for i in order:
        w, b = params[i]
        dy, x = mk(w)
        items.append((w, b, dy, x))
    plan = wgrad.build_plan(items, SimpleNamespace(_flat_g=flat_g), n_groups=4)
    assert plan is not None and len(plan.groups) == len(plan.deps) == len(plan.tables)
    base = flat_g.data_ptr()
    # (1) ranges tile [0, n)
    rs = sorted(plan.ranges)
    assert rs[0][0] == 0 and rs[-1][1] == n and all(a[1] == b[0] for a, b in zip(rs[:-1], rs[1:]))
    # writers of every arena offset, per group
    writes = {}
    for g, (descs, cnt) in enumerate(plan.groups):
        for i in range(cnt):
            d = descs[i]
            for ptr, acc in ((d.dw, d.accum_dw), (d.db, d.accum_db)):
                if ptr:
                    writes.setdefault((ptr - base) // 4, []).append((g, acc))
    for lo, hi, after, groups in plan.ranges:
        touching = {g for o, ws in writes.items() if lo <= o < hi for g, _ in ws}
        assert touching == set(groups), (lo, hi, touching, groups)                     # (2)
        assert after == (max(groups) if groups else -1)
    for o, ws in writes.items():                                                        # (3)
        gs = [g for g, _ in ws]
        assert gs == sorted(gs) and len(set(gs)) == len(gs), (o, ws)
        assert [a for _, a in ws] == [0] + [1] * (len(ws) - 1), (o, ws)
        for g_prev, g_next in zip(gs[:-1], gs[1:]):
            assert g_prev in plan.deps[g_next]
    first = [min(o for o, ws in writes.items() if any(g == gg for g, _ in ws)) for gg in range(len(plan.groups))]
    assert len(plan.groups) >= 4 and first[:4] == sorted(first[:4])                     # (4)
    # (5) round 6: groups cut at the exchange's STATIC range boundaries (`cuts`): the weights written once whose slot starts in static
    # range k form one group, in arena order, and static range k is final (parallel.range_finality) once that group -- or, for a
    # weight that straddles the boundary, its predecessor -- has run; it never waits for a LATER range's group
    from vln_hamt_amd.parallel import range_finality
    for w, b in params:
        w.grad = b.grad = None
    once = [i for i in range(len(params)) if order.count(i) == 1]
    items1 = []
    for i in once:
        w, b = params[i]
        dy, x = mk(w)
        items1.append((w, b, dy, x))
    cuts = [n // 4 // 8 * 8, n // 2 // 8 * 8, 3 * n // 4 // 8 * 8]
    plan = wgrad.build_plan(items1, SimpleNamespace(_flat_g=flat_g), n_groups=2, cuts=cuts)
    static = list(zip([0] + cuts, cuts + [n]))
    import bisect
    g_of = {}
    for g, (descs, cnt) in enumerate(plan.groups):
        for i in range(cnt):
            g_of[(descs[i].dw - base) // 4] = g
    by_range = {}
    for o, g in g_of.items():
        by_range.setdefault(bisect.bisect_right(cuts, o), set()).add(g)
    assert all(len(v) == 1 for v in by_range.values()), by_range                         # one group per static range
    ks = sorted(by_range)
    assert [next(iter(by_range[k])) for k in ks] == list(range(len(ks))), by_range        # numbered in arena order, none empty
    fin = range_finality(static, plan.ranges)
    for k, (lo, hi, after, touched) in enumerate(fin):
        own = next(iter(by_range[k])) if k in by_range else -1
        assert after <= max(own, max((next(iter(by_range[j])) for j in by_range if j < k), default=-1)), (k, after, own)


def test_shard_cuts_invariants():
    """parallel.shard_cuts / range_finality: the STATIC ranges tile the arena, end on 8 * world granules, contain the bf16- / fp32-
    gathered region boundary, depend on nothing but (arena layout, world size) -- so the owner of an element never changes between
    steps -- and a static range is exchanged no earlier than every weight-gradient launch group that writes into it."""
    from vln_hamt_amd.parallel import range_finality, shard_cuts
    import random
    rnd = random.Random(0)
    for world in (1, 2, 4, 8):
        q = 8 * world
        for _ in range(50):
            n = q * rnd.randint(4, 400)
            n_a = q * rnd.randint(0, n // q)
            parts = rnd.randint(1, 9)
            cuts = shard_cuts(n, n_a, world, parts)
            assert cuts == shard_cuts(n, n_a, world, parts)
            assert cuts[0] == 0 and cuts[-1] == n and n_a in cuts and cuts == sorted(set(cuts))
            assert all(c % q == 0 for c in cuts) and len(cuts) <= parts + 2
            ranges = list(zip(cuts[:-1], cuts[1:]))
            owned = sum((hi - lo) // world for lo, hi in ranges)
            assert owned * world == n                          # every element has exactly one owner
            # a step's plan: arbitrary cut points (they differ from task to task), each plan range final after some launch group
            pb = [0] + sorted(rnd.sample(range(1, n), min(4, n - 1))) + [n]
            plan = [(lo, hi, rnd.randint(-1, 3), frozenset(rnd.sample(range(4), rnd.randint(0, 3)))) for lo, hi in zip(pb[:-1], pb[1:])]
            fin = range_finality(ranges, plan)
            assert [(lo, hi) for lo, hi, _, _ in fin] == ranges          # ownership is the plan's business in no way
            for lo, hi, after, touched in fin:
                for plo, phi, paf, pt in plan:
                    if plo < hi and phi > lo:
                        assert after >= paf and set(pt) <= touched


@pytest.mark.parametrize("n", [2, 8])
def test_bench_starts_its_own_ranks(n):
    """`python bench.py --gpus N` run plainly (no torchrun around it, no WORLD_SIZE in the environment) starts N ranks as a child
    `torch.distributed.run` before anything touches a GPU and exits with its return code (README.md:48-55 starts the reference with
    `torch.distributed.launch --nproc_per_node N`; VERDICT r5: the plain call used to run ONE rank and report n_gpus 1).
    `--launch-check`: rendezvous on 127.0.0.1 + one all-reduce + rank 0's JSON line, no model -- runs here over gloo."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(HAMT_DIST_BACKEND="gloo", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--launch-check"], env=env,
                       capture_output=True, text=True, timeout=300, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                      # ONE line, from rank 0
    out = json.loads(lines[0])
    assert out == {"launch_check": True, "n_gpus": n, "backend": "gloo"}, out


def test_bench_self_launch_propagates_failure():
    """a rank that fails makes the plain call fail with the launcher's code (here: --gpus 2 against a launcher told to start 3 ranks)"""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    env_keep = dict(os.environ)
    os.environ["HAMT_DIST_BACKEND"] = "gloo"
    try:
        rc = bench.self_launch(2, ["--gpus", "2", "--launch-check"],
                               launcher=[sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3",
                                         "--master-addr", "127.0.0.1", "--master-port", str(bench.free_port())])
    finally:
        os.environ.clear(); os.environ.update(env_keep)
    assert rc != 0
