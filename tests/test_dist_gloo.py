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
