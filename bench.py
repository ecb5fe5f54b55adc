#!/usr/bin/env python3
"""bench.py -- HAMT R2R proxy-task pretraining throughput on MI355X (panorama-steps/s).

One "step" = one full optimisation step of the hot path on one synthetic minibatch per GPU:
forward + backward (+ RCCL gradient all-reduce when N > 1) + global-norm clip + AdamW + zero_grad, dropout on,
on the R2R-canon model (H=768, 12 heads, 9 text + 2 pano + 4 cross-modal layers, vocab 30522, 174.8 M params),
36x768 view features, 80-token instructions, history 5, tasks cycled with the reference's 5:1:1:1:2:2 mix.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--prec bf16|fp32] [--task mix|sap|...]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` and `cpu_baseline` objects.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0      # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md (chip-level parameters)
PEAK_HBM_GBS = 8000.0          # HBM3E peak, same guide
H, FFN, L_TXT, T_HIST, V = 768, 3072, int(os.environ.get("HAMT_BENCH_L", 80)), 5, 36      # (HAMT_BENCH_L: what-if measurements only)


def layer_flops(S):            # 24*S*H^2 + 4*S^2*H
    return 24 * S * H * H + 4 * S * S * H


def xlayer_flops(Lq, Vn):
    return 32 * (Lq + Vn) * H * H + 8 * Lq * Vn * H + 4 * Lq * Lq * H + 4 * Vn * Vn * H


def trunk_fwd_flops(task, L=L_TXT, T=T_HIST):
    """forward FLOPs per ORIGINAL sample (SURVEY.md 8a formula; ITM runs the x-layers on 5x the batch)."""
    ob = task in ("sap", "sar", "sprel")
    f = 9 * layer_flops(L) + T * (2 * layer_flops(V) + 2 * V * H * H + 2 * V * 4 * H) + T * (2 * H * H + 2 * 4 * H)
    if ob:
        f += (V + 1) * (2 * H * H + 2 * 4 * H)
    vn = T + 1 + (V + 1 if ob else 0)
    f += (5 if task == "itm" else 1) * 4 * xlayer_flops(L, vn)
    if task == "mlm":
        f += 2 * 12 * H * 30522 + 2 * 12 * H * H
    return f


def dead_fwd_flops(task, L=L_TXT, T=T_HIST, n_masked=12):
    """forward FLOPs per ORIGINAL sample that the reference launches and nobody reads ("dead code in the last cross-modal layer"):
    the side of the last layer whose output the head does not read (28 S H^2 + 4 C H^2 + 4 S C H + 4 S^2 H for a side of S rows with a
    context of C rows) and, of a side read at R rows only, the feed-forward block of the other S - R rows (16 H^2 each).  The reference's
    autograd never runs the backward of any of it, so of SURVEY 8a's 3 x forward these FLOPs are counted three times and executed once
    there, zero times here."""
    ob = task in ("sap", "sar", "sprel")
    vn = T + 1 + (V + 1 if ob else 0)
    side = lambda S, C: 28 * S * H * H + 4 * C * H * H + 4 * S * C * H + 4 * S * S * H
    ffn = lambda rows: 16 * rows * H * H
    if task == "mlm":
        return side(vn, L) + ffn(L - n_masked)
    if task == "sar":
        return side(vn, L) + ffn(L - 1)
    if task in ("mrc", "sprel"):
        return side(L, vn)
    if task == "sap":
        return ffn(L - 1)
    if task == "itm":
        return 5 * (ffn(L - 1) + ffn(vn - 1))
    return 0


def box_probe(device):
    """What THIS box's memory system does on the simplest streaming pattern: boxes of the pool differ by ~20 % on the update kernel
    (0.94 - 1.14 ms for the same 5.2 GB: VERDICT r4 weak 11) and by ~4 % on the step; a device-to-device copy of 1 GiB (read 1 + write 1)
    and a read-only pass say how much of that is the box.  HIP events on the current stream, best of 5."""
    n = 1 << 30
    a = torch.empty(n, dtype=torch.uint8, device=device).fill_(1)
    b = torch.empty_like(a)
    f = a.view(torch.float32)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best_c, best_r = 1e9, 1e9
    for _ in range(5):
        e0.record(); b.copy_(a); e1.record(); e1.synchronize()
        best_c = min(best_c, e0.elapsed_time(e1))
        e0.record(); f.sum(); e1.record(); e1.synchronize()
        best_r = min(best_r, e0.elapsed_time(e1))
    del a, b, f
    return {"d2d_copy_gbs": round(2 * n / best_c / 1e6, 1), "read_gbs": round(n / best_r / 1e6, 1), "how": "1 GiB torch copy_ / float32 sum, best of 5, HIP events"}


def build_model(prec, device):
    from vln_hamt_amd.model.pretrain_cmt import MultiStepNavCMTPreTraining
    from vln_hamt_amd.modeling import HamtConfig
    cfg = HamtConfig(hamt_precision=prec, pretrain_tasks={"mlm", "sap", "sar", "sprel", "mrc", "itm"})
    torch.manual_seed(0)
    model = MultiStepNavCMTPreTraining(cfg).to(device)
    model.train()
    return model, cfg


def time_xattn_probe(batch, device, reps=20):
    """The cross-modal attention kernels (vilmodel.py:327-348 inside the x-layers) at the step's own shapes: text queries over
    the 6 history + 37 observation tokens and the reverse, 12 heads of 64, bf16, dropout 0.1, key mask.  They are HBM /
    latency bound (40 flop per byte: the MFMA roofline is out of reach by construction), so the bound is HBM:
    algorithmic bytes = Q + K + V + O (+ dO, dQ, dK, dV for backward) once each.  Timed as a hipGraph of `reps` launches
    (an eager loop would measure the host's launch rate)."""
    import ctypes as C
    from vln_hamt_amd import _lib as Lb, ops
    lib, p = Lb.load(), ops._p
    rng = ops.rng_state(device)
    res = {}
    st = torch.cuda.Stream()
    for name, Sq, Sk in (("text<-vision", L_TXT, T_HIST + 1 + V + 1), ("vision<-text", T_HIST + 1 + V + 1, L_TXT)):
        q = torch.randn(batch * Sq, H, device=device).to(torch.bfloat16)
        kv = torch.randn(batch * Sk, 2 * H, device=device).to(torch.bfloat16)
        k, v = kv[:, :H], kv[:, H:]
        o, do = torch.empty_like(q), torch.randn(batch * Sq, H, device=device).to(torch.bfloat16)
        dq, dkv = torch.empty_like(q), torch.empty_like(kv)
        lse = torch.empty(batch * 12 * Sq, device=device)
        mask = torch.zeros(batch, Sk, device=device)
        d = Lb.AttnDesc(batch, 12, Sq, Sk, 64, H, 2 * H, 2 * H, H, Lb.HAMT_BF16, Lb.HAMT_BF16, 0.125, 0.1, 7, Lb.PREC_BF16)

        def fwd():
            Lb.check(lib.hamt_attn_small_fwd(C.byref(d), p(q), p(k), p(v), p(mask), p(o), p(lse), p(rng), ops._stream()), "attn fwd")

        def bwd():
            Lb.check(lib.hamt_attn_small_bwd(C.byref(d), p(q), p(k), p(v), p(mask), p(o), p(do), p(lse), None, p(dq), p(dkv[:, :H]),
                                             p(dkv[:, H:]), p(rng), ops._stream()), "attn bwd")
        nb_f = 2.0 * (2 * q.numel() + kv.numel())
        nb_b = 2.0 * (4 * q.numel() + 2 * kv.numel())
        for tag, fn, nbytes in (("fwd", fwd, nb_f), ("bwd", bwd, nb_b)):
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                fn()
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=st):
                    for _ in range(reps):
                        fn()
                g.replay()
                torch.cuda.synchronize()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record(st)
                for _ in range(5):
                    g.replay()
                e.record(st)
                torch.cuda.synchronize()
            us = s.elapsed_time(e) * 1e3 / (5 * reps)
            flops = 4.0 * batch * 12 * Sq * Sk * 64 * (1.0 if tag == "fwd" else 2.5)
            res[f"{name} {tag}"] = {"Sq": Sq, "Sk": Sk, "avg_launch_us": round(us, 2), "achieved": round(nbytes / us / 1e3, 1),
                                    "frac": round(nbytes / us / 1e3 / PEAK_HBM_GBS, 4), "tflops": round(flops / us / 1e6, 1)}
    return {"kernel": f"attn_s128_fwd/bwd_kernel (x-layer cross attention, B={batch}, 12 heads x 64, bf16, dropout 0.1)", "bound": "hbm",
            "peak": PEAK_HBM_GBS, "unit": "GB/s", "cases": res}


def cpu_baseline(budget_s=20.0, batch=16, max_steps=8):
    """The CPU oracle (oracle/hamt_oracle.py, a port pinned to the reference's goldens) timed on the host:
    SAP train step (forward, mean, backward, clip 5.0, HF AdamW), dropout on, fp32, all host cores."""
    import torch as th
    from oracle.hamt_oracle import (HamtOracle, OracleConfig, adamw_step, clip_grad_norm, make_state_dict,
                                    pretrain_param_shapes)
    from vln_hamt_amd.synth import make_batch
    # measured on the GPU box (2 x EPYC 9575F, 256 logical CPUs): this model's small GEMMs get SLOWER beyond ~16
    # threads (1.4 s/step at 16, 1.9 at 32, 3.9 at 64), so the baseline uses the fastest setting, not all cores
    cores = int(os.environ.get("HAMT_CPU_THREADS", min(16, os.cpu_count() or 1)))
    th.set_num_threads(cores)
    cfg = OracleConfig()
    sd = make_state_dict(pretrain_param_shapes(cfg), seed=1)
    params = {k: v.requires_grad_(True) for k, v in sd.items() if k != "mlm_head.predictions.decoder.weight"}
    state, times = {}, []
    t_start = time.time()
    i = 0
    while True:
        b = make_batch("sap", batch, cfg, seed=1000 + i)
        t0 = time.time()
        loss = HamtOracle(params, cfg, training=True).forward(b, "sap", True).mean()
        loss.backward()
        grads = {k: p.grad for k, p in params.items() if p.grad is not None}
        clip_grad_norm(list(grads.values()), 5.0)
        with th.no_grad():
            adamw_step(params, grads, state, 5e-5)
        for p in params.values():
            p.grad = None
        times.append(time.time() - t0)
        i += 1
        if i >= 2 and (time.time() - t_start > budget_s or i >= max_steps):
            break
    steady = times[1:] if len(times) > 1 else times
    per_step = sorted(steady)[len(steady) // 2]
    return {"value": round(batch / per_step, 3), "unit": "panorama-steps/s", "cores": cores, "kind": "port",
            "sample": f"SAP train step (fwd+bwd+clip+AdamW, dropout on, fp32), B={batch}, L=80, T=5, "
                      f"{len(times)} steps, median of steps 2.. ({per_step:.2f} s/step)"}


def log(msg):
    if int(os.environ.get("RANK", 0)) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n, argv, launcher=None):
    """Run this file as `n` ranks of one node under torch.distributed.run (one process per GPU, rendezvous on 127.0.0.1) and
    return the launcher's exit code.  Called only from a process that has not initialised the GPU."""
    import subprocess
    have = torch.cuda.device_count()                    # counting devices does not initialise the runtime
    if have < n and not os.environ.get("HAMT_DIST_BACKEND"):
        print(f"bench.py: --gpus {n} but this node shows {have} GPU(s)", file=sys.stderr, flush=True)
        return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this pool (RCCL across processes needs it)
    env.setdefault("OMP_NUM_THREADS", "8")
    env.setdefault("GLOO_SOCKET_IFNAME", "lo")          # (one node: no host-name lookups in a gloo rendezvous)
    cmd = launcher or [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
                       "--master-addr", "127.0.0.1", "--master-port", str(free_port())]
    cmd = cmd + [os.path.abspath(__file__)] + list(argv)
    log("self-launch: " + " ".join(cmd))
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=96)
    ap.add_argument("--warmup", type=int, default=12)
    ap.add_argument("--batch", type=int, default=64, help="per-GPU minibatch (ITM uses batch/2 originals, loader.py:130)")
    ap.add_argument("--prec", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--task", default="mix", help="mix (5:1:1:1:2:2 cycle) or one of mlm/sap/sar/sprel/mrc/itm")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    ap.add_argument("--no-probes", action="store_true", help="only the timed steps (for rocprofv3 runs: no probe launches in the kernel statistics)")
    ap.add_argument("--ragged", action="store_true", help="headline run on ragged batches (SURVEY 8d: L ~ U[20, 80], T ~ U[0, 7] per sample, padded to the "
                    "batch maximum as the reference's collate does) instead of full-length ones; without the flag the ragged variant is reported "
                    "next to `batch_sweep` as `ragged`")
    ap.add_argument("--regions", type=int, default=3, help="timed regions of --steps steps each: the first is `value` (the contract's K steps), all of "
                    "them are reported in `regions_ms_per_step` with their min / median")
    ap.add_argument("--also-batch", type=int, default=-1, help="another per-GPU batch reported in `batch_sweep` (default: 16 -- the reference's "
                    "per-GPU batch -- and 256 when --batch is left at 64 on one GPU; 0 = none)")
    ap.add_argument("--launch-check", action="store_true", help="plumbing check of the N-rank launch: rendezvous, one all-reduce, rank 0 prints "
                    "{\"launch_check\": true, \"n_gpus\": world}; no model, runs on a CPU-only host over gloo as well")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        # `python bench.py --gpus N` run plainly (the way the driver runs --gpus 1; the reference is started with
        # `python -m torch.distributed.launch --nproc_per_node N`, README.md:48-55): start the N ranks ourselves.  A CHILD process,
        # spawned before anything here has touched the GPU (no exec: replacing a process that holds the device takes the box down);
        # its stdout -- rank 0's JSON line -- is inherited, its return code is ours.
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))

    from vln_hamt_amd import ops
    from vln_hamt_amd.optim import AdamW, clip_grad_norm_
    from vln_hamt_amd.optim.misc import NO_DECAY
    from vln_hamt_amd.parallel import (TaskSchedule, make_grad_sync, allreduce_grads, barrier, broadcast_params, default_wire,
                                       init_distributed, max_over_ranks, sum_over_ranks)
    from vln_hamt_amd.synth import make_batch, make_itm_rng

    rank, local_rank, world = init_distributed()
    assert world == args.gpus, f"--gpus {args.gpus} but the launcher started {world} rank(s)"
    if args.launch_check:
        t = torch.ones(1, device=torch.device("cuda", local_rank % torch.cuda.device_count()) if torch.cuda.is_available() else "cpu")
        if world > 1:
            torch.distributed.all_reduce(t)
        assert int(t.item()) == world, (t, world)
        if rank == 0:
            print(json.dumps({"launch_check": True, "n_gpus": world, "backend": torch.distributed.get_backend() if world > 1 else None}), flush=True)
        if world > 1:
            torch.distributed.destroy_process_group()
        return
    assert torch.cuda.is_available(), "bench.py needs a GPU: the HAMT kernels have no CPU path"
    device = torch.device("cuda", (local_rank if world > 1 else 0) % max(1, torch.cuda.device_count()))
    torch.cuda.set_device(device)
    ops.manual_seed(1234 + rank, device)

    log("building model")
    model, cfg = build_model(args.prec, device)
    named = list(model.named_parameters())
    groups = [{"params": [p for n, p in named if not any(nd in n for nd in NO_DECAY)], "weight_decay": 0.01},
              {"params": [p for n, p in named if any(nd in n for nd in NO_DECAY)], "weight_decay": 0.0}]
    opt = AdamW(groups, lr=5e-5, betas=(0.9, 0.98))
    opt.materialize()                                   # flat fp32 parameter / gradient / moment arenas + bf16 shadow
    dist_on = torch.distributed.is_available() and torch.distributed.is_initialized()
    if dist_on:
        broadcast_params(opt)                           # every rank starts from rank 0's weights (DDP does this at wrap time)
    # gradient exchange: flat-arena RCCL all-reduces, range by range behind the grouped weight-gradient GEMMs
    grad_sync = None
    if dist_on:
        wire = default_wire(args.prec)                  # bf16 mode: bf16 on the wire (DDP bf16_compress_hook arithmetic)
        ng = int(os.environ.get("HAMT_SYNC_GROUPS", 4))
        if os.environ.get("HAMT_NO_OVERLAP"):
            grad_sync = lambda o: allreduce_grads(o, wire)
        else:     # bf16 mode: reduce-scatter -> owned-slice AdamW -> all-gather (ShardedGradSync) when the arenas split evenly over the
            # ranks; else / HAMT_SHARDED=0 / fp32 mode: all-reduce + full AdamW on every rank (OverlappedGradSync)
            grad_sync = make_grad_sync(opt, args.prec, n_groups=ng, wire=wire)
    net = model

    from vln_hamt_amd.model import vilmodel as _vil
    sched = TaskSchedule(cyclic=True) if args.task == "mix" else None
    n_distinct = 12
    batches = {}

    ragged_mode = [bool(args.ragged)]
    full_mode = [False]                                 # the `full_work` region: HAMT_NO_DCE behaviour (every launch of the reference's forward)

    def _sfx():
        return (("ragged",) if ragged_mode[0] else ()) + (("full",) if full_mode[0] else ())

    sample_flops = {}                                   # batch key -> 3 x forward FLOPs summed over the batch's samples at THEIR lengths
    dead_flops, dead_sum = {}, [0.0]                    # ... of which never read by anything (dead_fwd_flops), and their sum over the last timed region

    def get_batch(step, bsz=None):
        bsz = bsz or args.batch
        task = sched.task_at(step) if sched else args.task
        rg = ragged_mode[0]
        key = (task, step % n_distinct, bsz) + _sfx()
        if key not in batches:                          # synthetic inputs resident in HBM before the timed region
            b = make_batch(task, bsz, cfg, seed=1234 + rank + 7919 * (step % n_distinct), txt_len=L_TXT,
                           hist_len=7 if rg else T_HIST, ragged=rg, mlm_exact=12 if (task == "mlm" and not rg) else None, device=device)
            if task == "itm":
                r = make_itm_rng(b, seed=step)
                b["itm_neg_idxs"], b["itm_shuffled_pos_ids"] = r["neg_idxs"], r["shuffled_pos_ids"]
            batches[key] = b
            if rg:      # algorithmic work of a ragged batch: every sample at its own instruction / history length
                ls = b["txt_masks"].sum(1).tolist()
                ts = (b["hist_masks"].sum(1) - 1).tolist() if b.get("hist_masks") is not None else [0] * len(ls)
                nm = int(b["txt_label_idx"].numel()) if task == "mlm" else 0
                sample_flops[key] = 3.0 * (sum(trunk_fwd_flops(task, int(l), int(t)) - (2 * 12 * H * 30522 + 2 * 12 * H * H if task == "mlm" else 0)
                                               for l, t in zip(ls, ts)) + 2 * nm * H * (30522 + H))
                # (the padded rows of a padded batch are launched, so the dead rows are counted at the padded length; packed text: at the sample's)
                dead_flops[key] = 3.0 * sum(dead_fwd_flops(task, int(l), int(b["hist_masks"].shape[1]) - 1 if b.get("hist_masks") is not None else 0,
                                                           nm / max(len(ls), 1)) for l in ls)
            else:
                sample_flops[key] = 3.0 * trunk_fwd_flops(task) * b["txt_ids"].shape[0]
                dead_flops[key] = 3.0 * dead_fwd_flops(task) * b["txt_ids"].shape[0]
        return task, batches[key]

    log("generating synthetic batches")
    for s in range(args.warmup + args.steps):
        get_batch(s)
    log("batches resident in HBM")
    gstep = [0]
    use_graph = not args.no_graph
    graphed = None
    if use_graph:
        from vln_hamt_amd.graph import GraphedTrainStep
        graphed = GraphedTrainStep(model, opt, max_grad_norm=5.0, grad_sync=grad_sync)

    def train_step(step, bsz=None):
        bsz = bsz or args.batch
        task, b = get_batch(step, bsz)
        if graphed is not None:
            gstep[0] += 1
            lr = 5e-5 * min(1.0, gstep[0] / 10000.0)
            for g in opt.param_groups:
                g["lr"] = lr
            key = (task, step % n_distinct, bsz) + _sfx()
            graphed.step(key, b, task)
            # the bench's inputs are resident in HBM: keep working on the captured step's own static input tensors (a
            # loader would write each new batch into them; GraphedTrainStep.step copies any other batch in)
            batches[key] = graphed.static_batch(key)
            return task, b["txt_ids"].shape[0]
        loss = net(b, task, True).mean()
        loss.backward()
        if grad_sync is not None:
            grad_sync(opt)
        gstep[0] += 1
        lr = 5e-5 * min(1.0, gstep[0] / 10000.0)
        for g in opt.param_groups:
            g["lr"] = lr
        if getattr(grad_sync, "sharded", False):
            opt.prepare_step()
            grad_sync.update(5.0)
        else:
            clip_grad_norm_(model.parameters(), 5.0, optimizer=opt)
            opt.step()
        opt.zero_grad()
        ops.advance_rng_epoch(device)
        return task, b["txt_ids"].shape[0]

    if graphed is not None and args.warmup < 2 * n_distinct:
        # the first pass over the 12 distinct (task, batch) keys CAPTURES their graphs (a host-bound eager step + the capture each: the GPU
        # idles through most of it); a second pass replays them, so that the timed region starts from a busy chip, not from idle clocks
        log(f"note: warmup raised to {2 * n_distinct}: one pass captures every (task, batch) graph, one replays them before the timed region")
        args.warmup = 2 * n_distinct
    def timed_region(bsz, warmup, steps, verbose=True):
        """`warmup` untimed steps, then exactly `steps` steps between barrier + synchronize; returns (seconds = max over ranks, samples, flops)"""
        for s in range(warmup):
            t_ = time.perf_counter()
            tk, _ = train_step(s, bsz)
            if s < 3 and verbose:
                torch.cuda.synchronize()
                log(f"warmup step {s} ({tk}): {time.perf_counter() - t_:.3f} s")
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        samples, flops = 0, 0.0
        dead_sum[0] = 0.0
        for s in range(warmup, warmup + steps):
            task, n = train_step(s, bsz)
            samples += n
            flops += sample_flops[(task, s % n_distinct, bsz) + _sfx()]
            dead_sum[0] += dead_flops[(task, s % n_distinct, bsz) + _sfx()]
        if graphed is not None:
            graphed.finish()                 # the last step's parameter update (overlap_update: a replay applies the previous step's)
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        return max_over_ranks(time.perf_counter() - t0, device), samples, flops

    dt, samples, flops = timed_region(args.batch, args.warmup, args.steps)
    dead_main = sum_over_ranks(dead_sum[0], device)
    log(f"timed region: {dt:.3f} s for {args.steps} steps")
    # the driver's K may be small (20 steps = 0.2 s): repeat the region so that box-to-box / run-to-run spread is visible
    regions = [dt / args.steps * 1e3]
    for _ in range(max(0, args.regions - 1)):
        dt_r, _, _ = timed_region(args.batch, 0, args.steps, verbose=False)
        regions.append(dt_r / args.steps * 1e3)
    exposed_comm_ms = None
    if dist_on and not args.no_probes:
        # the same steps with the collectives themselves skipped (every rank keeps its own values: timing only, after the measurement):
        # the difference is the part of the exchange that compute does not hide
        from vln_hamt_amd import parallel as par
        par.DRY[0] = True
        dt_dry, _, _ = timed_region(args.batch, 2, args.steps, verbose=False)
        par.DRY[0] = False
        exposed_comm_ms = round((dt - dt_dry) / args.steps * 1e3, 3)
        log(f"without the collectives: {dt_dry / args.steps * 1e3:.3f} ms/step -> exposed communication {exposed_comm_ms} ms/step")
    total_samples = sum_over_ranks(float(samples), device)
    total_flops = sum_over_ranks(flops, device)

    executed_flops = total_flops - (dead_main if _vil.DEAD_SIDE_ELIMINATION else dead_main * 2.0 / 3.0)
    if rank == 0:
        out = {
            "metric": "panorama-steps/sec, R2R proxy pretrain (36x768 views, 80-tok instr)",
            "value": round(total_samples / dt, 2), "unit": "panorama-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.prec, "data": "synthetic",
            "config": {"workload": "R2R 6-proxy-task pretrain step (fwd+bwd+clip+AdamW, dropout 0.1), fixed ViT features, "
                                   "R2R-canon model 174.8M params" if args.task == "mix" else f"R2R {args.task} pretrain step",
                       "per_gpu_batch": args.batch, "global_batch": args.batch * world, "txt_len": "U[20,80]" if args.ragged else L_TXT,
                       "hist_len": "U[0,7]" if args.ragged else T_HIST,
                       "views": V, "task_mix": "mlm:sap:sar:sprel:mrc:itm=5:1:1:1:2:2" if args.task == "mix" else args.task,
                       "parallelism": f"dp{world}" + (f" (flat-arena {'RCCL' if torch.distributed.get_backend() == 'nccl' else torch.distributed.get_backend()} " + ("reduce-scatter + owned-slice AdamW + all-gather" if getattr(grad_sync, "sharded", False) else "all-reduce") + f", {wire} on the wire" + (", overlapped with wgrad" if getattr(grad_sync, "overlapped", False) else "") + ")" if dist_on else ""),
                       "launch": "hipGraph replay" if graphed is not None else "eager",
                       # what no head reads of the LAST cross-modal layer is not launched (results unchanged: loss and every gradient);
                       # `full_work` below is the same step with every launch of the reference's forward (HAMT_NO_DCE=1)
                       "dead_code_elimination": bool(_vil.DEAD_SIDE_ELIMINATION)},
            "per_gpu": round(total_samples / dt / world, 2),
            # SURVEY 8a's 3 x forward (the figure BASELINE's metric is priced in) counts work the reference launches and never reads
            # (dead_fwd_flops) three times; executed here: none of it.  The end-to-end roofline fraction is of the EXECUTED FLOPs
            # (ADVICE r5); `..._reference_work` = the same step priced at 3 x forward, like-for-like with `full_work` below
            "model_tflops_per_gpu": round(total_flops / dt / world / 1e12, 2),
            "executed_tflops_per_gpu": round(executed_flops / dt / world / 1e12, 2),
            "mfma_roofline_frac_end_to_end": round(executed_flops / dt / world / 1e12 / PEAK_BF16_TFLOPS, 4),
            "mfma_roofline_frac_reference_work": round(total_flops / dt / world / 1e12 / PEAK_BF16_TFLOPS, 4),
            "hbm_peak_allocated_gb": round(torch.cuda.max_memory_allocated(device) / 2 ** 30, 1),
            "exposed_comm_ms_per_step": exposed_comm_ms,
            "regions_ms_per_step": [round(r, 3) for r in regions], "regions_min_ms": round(min(regions), 3),
            "regions_median_ms": round(sorted(regions)[len(regions) // 2], 3),
            # the timed steps trained a real model: every master parameter and both moments are finite afterwards (a NaN anywhere
            # spreads within a few steps; NaN-filled operands also run at other clocks than real data, see DESIGN 4 on DVFS)
            "state_finite_after_timed_region": bool(torch.isfinite(opt._flat_p).all() and torch.isfinite(opt._flat_m).all() and torch.isfinite(opt._flat_v).all()),
        }
        # the other two per-GPU batches SURVEY 8d names next to the headline one -- 16 is the reference's own (pretrain_r2r.json:9) -- on
        # the same model / optimizer / number of steps
        also = [args.also_batch] if args.also_batch > 0 else ([16, 256] if (args.also_batch < 0 and args.batch == 64 and world == 1 and args.task == "mix") else [])
        if also and world == 1 and not args.no_probes:
            out["batch_sweep"] = []
            for bsz in also:
                dt2, smp2, fl2 = timed_region(bsz, args.warmup, args.steps if bsz <= 64 else max(12, args.steps // 2), verbose=False)
                n2 = args.steps if bsz <= 64 else max(12, args.steps // 2)
                ex2 = fl2 - (dead_sum[0] if _vil.DEAD_SIDE_ELIMINATION else dead_sum[0] * 2.0 / 3.0)
                out["batch_sweep"].append({"per_gpu_batch": bsz, "value": round(smp2 / dt2, 2), "unit": "panorama-steps/s", "steps": n2,
                                           "ms_per_step": round(dt2 / n2 * 1e3, 3), "model_tflops_per_gpu": round(fl2 / dt2 / 1e12, 2),
                                           "executed_tflops_per_gpu": round(ex2 / dt2 / 1e12, 2),
                                           "mfma_roofline_frac_end_to_end": round(ex2 / dt2 / 1e12 / PEAK_BF16_TFLOPS, 4),
                                           "state_finite_after_timed_region": bool(torch.isfinite(opt._flat_p).all() and torch.isfinite(opt._flat_m).all())})
                log(f"batch {bsz}: {dt2 / n2 * 1e3:.3f} ms/step")
                for k in [k for k in batches if k[2] == bsz]:
                    del batches[k]
        if _vil.DEAD_SIDE_ELIMINATION and not args.no_probes and world == 1:
            # the same step with EVERY launch of the reference's forward (the unread side / rows of the last cross-modal layer included)
            full_mode[0] = True
            _vil.DEAD_SIDE_ELIMINATION = False
            try:
                dt4, smp4, fl4 = timed_region(args.batch, 2 * n_distinct, max(24, args.steps // 2), verbose=False)
            finally:
                _vil.DEAD_SIDE_ELIMINATION = True
                full_mode[0] = False
            n4 = max(24, args.steps // 2)
            out["full_work"] = {"what": "HAMT_NO_DCE=1: nothing of the reference's forward skipped", "steps": n4, "ms_per_step": round(dt4 / n4 * 1e3, 3),
                                "value": round(smp4 / dt4, 2), "unit": "panorama-steps/s"}
            log(f"full work (no dead-code elimination): {dt4 / n4 * 1e3:.3f} ms/step")
            for k in [k for k in batches if k[-1] == "full"]:
                del batches[k]
        if not args.ragged and not args.no_probes and args.task == "mix" and world == 1:
            # SURVEY 8d's ragged variant: per-sample L ~ U[20, 80], T ~ U[0, 7], padded to the batch maximum (the reference's collate);
            # throughput in samples/s, FLOPs counted at every sample's OWN lengths (padding is work the path does, not work it is credited for)
            ragged_mode[0] = True
            dt3, smp3, fl3 = timed_region(args.batch, 2 * n_distinct, args.steps, verbose=False)
            s3, f3 = float(smp3), fl3
            out["ragged"] = {"per_gpu_batch": args.batch, "txt_len": "U[20,80]", "hist_len": "U[0,7]", "value": round(s3 / dt3, 2), "unit": "panorama-steps/s",
                             "steps": args.steps, "ms_per_step": round(dt3 / args.steps * 1e3, 3),
                             "model_tflops_per_gpu_at_sample_lengths": round(f3 / dt3 / world / 1e12, 2)}
            log(f"ragged batches: {dt3 / args.steps * 1e3:.3f} ms/step")
            ragged_mode[0] = False
            for k in [k for k in batches if k[-1] == "ragged"]:
                del batches[k]
        if args.task == "mix":
            cycle = [get_batch(s_) for s_ in range(len(sched.cycle))]
        else:
            cycle = [get_batch(0)]
        if dist_on:
            # the probes below run model passes on THIS rank only: a gradient exchange they started would wait for ranks that sit
            # in the final barrier.  (Found with a 2-rank run at the end of round 2: the bench hung here for any N > 1.)
            from vln_hamt_amd import parallel as par
            par.DRY[0] = True
        if not args.no_probes:
            from tools import roofline_probe as rp
            table = rp.kernel_table(model, opt, cycle, device, live=not dist_on)
            dom = dict(table[0])                            # the kernel with the largest share of a step
            traffic, src = rp.traffic_of(dom["kernel"], args.batch)
            out["roofline"] = {"bound": dom["bound"], "achieved": dom["achieved"], "peak": dom["peak"], "unit": dom["unit"], "frac": dom["frac"],
                               "traffic": traffic, "traffic_source": src, "kernel": dom["kernel"], "per_step_ms": dom["per_step_ms"],
                               "launches_per_step": dom["launches_per_step"], "avg_launch_us": dom["avg_launch_us"],
                               "algorithmic_work_per_launch": dom["work_per_launch"],
                               "how": "dominant kernel of the step = largest per-step total among the kernels of one task-mix cycle.  The grouped "
                                      "weight-gradient kernel is timed where it runs: one eager forward + backward per task, HIP events recorded on the "
                                      "launch stream right around each kernel launch (hamt_debug_wgrad_timing; avg_launch_us is per KERNEL launch, the "
                                      "unit of the rocprofv3 summary in profiles/); back_to_back_ms_per_step = the same problems re-issued back to back "
                                      "on fresh targets.  GEMM groups / AdamW: re-issued (same operands, same epilogues) as a captured hipGraph, HIP "
                                      "events on the launch stream.  work = 2MNK per GEMM launch (30 B per parameter, + 4 where the gradient slot is "
                                      "zeroed, for the AdamW kernel)"}
            for k_ in ("back_to_back_ms_per_step", "back_to_back_achieved"):
                if k_ in dom:
                    out["roofline"][k_] = dom[k_]
            out["kernel_table"] = [{k: v for k, v in r.items() if k != "work_per_launch"} for r in table[:12]]
            out["box"] = box_probe(device)
            out["roofline_subblock_xattn"] = rp.subblock_xattn(model, args.batch, device)
            out["roofline_probe_xattn"] = time_xattn_probe(args.batch, device)
        if world == 1 and not dist_on and not args.no_probes and graphed is not None and args.prec == "bf16" and args.task == "mix":
            # What the multi-GPU step's machinery costs before a byte moves: the same steps through the sharded exchange
            # (parallel.ShardedGradSync: launch groups on two lanes, wire staging, reduce-scatter / owned-slice AdamW / all-gather
            # calls, the eager section between the graphs) with a ONE-rank RCCL group, against the plain single-GPU step above.
            try:
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 200))
                torch.distributed.init_process_group(backend=os.environ.get("HAMT_DIST_BACKEND") or "nccl", rank=0, world_size=1)
                gs1 = make_grad_sync(opt, args.prec, n_groups=int(os.environ.get("HAMT_SYNC_GROUPS", 4)), wire=default_wire(args.prec))
                plain_graphed, graphed = graphed, GraphedTrainStep(model, opt, max_grad_norm=5.0, grad_sync=gs1)
                dt_w1, _, _ = timed_region(args.batch, 2 * n_distinct, args.steps, verbose=False)
                out["exchange_overhead_ms_world1"] = round(dt_w1 / args.steps * 1e3 - min(regions), 3)
                out["world1_exchange_ms_per_step"] = round(dt_w1 / args.steps * 1e3, 3)
                log(f"one-rank sharded exchange: {dt_w1 / args.steps * 1e3:.3f} ms/step (plain {min(regions):.3f})")
                gs1.close()
                graphed = plain_graphed
                torch.distributed.destroy_process_group()
            except Exception as e:      # a measurement aid must never cost the bench line
                out["exchange_overhead_ms_world1"] = None
                log(f"one-rank exchange probe failed: {type(e).__name__}: {e}")
        log("roofline probes done; timing the CPU oracle baseline")
        if world == 1 and not args.no_cpu_baseline and not args.no_probes:
            out["cpu_baseline"] = cpu_baseline(args.cpu_budget)
            # SURVEY 8d names two CPU cases: B = 16 (above: the reference's per-GPU batch) and B = 2 (BASELINE config 1, the reference's
            # own CPU-runnable plumbing case)
            b2 = cpu_baseline(min(8.0, args.cpu_budget), batch=2, max_steps=10)
            out["cpu_baseline"]["batch2"] = {"value": b2["value"], "unit": b2["unit"], "sample": b2["sample"]}
        else:
            out["cpu_baseline"] = None
        if world == 1 and not dist_on and not args.no_probes and args.task == "mix" and args.prec == "bf16" and not os.environ.get("HAMT_BENCH_NO_EXTRA"):
            # BASELINE configs 4 and 5 on this GPU, driver-visible (VERDICT r3 missing 6): measurement aids behind the headline -- they build
            # their own models and must never cost the bench line
            for key, mod, kw in (("e2e_image_step", "e2e_bench", dict(B=1, steps=12, use_graph=True)),
                                 ("rollout_step", "rollout_bench", dict(batch=8, txt=160, steps=20, feat=512, reps=4))):
                try:
                    import importlib
                    out[key] = importlib.import_module("tools." + mod).run(dev=device, **kw)
                    log(f"{key}: {out[key]['ms_per_step']} ms/step, {out[key]['value']} {out[key]['unit']}")
                except Exception as e:
                    out[key] = None
                    log(f"{key} failed: {type(e).__name__}: {e}")
        import ctypes
        ctypes.CDLL(None).fflush(None)      # RCCL printf()s its library path into C stdio: keep the JSON the LAST line
        sys.stdout.flush()
        print(json.dumps(out), flush=True)
    barrier()
    if dist_on:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
