#!/usr/bin/env python3
"""How does a replayed hipGraph run two independent branches?  Two chains of N small kernels each on two streams (fork / join
inside the capture), captured (a) chain A completely, then chain B, (b) interleaved A1 B1 A2 B2 ..., (c) in blocks of K;
replay time against one chain alone and against two separate graphs launched on two streams.  Kernel length is varied with
the tensor size (latency-bound 4 us kernels .. ~40 us bandwidth-bound ones)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

dev = torch.device("cuda")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100


def chain_step(x):
    x.mul_(1.0001)


def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for numel in (1 << 14, 1 << 22, 1 << 24):
    a = torch.ones(numel, device=dev)
    b = torch.ones(numel, device=dev)
    s0, s1, s2 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    res = {}

    def capture(order):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s0):
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s0):
                s1.wait_stream(s0)
                for which in order:
                    if which == "a":
                        chain_step(a)
                    else:
                        with torch.cuda.stream(s1):
                            chain_step(b)
                s0.wait_stream(s1)
        return g

    def capture_split(order):
        from vln_hamt_amd.graph import SplitGraph
        g = torch.cuda.CUDAGraph(keep_graph=True)
        with torch.cuda.stream(s0):
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s0):
                s1.wait_stream(s0)
                for which in order:
                    if which == "a":
                        chain_step(a)
                    else:
                        with torch.cuda.stream(s1):
                            chain_step(b)
                s0.wait_stream(s1)
        sg = SplitGraph(g, 2)
        return sg

    g_a = capture(["a"] * N)
    res["one chain"] = timeit(g_a.replay)
    g_seq = capture(["a"] * N + ["b"] * N)
    res["A then B"] = timeit(g_seq.replay)
    g_seq2 = capture(["b"] * N + ["a"] * N)
    res["B then A"] = timeit(g_seq2.replay)
    g_il = capture(["a", "b"] * N)
    res["interleaved 1:1"] = timeit(g_il.replay)
    sg = capture_split(["a"] * N + ["b"] * N)
    res[f"A then B, SPLIT {sg.info()}"] = timeit(sg.replay)
    # correctness of the split replay: both chains advance exactly like the plain replay
    a.fill_(1.0); b.fill_(1.0); torch.cuda.synchronize()
    sg.replay(); torch.cuda.synchronize()
    va, vb = float(a[0]), float(b[0])
    a.fill_(1.0); b.fill_(1.0); torch.cuda.synchronize()
    g_seq.replay(); torch.cuda.synchronize()
    assert abs(va - float(a[0])) < 1e-6 and abs(vb - float(b[0])) < 1e-6 and va > 1.0, (va, vb, float(a[0]), float(b[0]))
    for K in (4, 16):
        order = []
        for i in range(0, N, K):
            order += ["a"] * K + ["b"] * K
        res[f"blocks of {K}"] = timeit(capture(order).replay)
    # two graphs on two streams
    def cap_one(x, st):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(st):
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=st):
                for _ in range(N):
                    chain_step(x)
        return g
    ga, gb = cap_one(a, s0), cap_one(b, s2)

    def two():
        with torch.cuda.stream(s0):
            ga.replay()
        with torch.cuda.stream(s2):
            gb.replay()
    res["two graphs, two streams"] = timeit(two)

    def eager_two():
        for _ in range(N):
            with torch.cuda.stream(s0):
                chain_step(a)
            with torch.cuda.stream(s2):
                chain_step(b)
    res["eager, two streams"] = timeit(eager_two, 5)
    print(f"numel {numel} ({numel * 8 / 1e6:.1f} MB moved per kernel), {N} kernels per chain: " + "; ".join(f"{k} {v:.3f} ms" for k, v in res.items()), flush=True)
