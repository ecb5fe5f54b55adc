#!/usr/bin/env python3
"""How precisely does a cross-stream event dependency resolve?  Stream s0 runs a chain A of N kernels with an event recorded
after its K-th kernel; stream s1 waits for that event and runs a chain B of N - K kernels.  Precise: B overlaps A's tail, total
~= N kernels; coarse (B only starts when A's batch is done): ~= 2N - K kernels.  Eager launches, one captured graph, and
separate graphs per piece (what hamt_graph_split_launch issues)."""
import os, sys, time
import torch

dev = torch.device("cuda")
N, K = 100, 10
numel = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 22      # ~5 us per kernel; the chains do not saturate HBM together? (they do at 1<<24)
a, b = torch.ones(numel, device=dev), torch.ones(numel, device=dev)
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def eager():
    ev = torch.cuda.Event()
    with torch.cuda.stream(s0):
        for i in range(N):
            a.mul_(1.0001)
            if i == K - 1:
                ev.record(s0)
    with torch.cuda.stream(s1):
        s1.wait_event(ev)
        for i in range(N - K):
            b.mul_(1.0001)
    s0.wait_stream(s1)


def chain(x, n, st):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=st):
            for _ in range(n):
                x.mul_(1.0001)
    return g


one = chain(a, N, s0)
gA1, gA2, gB = chain(a, K, s0), chain(a, N - K, s0), chain(b, N - K, s1)
ev = torch.cuda.Event()


def pieces():
    with torch.cuda.stream(s0):
        gA1.replay()
        ev.record(s0)
        gA2.replay()
    with torch.cuda.stream(s1):
        s1.wait_event(ev)
        gB.replay()
    s0.wait_stream(s1)


def pieces_b_first():      # B's graph is issued BEFORE A's tail
    with torch.cuda.stream(s0):
        gA1.replay()
        ev.record(s0)
    with torch.cuda.stream(s1):
        s1.wait_event(ev)
        gB.replay()
    with torch.cuda.stream(s0):
        gA2.replay()
    s0.wait_stream(s1)


small = [chain(a, 10, s0) for _ in range((N - K) // 10)]


def pieces_small():      # A's tail as graphs of 10 nodes each
    with torch.cuda.stream(s0):
        gA1.replay()
        ev.record(s0)
    with torch.cuda.stream(s1):
        s1.wait_event(ev)
        gB.replay()
    with torch.cuda.stream(s0):
        for g_ in small:
            g_.replay()
    s0.wait_stream(s1)


def pieces_sync_between():      # host waits for the event before issuing B: what a precise dependency would give (plus the host round trip)
    with torch.cuda.stream(s0):
        gA1.replay()
        ev.record(s0)
        gA2.replay()
    ev.synchronize()
    with torch.cuda.stream(s1):
        gB.replay()
    s0.wait_stream(s1)


g1 = torch.cuda.CUDAGraph()
with torch.cuda.stream(s0):
    torch.cuda.synchronize()
    with torch.cuda.graph(g1, stream=s0):
        for i in range(N):
            a.mul_(1.0001)
            if i == K - 1:
                s1.wait_stream(s0)
        with torch.cuda.stream(s1):
            for i in range(N - K):
                b.mul_(1.0001)
        s0.wait_stream(s1)

t1 = timeit(one.replay)
print(f"numel {numel}: one chain of {N}: {t1:.3f} ms ({t1 / N * 1e3:.1f} us per kernel); fork after kernel {K}: eager {timeit(eager, 3):.3f} ms, one graph {timeit(g1.replay):.3f} ms, "
      f"graph pieces (A1, event, A2 | wait, B) {timeit(pieces):.3f} ms, pieces with B issued before A2 {timeit(pieces_b_first):.3f} ms, A2 as 9 graphs of 10 {timeit(pieces_small):.3f} ms, "
      f"host waits for the event then issues B {timeit(pieces_sync_between):.3f} ms   [precise = ~{t1:.2f}, coarse = ~{t1 * (2 * N - K) / N:.2f}]", flush=True)
