#!/usr/bin/env python3
"""When does a branch of a captured graph that is WAITING for its dependency get to run next to a long chain on the other stream?
Model of the step's backward: s0 runs a prefix P and then a long chain A; s1 runs a short early chain S, then has to wait for the end
of P (the event in the middle of s0's work) before its chain B.  One-workgroup scan kernels (~10-20 us, one CU each): the chains compete
for nothing.  Ideal: P + max(A, B); serialized: P + A + B.  Variants: `bounce` = every Q kernels chain A takes a detour through a
third stream (fork + one tiny kernel + join: s0 has to WAIT for another queue for a moment).
usage: bounce_probe.py [L=131072]"""
import sys, time
import torch

dev = torch.device("cuda")
L = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 17
NP, NA, NB, NS = 40, 120, 60, 10
xa, xb, xc = (torch.ones(1, L, device=dev) for _ in range(3))
tiny = torch.ones(64, device=dev)
s0, s1, s2 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()


def k(x):
    torch.cumsum(x, 1, out=x)
    x.mul_(0.0).add_(1.0)


def build(bounce=0, b_first=False, late=True, external=False):
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.stream(s0):
        with torch.cuda.graph(g, stream=s0):
            s1.wait_stream(s0)
            with torch.cuda.stream(s1):
                for _ in range(NS):
                    k(xb)
            for _ in range(NP):
                k(xa)
            ev = torch.cuda.Event(external=True) if external else torch.cuda.Event()
            ev.record(s0)

            def chain_b():
                with torch.cuda.stream(s1):
                    if late:
                        s1.wait_event(ev)
                    for _ in range(NB):
                        k(xb)

            def chain_a():
                for i in range(NA):
                    k(xa)
                    if bounce and (i + 1) % bounce == 0 and i + 1 < NA:
                        s2.wait_stream(s0)
                        with torch.cuda.stream(s2):
                            tiny.add_(1.0)
                        s0.wait_stream(s2)
            if b_first:
                chain_b(); chain_a()
            else:
                chain_a(); chain_b()
            s0.wait_stream(s1)
    return g


def timeit(g, reps=10):
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def single(n, x):
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.stream(s0):
        with torch.cuda.graph(g, stream=s0):
            for _ in range(n):
                k(x)
    return timeit(g)


tP, tA, tB = single(NP, xa), single(NA, xa), single(NB, xb)
print(f"P {tP:.3f} ms, A {tA:.3f} ms, B {tB:.3f} ms: ideal {tP + max(tA, tB):.3f}, serialized {tP + tA + tB:.3f}")
for name, kw in [("plain, A captured first", dict()), ("plain, B captured first", dict(b_first=True)), ("B not waiting (control)", dict(late=False)),
                 ("bounce every 32", dict(bounce=32)), ("bounce every 16", dict(bounce=16)), ("bounce every 8", dict(bounce=8)), ("bounce every 4", dict(bounce=4)),
                 ("bounce every 16, B first", dict(bounce=16, b_first=True))]:
    # (torch.cuda.Event(external=True) -- event record / wait NODES instead of internal edges -- aborts the process inside the capture on this
    #  ROCm 7.0 / torch 2.10 build: build(external=True) is kept for a later runtime)
    try:
        g = build(**kw)
    except Exception as e:
        print(f"{name:32s} failed: {str(e)[:120]}")
        continue
    print(f"{name:32s} {timeit(g):.3f} ms")
