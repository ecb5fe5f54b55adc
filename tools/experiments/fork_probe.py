#!/usr/bin/env python3
"""Do two graph launches that become ready at the same moment on two streams run concurrently?  A prefix P on s0 (long enough that
the host has issued everything before the GPU gets to the fork), an event behind it, then A on s0 and B on s1 (B waits for the
event).  Kernels are one-workgroup scans (~tens of us, one CU each): two chains do not compete for anything.  Concurrent: P + max(A, B);
serialized: P + A + B.  Variants: issue order (A first / B first), B without the event wait, A and B as eager launches."""
import os, sys, time
import torch

dev = torch.device("cuda")
NP, NA, NB = 40, 60, 60
L = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 17
xs = [torch.ones(1, L, device=dev) for _ in range(3)]
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
if os.environ.get("SKIP"):
    _d = [torch.cuda.Stream() for _ in range(int(os.environ["SKIP"]))]
    s1 = torch.cuda.Stream()


def k(x):
    torch.cumsum(x, 1, out=x)
    x.mul_(0.0).add_(1.0)


def chain(x, n, st):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=st):
            for _ in range(n):
                k(x)
    return g


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


gP, gA, gB = chain(xs[0], NP, s0), chain(xs[1], NA, s0), chain(xs[2], NB, s1)
ev = torch.cuda.Event()
tP, tA, tB = timeit(gP.replay), timeit(gA.replay), timeit(gB.replay)


def run(order, wait=True, eager_b=False):
    with torch.cuda.stream(s0):
        gP.replay()
        ev.record(s0)
    for which in order:
        if which == "A":
            with torch.cuda.stream(s0):
                gA.replay()
        else:
            with torch.cuda.stream(s1):
                if wait:
                    s1.wait_event(ev)
                if eager_b:
                    for _ in range(NB):
                        k(xs[2])
                else:
                    gB.replay()
    s0.wait_stream(s1)


# ---- the same dependency through stream memory operations (hipStreamWriteValue32 behind P, hipStreamWaitValue32 in front of B)
import ctypes as C
hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
flag = C.c_void_p()
hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
rc = hip.hipExtMallocWithFlags(C.byref(flag), 8, 2)      # hipMallocSignalMemory: 8 bytes
if rc != 0:
    print("signal memory allocation failed:", rc, "-> plain hipMalloc", flush=True)
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    assert hip.hipMalloc(C.byref(flag), 64) == 0
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
hip.hipMemset(flag, 0, 8)
torch.cuda.synchronize()
hip.hipStreamWaitValue32.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint, C.c_uint32]
hip.hipStreamWriteValue32.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint]
epoch = [0]


def run_value(order):
    epoch[0] += 1
    with torch.cuda.stream(s0):
        gP.replay()
        assert hip.hipStreamWriteValue32(C.c_void_p(s0.cuda_stream), flag, epoch[0], 0) == 0
    for which in order:
        if which == "A":
            with torch.cuda.stream(s0):
                gA.replay()
        else:
            assert hip.hipStreamWaitValue32(C.c_void_p(s1.cuda_stream), flag, epoch[0], 0, 0xFFFFFFFF) == 0      # Gte
            with torch.cuda.stream(s1):
                gB.replay()
    s0.wait_stream(s1)


def run_jit(order):
    """the host waits until P is (nearly) done and only then issues A and B"""
    with torch.cuda.stream(s0):
        gP.replay()
        ev.record(s0)
    ev.synchronize()
    for which in order:
        if which == "A":
            with torch.cuda.stream(s0):
                gA.replay()
        else:
            with torch.cuda.stream(s1):
                gB.replay()
    s0.wait_stream(s1)


evq = torch.cuda.Event()


def run_jit_early(order, frac=0.8):
    """the host waits for an event recorded INSIDE P (P as two graphs), then issues A and B while P's tail still runs"""
    with torch.cuda.stream(s0):
        gP1.replay()
        evq.record(s0)
        gP2.replay()
        ev.record(s0)
    evq.synchronize()
    for which in order:
        if which == "A":
            with torch.cuda.stream(s0):
                gA.replay()
        else:
            with torch.cuda.stream(s1):
                s1.wait_event(ev)
                gB.replay()
    s0.wait_stream(s1)


gP1, gP2 = chain(xs[0], NP * 3 // 4, s0), chain(xs[0], NP - NP * 3 // 4, s0)
print(f"L {L}: host waits for P, then issues A and B: {timeit(lambda: run_jit('AB')):.3f}; host waits for 3/4 of P, then issues A and (event wait +) B: {timeit(lambda: run_jit_early('AB')):.3f} / B first {timeit(lambda: run_jit_early('BA')):.3f}", flush=True)
tv1, tv2 = timeit(lambda: run_value("AB")), timeit(lambda: run_value("BA"))
print(f"L {L}: stream memory operations instead of the event: A then B {tv1:.3f}; B then A {tv2:.3f}", flush=True)
print(f"L {L}: P {tP:.3f} ms, A {tA:.3f}, B {tB:.3f}  [concurrent = {tP + max(tA, tB):.3f}, serialized = {tP + tA + tB:.3f}]:  A then B {timeit(lambda: run('AB')):.3f}; "
      f"B then A {timeit(lambda: run('BA')):.3f}; B without the event wait, A then B {timeit(lambda: run('AB', wait=False)):.3f} (lower bound {max(tP + tA, tB):.3f}); "
      f"B eager, A then B {timeit(lambda: run('AB', eager_b=True)):.3f}", flush=True)
