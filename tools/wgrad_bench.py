#!/usr/bin/env python3
"""Micro-benchmark of hamt_wgrad_grouped.  usage: wgrad_bench.py [count:MxNxK ...]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vln_hamt_amd import _lib as L, ops

def bench(specs, with_db=True, iters=20, warm=3):
    lib = L.load()
    probs, keep, flops = [], [], 0.0
    for cnt, M, N, K in specs:
        for _ in range(cnt):
            dy = torch.randn(K, M, device="cuda").to(torch.bfloat16); x = torch.randn(K, N, device="cuda").to(torch.bfloat16)
            dw = torch.empty(M, N, device="cuda"); db = torch.empty(M, device="cuda")
            keep += [dy, x, dw, db]; probs.append((dy, x, dw, db, M, N, K)); flops += 2.0 * M * N * K
    descs = (L.WgradDesc * len(probs))()
    for i, (dy, x, dw, db, M, N, K) in enumerate(probs):
        d = descs[i]
        d.dy, d.x, d.dw, d.db = dy.data_ptr(), x.data_ptr(), dw.data_ptr(), (db.data_ptr() if with_db else None)
        d.M, d.N, d.K, d.ldy, d.ldx, d.ldw, d.accum_dw, d.accum_db = M, N, K, M, N, N, 0, 0
    tab = torch.empty(sum((pr[4] + 63) // 64 for pr in probs) * L.WGRAD_TABLE_ENTRY, dtype=torch.uint8, device="cuda")
    fn = lambda: L.check(lib.hamt_wgrad_grouped(len(probs), descs, tab.data_ptr(), tab.numel(), ops._stream()), "wgrad")
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / iters * 1e3
    return us, flops / us / 1e6

if __name__ == "__main__":
    sets = [a for a in sys.argv[1:]] or ["13:768x768x5120", "39:768x768x5120", "13:3072x768x5120", "13:768x3072x5120",
                                          "52:768x768x5120,13:3072x768x5120,13:768x3072x5120", "1:4096x4096x4096", "8:768x768x11520,2:3072x768x11520,2:768x3072x11520"]
    for st in sets:
        specs = []
        for part in st.split(","):
            c, dims = part.split(":"); M, N, K = (int(v) for v in dims.split("x")); specs.append((int(c), M, N, K))
        # the two forms alternate (the first timed launches of a process run at other clocks than the rest): median of 5 rounds
        bench(specs, True, warm=30)
        res = {True: [], False: []}
        for _ in range(5):
            for wdb in (True, False):
                res[wdb].append(bench(specs, wdb))
        for wdb in (True, False):
            us, tf = sorted(res[wdb])[2]
            print(f"{st:70s} db={int(wdb)}: {us:8.1f} us  {tf:7.1f} TFLOP/s   (min {min(r[0] for r in res[wdb]):.1f} max {max(r[0] for r in res[wdb]):.1f})")
