#!/usr/bin/env python3
"""Cycle accounting of the GEMM main loop (where a wave's time goes: waiting for the DMA of the next k-tile, at the
workgroup barrier, issuing DMA, reading fragments + issuing MFMAs).  Needs a private build of the library with the
instrumentation compiled in -- never the shipped one:

    rm vln_hamt_amd/csrc/build/gemm_fast.o; HAMT_EXTRA_FLAGS=-DHAMT_PROF python vln_hamt_amd/csrc/build.py
    python tools/gemm_prof.py            # then rebuild without the flag

Reports per-wave average shader cycles per k-tile for a few shapes / tile heights and for the grouped weight-gradient
kernel on the problem list of one SAP backward pass."""
import ctypes as C, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vln_hamt_amd import _lib as L, ops

lib = L.load()
if not hasattr(lib, "hamt_prof_fetch"):
    sys.exit("library was built without -DHAMT_PROF")
buf = (C.c_ulonglong * 16)()

def fetch(reset=True):
    lib.hamt_prof_fetch(buf, int(reset))
    return list(buf)

def report(tag):
    w, b, d, m, tot, waves, nk = fetch()[:7]
    if not waves:
        print(tag, "no data"); return
    per = lambda x: x / max(nk, 1)
    print(f"{tag:58s} waves {waves:7d} k-tiles/wave {nk/waves:6.1f} | per k-tile cycles: dma-wait {per(w):7.0f}  barrier {per(b):7.0f}  "
          f"dma-issue {per(d):6.0f}  frag+mfma {per(m):7.0f}  | loop total/wave {tot/waves:9.0f}")

def gemm(M, N, K, bm, layout="nt", iters=5):
    os.environ["HAMT_FAST_BM"] = str(bm)
    code = f"""
import os, sys, ctypes as C
sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r})
import torch
from vln_hamt_amd import _lib as L, ops
lib = L.load(); buf = (C.c_ulonglong * 16)()
a = torch.randn({M}, {K}, device="cuda").bfloat16(); b = torch.randn({N}, {K}, device="cuda").bfloat16()
out = torch.empty({M}, {N}, device="cuda", dtype=torch.bfloat16)
ops.gemm(a, b, out); lib.hamt_prof_fetch(buf, 1)
for _ in range({iters}): ops.gemm(a, b, out)
lib.hamt_prof_fetch(buf, 1); print(" ".join(str(x) for x in buf))
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ))
    vals = [int(x) for x in r.stdout.strip().split()[-16:]]
    w, b_, d, m, tot, waves, nk, pro, epi = vals[:9]
    per = lambda x: x / max(nk, 1)
    print(f"nt {M}x{N}x{K} BM={bm:3d}: waves {waves:7d} k-tiles/wave {nk/max(waves,1):6.1f} | per k-tile cycles: dma-wait {per(w):7.0f}  barrier {per(b_):7.0f}  "
          f"dma-issue {per(d):6.0f}  frag+mfma {per(m):7.0f}  | per wave: prologue {pro/max(waves,1):7.0f}  loop {tot/max(waves,1):9.0f}  epilogue {epi/max(waves,1):8.0f}")

for shp in [(4096, 4096, 4096), (5120, 3072, 768), (5120, 2304, 768)]:
    for bm in (256, 128, 64):
        gemm(*shp, bm)
