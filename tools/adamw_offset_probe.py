#!/usr/bin/env python3
"""Does the relative placement of the p / g / m / v arenas matter to the update kernel?  One 175 M-element parameter, arenas carved out of
one allocation at chosen relative offsets (bytes); hamt_adamw_table timed with HIP events.  (The four read streams and four write streams of a
block walk the same element range: with identical offsets into equally aligned allocations every access of a wave maps to the same HBM channel.)"""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vln_hamt_amd import _lib as L
dev = torch.device("cuda", 0)
n = 174_800_000 // 2048 * 2048
lib = L.load()
ends = torch.tensor([n], dtype=torch.int32, device=dev)
hyp = torch.tensor([[5e-5, 5e-5, 0.01, 2.0]], dtype=torch.float32, device=dev)
gn = torch.ones(1, device=dev)
def run(offs, iters=10):
    pad = 64 << 20
    big = torch.empty((4 * n * 4 + 5 * pad) // 4, dtype=torch.float32, device=dev)
    big.normal_()
    base = (big.data_ptr() + (1 << 21) - 1) // (1 << 21) * (1 << 21)          # 2 MiB aligned
    ptrs = [base + i * ((n * 4 + (1 << 21) - 1) // (1 << 21) * (1 << 21) + (8 << 20)) + offs[i] for i in range(4)]
    p16 = torch.empty(n, dtype=torch.bfloat16, device=dev)
    big[: n * 4 + 4].abs_()
    def go():
        L.check(lib.hamt_adamw_table(n, C.c_void_p(ptrs[0]), C.c_void_p(ptrs[1]), C.c_void_p(ptrs[2]), C.c_void_p(ptrs[3]), C.c_void_p(p16.data_ptr()),
                                     C.c_void_p(ends.data_ptr()), C.c_void_p(hyp.data_ptr()), 1, C.c_void_p(gn.data_ptr()), 5.0, 0.9, 0.98, 1e-6, 0,
                                     C.c_void_p(torch.cuda.current_stream().cuda_stream)), "adamw")
    for _ in range(3): go()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): go()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / iters * 1e3
    del big
    return us
for name, offs in [("aligned (0,0,0,0)", (0, 0, 0, 0)), ("+256 B steps", (0, 256, 512, 768)), ("+1 KiB steps", (0, 1024, 2048, 3072)),
                   ("+4 KiB steps", (0, 4096, 8192, 12288)), ("+16 KiB steps", (0, 16384, 32768, 49152)), ("+64 KiB steps", (0, 65536, 131072, 196608)),
                   ("+256 KiB steps", (0, 1 << 18, 2 << 18, 3 << 18)), ("+1 MiB steps", (0, 1 << 20, 2 << 20, 3 << 20)), ("aligned again", (0, 0, 0, 0))]:
    us = run(offs)
    print(f"{name:22s} {us:8.1f} us  {30.0 * n / us / 1e6:6.2f} TB/s", flush=True)
