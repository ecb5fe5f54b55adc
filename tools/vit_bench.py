#!/usr/bin/env python3
"""ViT-B/16 backbone (row N3) throughput: the no-grad panorama pass (image_vilmodel.py:40-59: B*T*36 views) and a
forward+backward pass (observation / history images), eager launches, bf16 path.  usage: vit_bench.py [n_images]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vln_hamt_amd.model.vision_transformer import vit_base_patch16_224

n = int(sys.argv[1]) if len(sys.argv) > 1 else 144
torch.manual_seed(0)
model = vit_base_patch16_224(hamt_precision="bf16").cuda()
x = torch.randn(n, 3, 224, 224, device="cuda")
FWD_GF = 2 * (197 * (768 * 2304 + 768 * 768 + 2 * 768 * 3072) + 2 * 197 * 197 * 768) * 12 / 1e9 + 2 * 196 * 768 * 768 / 1e9

def t(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps

def fwd():
    with torch.no_grad():
        model.eval(); model.forward_features(x)
def fb():
    model.train(); model.forward_features(x).sum().backward(); model.zero_grad(set_to_none=True)
d = t(fwd); print(f"no-grad forward: {n} images in {d*1e3:7.1f} ms = {n/d:8.1f} images/s, {n*FWD_GF/d/1e3:6.1f} TFLOP/s ({FWD_GF:.1f} GFLOP/image)")
d = t(fb); print(f"forward+backward: {n} images in {d*1e3:7.1f} ms = {n/d:8.1f} images/s, {3*n*FWD_GF/d/1e3:6.1f} TFLOP/s")
