#!/usr/bin/env python3
"""The narrow-output (N = 768) and small-M GEMMs of the step under forced tile height / ring depth (HAMT_FAST_BM, HAMT_FAST_STAGES)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["GRAPH"] = "1"
os.environ.setdefault("HAMT_P8", "0")
from tools.gemm_bench import bench
B = 64
T, P, V = B * 80, B * 5 * 36, B * 43
specs = []
for M, tag in ((T, "text"), (P, "pano"), (V, "visn")):
    specs += [("nt", M, 768, 768, "bias", "f32", tag + " out"), ("nt", M, 768, 3072, "bias", "f32", tag + " ffn2"),
              ("nn", M, 768, 768, "none", "bf16", tag + " d_out"), ("nn", M, 768, 2304, "acc", "f32", tag + " d_qkv"), ("nn", M, 768, 3072, "acc", "f32", tag + " d_ffn1")]
specs += [("nt", V, 2304, 768, "bias", "bf16", "visn qkv"), ("nt", V, 1536, 768, "bias", "bf16", "visn kv"), ("nt", T, 1536, 768, "bias", "bf16", "text kv")]
print(f"# BM={os.environ.get('HAMT_FAST_BM','auto')} stages={os.environ.get('HAMT_FAST_STAGES','2')}")
for layout, M, N, K, epi, cdt, tag in specs:
    us, tf = bench(layout, M, N, K, epi, cdt)
    print(f"{tag:12s} {layout} {M:6d} {N:5d} {K:5d} {us:8.1f} us {tf:7.1f} TF/s", flush=True)
