#!/usr/bin/env python3
"""Build libhamt_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python vln_hamt_amd/csrc/build.py [--force]

Objects go to csrc/build/, the library next to the Python package (vln_hamt_amd/libhamt_hip.so), so it
travels with the tree to the GPU box.  Re-compiles only sources newer than their object.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
ROOT = os.path.dirname(PKG)
OUT = os.path.join(PKG, "libhamt_hip.so")
SOURCES = ["abi.hip", "gemm.hip", "gemm_fast.hip", "gemm_q4.hip", "attn.hip", "attn16.hip", "norm.hip", "vis_embed.hip", "elementwise.hip", "loss.hip", "optim.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-comment"] + os.environ.get("HAMT_EXTRA_FLAGS", "").split()


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def build(force=False, verbose=True):
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    deps = [os.path.join(HERE, "common.h"), os.path.join(HERE, "gemm_frag.h"), os.path.join(HERE, "gemm_epi.h"), os.path.join(HERE, "gemm_args.h"), os.path.join(ROOT, "include", "hamt.h")]
    dep_m = max(os.path.getmtime(d) for d in deps)
    jobs = []
    for s in SOURCES:
        src, obj = os.path.join(HERE, s), os.path.join(HERE, "build", s.replace(".hip", ".o"))
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), dep_m):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        r = subprocess.run([hipcc(), *FLAGS, "-c", src, "-o", obj], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr[-4000:]}")
        if verbose:
            print("compiled", os.path.basename(src), flush=True)

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(cc, jobs))
    objs = [os.path.join(HERE, "build", s.replace(".hip", ".o")) for s in SOURCES]
    if jobs or not os.path.exists(OUT):
        r = subprocess.run([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT, *objs], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
        if verbose:
            print("linked", OUT, flush=True)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
