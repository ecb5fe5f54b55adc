#!/usr/bin/env python3
"""Split-K sweep of the narrow (N = 768) GEMMs of a step: K slices (HAMT_KS) x tile height (HAMT_FAST_BM), each configuration in its
own process (both variables are read once), hipGraph-replayed chains, with a correctness check against an fp32 matmul of the same
bf16 operands.  HAMT_KFIX=0 in the environment measures the two-pass form (partials + reduce kernel).
usage: ksplit_sweep.py [--batch 16] [--ks 1,2,3,4,6,8] [--bm 0,64,128]"""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def shapes(B):
    T, P, V = B * 80, B * 5 * 36, B * 43
    out = []
    for M, tag in ((T, "text"), (V, "visn"), (T + V, "cross"), (P, "pano")):
        out += [("nt", M, 768, 768, "bias", "f32", tag + " out"), ("nt", M, 768, 3072, "bias", "f32", tag + " ffn2"),
                ("nn", M, 768, 768, "none", "bf16", tag + " d_out"), ("nn", M, 768, 2304, "acc", "f32", tag + " d_qkv"),
                ("nn", M, 768, 3072, "acc", "f32", tag + " d_ffn1")]
    return out


def worker(batch):
    os.environ.setdefault("GRAPH", "1")
    import torch
    from tools.gemm_bench import bench
    from vln_hamt_amd import ops
    for layout, M, N, K, epi, cdt, tag in shapes(batch):
        torch.manual_seed(0)
        A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        Bm = (torch.randn(K, N, device="cuda") * 0.05).to(torch.bfloat16)
        b = Bm.t().contiguous() if layout == "nt" else Bm
        out = torch.zeros(M, N, device="cuda", dtype=torch.float32 if cdt == "f32" else torch.bfloat16)
        bias = torch.randn(N, device="cuda") if epi == "bias" else None
        ops.gemm(A, b, out, b_kmajor=layout == "nn", bias=bias)
        ref = A.float() @ Bm.float() + (bias if bias is not None else 0.0)
        err = float((out.float() - ref).abs().max() / ref.abs().max())
        us, tf = bench(layout, M, N, K, epi, cdt)
        print(f"R {tag}|{M}|{K}|{us:.1f}|{err:.1e}", flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--ks", default="0,1,2,3,4,6,8")
    ap.add_argument("--bm", default="0,64,128")
    ap.add_argument("--kg", default="", help="comma list of HAMT_KG variants (10 * tile rows + groups), e.g. 322,323,642,643: sweeps these instead")
    ap.add_argument("--worker", action="store_true")
    a = ap.parse_args()
    if a.worker:
        worker(a.batch)
        sys.exit(0)
    cols, table, errs = [], {}, {}
    configs = [(bm, ks, "") for bm in a.bm.split(",") for ks in a.ks.split(",")]
    if a.kg:
        configs = [("0", "0", "")] + [("0", "0", kg) for kg in a.kg.split(",")]
    for bm, ks, kg in configs:
        if True:
            env = dict(os.environ)
            if kg:
                env["HAMT_KG"] = kg
            if int(ks):
                env["HAMT_KS"] = ks
            if int(bm):
                env["HAMT_FAST_BM"] = bm
            env["HAMT_P8"] = env.get("HAMT_P8", "0") if int(ks) else env.get("HAMT_P8", "-1")
            if env["HAMT_P8"] == "-1":
                env.pop("HAMT_P8")
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", "--batch", str(a.batch)], env=env, capture_output=True, text=True)
            col = f"kg{kg}" if kg else f"bm{bm}/ks{ks}"
            cols.append(col)
            for line in r.stdout.splitlines():
                if line.startswith("R "):
                    tag, M, K, us, err = line[2:].split("|")
                    table.setdefault((tag, M, K), {})[col] = float(us)
                    errs[(tag, M, K)] = max(errs.get((tag, M, K), 0.0), float(err))
            if r.returncode:
                print(f"# {col}: exit {r.returncode}\n{r.stderr[-600:]}")
    print(f"# B={a.batch} KFIX={os.environ.get('HAMT_KFIX', '1')}   (ks0 = the library's own choice; us per launch)")
    print(f"{'shape':14s} {'M':>6s} {'K':>5s} " + " ".join(f"{c:>10s}" for c in cols) + "   max_err  best")
    for key, row in table.items():
        best = min(row, key=row.get)
        print(f"{key[0]:14s} {key[1]:>6s} {key[2]:>5s} " + " ".join(f"{row.get(c, float('nan')):10.1f}" for c in cols) + f"   {errs[key]:.1e}  {best}")
