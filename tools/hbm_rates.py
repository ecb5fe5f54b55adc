#!/usr/bin/env python3
"""Achieved HBM rate per kernel = (2 x FETCH_SIZE + WRITE_SIZE) per launch (rocprofv3 --pmc passes summarised by
pmc_summary.py; KB; x2 is the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md) / mean launch duration (kernel-trace
summary of prof_summary.py).  usage: hbm_rates.py pmc_fetch.txt pmc_write.txt kernel_stats.txt [--json out.json]
--json also writes {kernel: {read_bytes, write_bytes, avg_us}} per launch (bench.py's `roofline.traffic` reads it)."""
import json, re, sys
from prof_summary import short

def pmc(path, counter):
    out = {}
    for l in open(path):
        m = re.match(r"(.+?)\s+%s\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s*$" % counter, l)
        if m:
            out.setdefault(short(m.group(1).strip()), float(m.group(3)))
    return out

fetch, write = pmc(sys.argv[1], "FETCH_SIZE"), pmc(sys.argv[2], "WRITE_SIZE")
print(f"# (2 x FETCH_SIZE + WRITE_SIZE) KB per launch / mean duration; peak 8 000 GB/s (≈6 300 achievable, MI355X_MICROARCH.md)")
print(f"{'kernel':72s} {'avg_us':>8s} {'read MB':>9s} {'write MB':>9s} {'GB/s':>8s} {'of 8 TB/s':>9s}")
table = {}
for l in open(sys.argv[3]):
    m = re.match(r"(.{90})\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", l)
    if not m:
        continue
    k, us = m.group(1).strip(), float(m.group(4))
    key = next((q for q in fetch if q.startswith(k[:40]) or k.startswith(q[:40])), None)
    if key is None:
        continue
    rd, wr = 2 * fetch[key] / 1024, write.get(key, 0.0) / 1024
    gbs = (rd + wr) / 1024 / (us * 1e-6)
    print(f"{k[:72]:72s} {us:8.1f} {rd:9.1f} {wr:9.1f} {gbs:8.0f} {100*gbs/8000:8.1f}%")
    table[k] = {"read_bytes": int(rd * 2 ** 20), "write_bytes": int(wr * 2 ** 20), "avg_us": us}
if "--json" in sys.argv:
    json.dump(table, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
