#!/usr/bin/env python3
"""Per-kernel mean of a rocprofv3 --pmc counter CSV.  usage: pmc_summary.py counter_collection.csv [top]"""
import collections, csv, sys
rows = csv.DictReader(open(sys.argv[1]))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r["Kernel_Name"][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
tot = {k: sum(sum(v) for v in c.values()) for k, c in agg.items()}
print(f"{'kernel':92s} {'counter':12s} {'launches':>8s} {'mean/launch':>14s} {'total':>14s}")
for k in sorted(agg, key=lambda k: -tot[k])[:top]:
    for c, v in agg[k].items():
        print(f"{k:92s} {c:12s} {len(v):8d} {sum(v)/len(v):14.1f} {sum(v):14.1f}")
