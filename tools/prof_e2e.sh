#!/bin/bash
# rocprofv3 kernel-trace summary of the image-input step (BASELINE config 4, B = 1, hipGraph replay): tools/e2e_bench.py 1 12 graph
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r4e; mkdir -p $out
timeout 600 rocprofv3 --kernel-trace --stats -d $out/prof -o e2e -- python3 $R/tools/e2e_bench.py 1 12 graph > $out/e2e.log 2>&1
f=$(find $out/prof -name "*kernel_stats.csv" | head -1)
python3 - "$f" > $out/e2e_kernel_stats.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"# {len(rows)} distinct kernels, total {tot/1e6:.2f} ms")
print(f"{'kernel':90s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'pct':>6s}")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:45]:
    print(f"{r['Name'][:90]:90s} {int(r['Calls']):7d} {float(r['TotalDurationNs'])/1e6:10.3f} {float(r['AverageNs'])/1e3:9.2f} {100*float(r['TotalDurationNs'])/tot:6.2f}")
PY
tail -3 $out/e2e.log; head -40 $out/e2e_kernel_stats.txt
