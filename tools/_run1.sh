python3 -m pytest tests -q -m gpu -x -k "attn or attention" > gpurun_out/t.log 2>&1; grep -n "passed\|failed" gpurun_out/t.log | tail -n 3
python3 tools/attn_bench.py 42,12,80,80 43,12,80,80 64,12,80,80 128,12,80,80 320,12,36,36 64,12,80,43 64,12,43,80 2>&1 | grep "bwd" | grep -v "p_drop 0.0 mask 1"
