mkdir -p gpurun_out/r6f
python -m pytest tests/test_gpu_model.py -x -q -s -k "test_eight_ranks and fp32-wrapped-False-1" > gpurun_out/r6f/r8.txt 2>&1; grep "eight ranks\|exp_avg off\|passed\|failed" gpurun_out/r6f/r8.txt | cut -c1-200
python -m pytest tests/test_gpu_model.py -q -k "test_two_ranks and wrapped" > gpurun_out/r6f/r2.txt 2>&1; tail -3 gpurun_out/r6f/r2.txt
python -m pytest tests/test_gpu_model.py -q -k "test_eight_ranks" > gpurun_out/r6f/r8all.txt 2>&1; tail -6 gpurun_out/r6f/r8all.txt
for i in 1 2 3; do python -m pytest tests/test_gpu_robustness.py -q -k "test_update_overlapped_with_the_next_forward" 2>&1 | grep "assert 0\|passed\|failed"; done
HAMT_DENSE_OUT_BF16=1 python -m pytest tests/test_gpu_robustness.py -q -k "test_update_overlapped_with_the_next_forward" 2>&1 | grep "assert 0\|passed\|failed"
python -m pytest tests/test_gpu_ops.py -q -x > gpurun_out/r6f/ops.txt 2>&1; tail -2 gpurun_out/r6f/ops.txt
for v in "A=1" "HAMT_LN_NO_NT=1" "HAMT_NO_NT_AUX=1" "HAMT_LN_NO_NT=1 HAMT_NO_NT_AUX=1" "A=2"; do
  env $v python bench.py --no-probes --no-cpu-baseline --steps 48 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$v', d['regions_ms_per_step'])"
done
