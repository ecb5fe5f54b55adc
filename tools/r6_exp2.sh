mkdir -p gpurun_out/r6d
T="python -m pytest tests/test_gpu_model.py -x -q -s -k test_overlapped_grad_sync_matches_single_process_steps"
for v in "A=1" "HAMT_NO_GROUP_GRAPHS=1" "HAMT_ATOMIC_SCATTER=1" "AMD_SERIALIZE_KERNEL=3"; do
  env $v $T > gpurun_out/r6d/ov_$v.txt 2>&1; echo "$v -> $(grep -c 'passed' gpurun_out/r6d/ov_$v.txt) $(grep -m1 'aborting' gpurun_out/r6d/ov_$v.txt | cut -c1-200)"
done
python -m pytest tests/test_gpu_ops.py -x -q > gpurun_out/r6d/ops.txt 2>&1; tail -3 gpurun_out/r6d/ops.txt
python -m pytest tests/test_gpu_model.py -x -q -s -k "test_canon_b64_vs_oracle and packed" > gpurun_out/r6d/b64_f16.txt 2>&1; grep "outputs\|canon B=64\|passed\|failed" gpurun_out/r6d/b64_f16.txt
python bench.py --no-probes --no-cpu-baseline --steps 48 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('f16 bench', d['regions_ms_per_step'])"
