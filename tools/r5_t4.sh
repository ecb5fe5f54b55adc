cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5i
python3 -m pytest tests/test_gpu_ops.py -q -m gpu -k "scatter or embed or bench_shapes" > gpurun_out/r5i/test_ops.log 2>&1; tail -n 4 gpurun_out/r5i/test_ops.log
python3 -m pytest tests/test_gpu_model.py -q -m gpu -x -k "two_ranks and wrapped" > gpurun_out/r5i/test_model.log 2>&1; tail -n 4 gpurun_out/r5i/test_model.log
for t in mlm sap itm sprel; do python3 tools/grad_bitwise_repeat.py $t 30 2>&1 | grep -v amdgpu.ids >> gpurun_out/r5i/bitwise.txt; done; cat gpurun_out/r5i/bitwise.txt
python3 bench.py --steps 48 --warmup 12 --no-probes --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('step', d['ms_per_step'], d['regions_ms_per_step'])"
