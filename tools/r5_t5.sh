cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5j
rocm-smi --showuniqueid 2>/dev/null | grep -i "unique id" | head -2
python3 -m pytest tests/test_gpu_ops.py -q -m gpu -k "scatter or embed or bench_shapes" > gpurun_out/r5j/test_ops.log 2>&1; tail -n 6 gpurun_out/r5j/test_ops.log
python3 -m pytest tests/test_gpu_model.py -q -m gpu -x -k "two_ranks and wrapped" > gpurun_out/r5j/test_model.log 2>&1; tail -n 4 gpurun_out/r5j/test_model.log
python3 bench.py --steps 48 --warmup 12 --no-probes --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('step', d['ms_per_step'], d['regions_ms_per_step'])"
