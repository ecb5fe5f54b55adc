cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5e
python3 -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm or wgrad" > gpurun_out/r5e/test.log 2>&1; tail -n 5 gpurun_out/r5e/test.log
python3 -m pytest tests/test_gpu_model.py -x -q -m gpu -k "b64 or soak or canon_pretrain" > gpurun_out/r5e/test2.log 2>&1; tail -n 5 gpurun_out/r5e/test2.log
bash tools/prof_r5.sh r5e > gpurun_out/r5e/prof.log 2>&1
tail -n 8 gpurun_out/r5e/prof.log
python3 bench.py --steps 48 --warmup 12 --no-probes --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('step', d['ms_per_step'], d['regions_ms_per_step'])"
