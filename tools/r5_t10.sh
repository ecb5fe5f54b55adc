cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5n
for i in 1 2 3; do python3 -m pytest tests/test_gpu_model.py -q -m gpu -k "two_ranks and bf16" 2>&1 | tail -n 3; done
for i in 1 2; do
HAMT_XBIDIR=1 python3 bench.py --steps 48 --warmup 12 --no-probes --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('xbidir ', d['ms_per_step'], d['regions_ms_per_step'])"
python3 bench.py --steps 48 --warmup 12 --no-probes --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('default', d['ms_per_step'], d['regions_ms_per_step'])"
done
python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_robustness.py tests/test_vit.py -q -m gpu > gpurun_out/r5n/test_ops.log 2>&1; tail -n 3 gpurun_out/r5n/test_ops.log
python3 -m pytest tests/test_gpu_model.py -q -m gpu > gpurun_out/r5n/test_model.log 2>&1; grep "passed\|failed" gpurun_out/r5n/test_model.log
