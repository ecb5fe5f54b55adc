cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5h
python3 -m pytest tests/test_gpu_ops.py -q -m gpu > gpurun_out/r5h/test_ops.log 2>&1; tail -n 4 gpurun_out/r5h/test_ops.log
python3 -m pytest tests/test_gpu_model.py -q -m gpu -x > gpurun_out/r5h/test_model.log 2>&1; tail -n 4 gpurun_out/r5h/test_model.log
