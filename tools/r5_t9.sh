cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5m
rocm-smi --showuniqueid 2>/dev/null | grep -i "unique id" | head -1
python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_robustness.py tests/test_vit.py -q -m gpu > gpurun_out/r5m/test_ops.log 2>&1; tail -n 4 gpurun_out/r5m/test_ops.log
python3 -m pytest tests/test_gpu_model.py -q -m gpu > gpurun_out/r5m/test_model.log 2>&1; tail -n 4 gpurun_out/r5m/test_model.log
for t in mlm sap sprel mrc; do python3 tools/grad_bitwise_repeat.py $t 30 2>&1 | grep "^\[" ; done
