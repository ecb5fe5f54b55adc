cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5g
cp ab/libB.so vln_hamt_amd/libhamt_hip.so
python3 -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm" > gpurun_out/r5g/test.log 2>&1; tail -n 4 gpurun_out/r5g/test.log
bash tools/ab.sh 3 --no-cpu-baseline
cp ab/libB.so vln_hamt_amd/libhamt_hip.so
