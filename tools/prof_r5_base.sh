cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5a
python3 tools/gemm_sweep.py --batch 64 --blas > gpurun_out/r5a/gemm_sweep_b64.txt 2>&1
python3 bench.py --steps 24 --warmup 12 --no-probes --no-cpu-baseline > gpurun_out/r5a/bench.log 2>&1
tail -n 3 gpurun_out/r5a/bench.log | cut -c1-600
cat gpurun_out/r5a/gemm_sweep_b64.txt
