cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5b
python3 -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gelugrad or mulaux" > gpurun_out/r5b/test.log 2>&1; tail -n 5 gpurun_out/r5b/test.log
GRAPH=1 python3 tools/gemm_bench.py nt:5120x3072x768:bias:bf16 nt:5120x3072x768:gelugrad:bf16 nt:5120x3072x768:gelugrad8:bf16 nn:5120x3072x768:none:bf16 nn:5120x3072x768:mulaux:bf16 nn:5120x3072x768:mulaux8:bf16 \
   nt:11520x3072x768:bias:bf16 nt:11520x3072x768:gelugrad:bf16 nt:11520x3072x768:gelugrad8:bf16 nn:11520x3072x768:mulaux:bf16 nn:11520x3072x768:mulaux8:bf16 \
   nt:2752x3072x768:gelugrad:bf16 nt:2752x3072x768:gelugrad8:bf16 nn:2752x3072x768:mulaux:bf16 nn:2752x3072x768:mulaux8:bf16 > gpurun_out/r5b/bench_epi.txt 2>&1
cat gpurun_out/r5b/bench_epi.txt
for i in 1 2; do
HAMT_GELUP_BF16=1 python3 bench.py --steps 48 --warmup 12 --no-probes --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('bf16 gelup', d['ms_per_step'], d['regions_ms_per_step'])"
python3 bench.py --steps 48 --warmup 12 --no-probes --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('u8 gelup  ', d['ms_per_step'], d['regions_ms_per_step'])"
done
