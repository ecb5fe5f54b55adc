cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5c
cp ab/libB.so vln_hamt_amd/libhamt_hip.so
python3 -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm" > gpurun_out/r5c/test.log 2>&1; tail -n 5 gpurun_out/r5c/test.log
for V in A B; do
cp ab/lib$V.so vln_hamt_amd/libhamt_hip.so
echo "== lib$V" >> gpurun_out/r5c/bench_epi.txt
GRAPH=1 python3 tools/gemm_bench.py nt:5120x3072x768:bias:bf16 nt:5120x3072x768:gelugrad:bf16 nn:5120x3072x768:none:bf16 nn:5120x3072x768:mulaux:bf16 \
   nt:5120x2304x768:bias:bf16 nt:5120x768x768:bias:bf16 nn:5120x768x768:none:bf16 nt:5120x768x3072:bias:bf16 nn:5120x768x3072:acc:f32 nn:5120x768x2304:acc:f32 \
   nt:11520x3072x768:gelugrad:bf16 nn:11520x3072x768:mulaux:bf16 nt:11520x768x3072:bias:bf16 nn:11520x768x3072:acc:f32 nt:2752x3072x768:gelugrad:bf16 nn:2752x768x3072:acc:f32 >> gpurun_out/r5c/bench_epi.txt 2>&1
done
cat gpurun_out/r5c/bench_epi.txt
bash tools/ab.sh 2 --no-cpu-baseline
