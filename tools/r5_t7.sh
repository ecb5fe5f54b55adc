cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocm-smi --showuniqueid 2>/dev/null | grep -i "unique id" | head -1
python3 -m pytest tests/test_gpu_ops.py -q -m gpu -k "scatter" 2>&1 | tail -n 2
bash tools/prof_r5.sh r5l > gpurun_out/r5l.log 2>&1
grep -n "scatter\|embed_sum\|reduce_partials\|colsum" gpurun_out/r5l/r05_kernel_stats_b64.txt
head -n 24 gpurun_out/r5l/r05_kernel_stats_b64.txt
tail -n 14 gpurun_out/r5l/r05_kernel_stats_b64.txt
