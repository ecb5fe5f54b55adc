cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5k
rocm-smi --showuniqueid 2>/dev/null | grep -i "unique id" | head -2
cp vln_hamt_amd/libhamt_hip.so /tmp/libNew.so
for V in A New A New; do
  if [ $V = A ]; then cp ab/libA.so vln_hamt_amd/libhamt_hip.so; else cp /tmp/libNew.so vln_hamt_amd/libhamt_hip.so; fi
  python3 -m pytest tests/test_gpu_ops.py -q -m gpu -k "bench_shapes and (4096x4096 or 2318 or drop_res or 11520x3072 or 2752x3072)" 2>&1 | tail -n 3 | sed "s/^/lib$V: /"
done
cp /tmp/libNew.so vln_hamt_amd/libhamt_hip.so
dmesg 2>/dev/null | tail -5
