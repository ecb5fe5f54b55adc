cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5f; rm -f gpurun_out/r5f/q4w.txt
for v in 0 6; do echo "== HAMT_Q4=1 HAMT_Q4_VAR=$v" >> gpurun_out/r5f/q4w.txt; HAMT_Q4=1 HAMT_Q4_VAR=$v timeout 600 python3 tools/q4_probe.py nt:5120x768x768:bias:bf16 nt:5120x768x3072:bias:bf16 nt:2752x768x3072:bias:bf16 nt:5003x760x832:bias:f32 nt:5120x1536x768:bias:bf16 >> gpurun_out/r5f/q4w.txt 2>&1; done
cat gpurun_out/r5f/q4w.txt
