cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4u; mkdir -p $O
export HAMT_GRAPH_SPLIT=1 HAMT_INTERLEAVE=1
for k in 1 2 3 5; do
  HAMT_BRANCH_SKIP=$k python3 bench.py --steps 48 --no-probes --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('skip $k b64', d['ms_per_step'], d['regions_ms_per_step'])" >> $O/bench.txt
done
for q in 2 8; do
  GPU_MAX_HW_QUEUES=$q python3 bench.py --steps 48 --no-probes --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('GPU_MAX_HW_QUEUES $q b64', d['ms_per_step'], d['regions_ms_per_step'])" >> $O/bench.txt
done
