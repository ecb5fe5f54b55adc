cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
HAMT_ATOMIC_SCATTER=1 python3 bench.py --steps 48 --warmup 12 --no-probes --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('atomic ', d['ms_per_step'], d['regions_ms_per_step'])"
python3 bench.py --steps 48 --warmup 12 --no-probes --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('ordered', d['ms_per_step'], d['regions_ms_per_step'])"
done
