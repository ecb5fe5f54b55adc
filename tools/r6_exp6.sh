cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6h; mkdir -p $O
python -m pytest tests/test_gpu_model.py tests/test_gpu_robustness.py -q -x --durations=8 -k "two_ranks or eight_ranks or exchange_schedule or overlapped_grad_sync or two_rank_bench or canon_b64" > $O/tests.txt 2>&1; tail -14 $O/tests.txt | cut -c1-160
for v in "A=1" "HAMT_GROUPS_EQUAL_WORK=1" "A=2" "HAMT_GROUPS_EQUAL_WORK=1"; do
  env $v HAMT_BENCH_NO_EXTRA=1 python bench.py --steps 48 --also-batch 0 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$v', d['regions_ms_per_step'], 'w1', d.get('world1_exchange_ms_per_step'), 'overhead', d.get('exchange_overhead_ms_world1'))"
done
HAMT_FORCE_DIST=1 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python3 bench.py --steps 24 --no-probes --no-cpu-baseline > $O/kt.log 2>&1
DB=$(ls $O/kt/*results.db | head -n 1)
python3 tools/prof_timeline.py $DB 2 1200 2600 > $O/w1_timeline.txt
rm -rf $O/kt
grep -v "^#" $O/w1_timeline.txt | awk '$1 > -1600 && $1 < 1100' | cut -c1-120
