cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4v; mkdir -p $O
export HAMT_GRAPH_SPLIT=1
for il in 0 1; do
  HAMT_INTERLEAVE=$il timeout 200 python3 bench.py --steps 48 --no-probes --no-cpu-baseline 2>$O/err.log | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('values interleave=$il b64', d['ms_per_step'], d['regions_ms_per_step'], d['state_finite_after_timed_region'])" >> $O/bench.txt
  HAMT_INTERLEAVE=$il timeout 200 python3 bench.py --steps 48 --batch 16 --no-probes --no-cpu-baseline 2>>$O/err.log | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('values interleave=$il b16', d['ms_per_step'], d['regions_ms_per_step'])" >> $O/bench.txt
done
HAMT_GRAPH_SPLIT=0 HAMT_INTERLEAVE=0 timeout 200 python3 bench.py --steps 48 --no-probes --no-cpu-baseline 2>>$O/err.log | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('unsplit b64', d['ms_per_step'], d['regions_ms_per_step'])" >> $O/bench.txt
HAMT_GRAPH_SPLIT=0 HAMT_INTERLEAVE=0 timeout 200 python3 bench.py --steps 48 --batch 16 --no-probes --no-cpu-baseline 2>>$O/err.log | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('unsplit b16', d['ms_per_step'], d['regions_ms_per_step'])" >> $O/bench.txt
export HAMT_INTERLEAVE=1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt64 -o kt -- python3 bench.py --steps 24 --no-probes --no-cpu-baseline > $O/kt64.log 2>&1
DB=$(ls $O/kt64/*results.db | head -n 1)
cd tools
python3 prof_step_queues.py ../$DB 12 --kinds > ../$O/queues_b64.txt
python3 prof_step_queues.py ../$DB 0 --dump 1 > ../$O/step_b64_m1.txt
python3 prof_bins.py ../$O/step_b64_m1.txt 200 > ../$O/bins_b64_m1.txt
cd ..
rm -rf $O/kt64
