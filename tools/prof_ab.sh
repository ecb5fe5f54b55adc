# usage: prof_ab.sh OUTDIR "ENV_A" "ENV_B" : alternating bench runs of two settings on ONE box (B = 64 and B = 16), then a timeline of setting B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
for rep in 1 2; do
  for v in "$2" "$3"; do
    env $v timeout 200 python3 bench.py --steps 48 --no-probes --no-cpu-baseline 2>>$O/err.log | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$v] b64', d['ms_per_step'], d['regions_ms_per_step'], d['state_finite_after_timed_region'])" >> $O/bench.txt
    env $v timeout 200 python3 bench.py --steps 48 --batch 16 --no-probes --no-cpu-baseline 2>>$O/err.log | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$v] b16', d['ms_per_step'], d['regions_ms_per_step'])" >> $O/bench.txt
  done
done
env $3 timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt64 -o kt -- python3 bench.py --steps 24 --no-probes --no-cpu-baseline > $O/kt64.log 2>&1
DB=$(ls $O/kt64/*results.db | head -n 1)
cd tools
python3 prof_step_queues.py ../$DB 12 --kinds > ../$O/queues_b64.txt
python3 prof_step_queues.py ../$DB 0 --dump 1 > ../$O/step_b64_m1.txt
python3 prof_bins.py ../$O/step_b64_m1.txt 200 > ../$O/bins_b64_m1.txt
cd ..
rm -rf $O/kt64
cat $O/bench.txt
