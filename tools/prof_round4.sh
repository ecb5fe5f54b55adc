cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4p; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/kt64 -o kt -- python3 bench.py --steps 36 --no-probes > $O/kt64.log 2>&1
DB=$(ls $O/kt64/*results.db | head -n 1)
cd tools
python3 prof_summary.py ../$DB 70 > ../$O/kernel_stats_b64.txt
python3 prof_steps.py ../$DB 12 >> ../$O/kernel_stats_b64.txt
python3 prof_step_queues.py ../$DB 12 --kinds > ../$O/queues_b64.txt
for k in 1 2 3 4 5 6 7 8 9 10 11 12; do python3 prof_step_queues.py ../$DB 0 --dump $k > ../$O/step_b64_m$k.txt; done
cd ..
rocprofv3 --kernel-trace --stats -d $O/kt16 -o kt -- python3 bench.py --steps 36 --batch 16 --no-probes > $O/kt16.log 2>&1
DB=$(ls $O/kt16/*results.db | head -n 1)
cd tools
python3 prof_summary.py ../$DB 50 > ../$O/kernel_stats_b16.txt
python3 prof_step_queues.py ../$DB 12 --kinds > ../$O/queues_b16.txt
for k in 1 2 3 4 5 6 7 8 9 10 11 12; do python3 prof_step_queues.py ../$DB 0 --dump $k > ../$O/step_b16_m$k.txt; done
cd ..
rm -rf $O/kt64 $O/kt16
ls -la $O
