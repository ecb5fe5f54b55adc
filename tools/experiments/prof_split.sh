cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s; mkdir -p $O
export HAMT_GRAPH_SPLIT=1 HAMT_INTERLEAVE=1 HAMT_GRAPH_SPLIT_VERBOSE=1
python3 bench.py --steps 12 --no-probes --no-cpu-baseline > $O/plain.log 2> $O/plain.err
grep "graph split" $O/plain.err | head -14 | cut -c1-1500 > $O/split_info.txt
rocprofv3 --kernel-trace --stats -d $O/kt64 -o kt -- python3 bench.py --steps 36 --no-probes --no-cpu-baseline > $O/kt64.log 2>&1
DB=$(ls $O/kt64/*results.db | head -n 1)
cd tools
python3 prof_step_queues.py ../$DB 12 --kinds > ../$O/queues_b64.txt
python3 prof_step_queues.py ../$DB 0 --dump 1 > ../$O/step_b64_m1.txt
python3 prof_bins.py ../$O/step_b64_m1.txt 200 > ../$O/bins_b64_m1.txt
cd ..
rocprofv3 --kernel-trace --stats -d $O/kt16 -o kt -- python3 bench.py --steps 36 --batch 16 --no-probes --no-cpu-baseline > $O/kt16.log 2>&1
DB=$(ls $O/kt16/*results.db | head -n 1)
cd tools
python3 prof_step_queues.py ../$DB 12 --kinds > ../$O/queues_b16.txt
python3 prof_step_queues.py ../$DB 0 --dump 1 > ../$O/step_b16_m1.txt
python3 prof_bins.py ../$O/step_b16_m1.txt 100 > ../$O/bins_b16_m1.txt
cd ..
rm -rf $O/kt64 $O/kt16
