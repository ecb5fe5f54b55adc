cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5u
rocprofv3 --kernel-trace -d gpurun_out/r5u/kt -o kt -- python3 bench.py --steps 24 --task itm --no-probes --no-cpu-baseline > gpurun_out/r5u/kt.log 2>&1
DB=$(ls gpurun_out/r5u/kt/*results.db | head -n 1)
cd tools
python3 prof_step_queues.py ../$DB 4 --kinds > ../gpurun_out/r5u/queues.txt
python3 prof_step_queues.py ../$DB 1 --dump 2 > ../gpurun_out/r5u/dump.txt
python3 prof_summary.py ../$DB 40 > ../gpurun_out/r5u/stats.txt
cd ..
rm -rf gpurun_out/r5u/kt
cat gpurun_out/r5u/queues.txt
