cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5t
rocprofv3 --kernel-trace -d gpurun_out/r5t/kt -o kt -- python3 bench.py --steps 36 --batch 16 --no-probes --no-cpu-baseline > gpurun_out/r5t/kt.log 2>&1
DB=$(ls gpurun_out/r5t/kt/*results.db | head -n 1)
cd tools
python3 prof_alone.py ../$DB 24 50 > ../gpurun_out/r5t/alone.txt
python3 prof_step_queues.py ../$DB 12 --kinds > ../gpurun_out/r5t/queues.txt
python3 prof_step_queues.py ../$DB 1 --dump 12 > ../gpurun_out/r5t/dump_m12.txt
python3 prof_step_queues.py ../$DB 1 --dump 11 > ../gpurun_out/r5t/dump_m11.txt
python3 prof_summary.py ../$DB 45 > ../gpurun_out/r5t/stats.txt
cd ..
rm -rf gpurun_out/r5t/kt
head -n 40 gpurun_out/r5t/stats.txt
