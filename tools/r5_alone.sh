cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5s
rocprofv3 --kernel-trace -d gpurun_out/r5s/kt -o kt -- python3 bench.py --steps 36 --no-probes --no-cpu-baseline > gpurun_out/r5s/kt.log 2>&1
DB=$(ls gpurun_out/r5s/kt/*results.db | head -n 1)
cd tools
python3 prof_alone.py ../$DB 24 60 > ../gpurun_out/r5s/alone.txt
python3 prof_step_queues.py ../$DB 12 --kinds > ../gpurun_out/r5s/queues.txt
python3 prof_step_queues.py ../$DB 1 --dump 12 > ../gpurun_out/r5s/dump_m12.txt
python3 prof_step_queues.py ../$DB 1 --dump 11 > ../gpurun_out/r5s/dump_m11.txt
python3 prof_step_queues.py ../$DB 1 --dump 7 > ../gpurun_out/r5s/dump_m7.txt
cd ..
rm -rf gpurun_out/r5s/kt
head -n 70 gpurun_out/r5s/alone.txt
