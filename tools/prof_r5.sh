# usage: bash tools/prof_r5.sh <tag>   -- kernel trace of the B = 64 step (and B = 16) -> gpurun_out/<tag>/r05_kernel_stats_b{64,16}.txt
T=${1:-r5p}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$T
rocprofv3 --kernel-trace --stats -d gpurun_out/$T/kt64 -o kt -- python3 bench.py --steps 36 --no-probes --no-cpu-baseline > gpurun_out/$T/kt64.log 2>&1
python3 tools/prof_summary.py $(ls gpurun_out/$T/kt64/*results.db | head -n 1) 70 > gpurun_out/$T/r05_kernel_stats_b64.txt
python3 tools/prof_steps.py $(ls gpurun_out/$T/kt64/*results.db | head -n 1) 6 >> gpurun_out/$T/r05_kernel_stats_b64.txt
python3 tools/prof_step_queues.py $(ls gpurun_out/$T/kt64/*results.db | head -n 1) > gpurun_out/$T/r05_step_queues_b64.txt 2>&1
for k in 1 2 3 4 5 6 7 8 9 10 11 12; do python3 tools/prof_step_queues.py $(ls gpurun_out/$T/kt64/*results.db | head -n 1) 1 --dump $k > gpurun_out/$T/step_dump_$k.txt 2>&1; done
rm -rf gpurun_out/$T/kt64
head -n 45 gpurun_out/$T/r05_kernel_stats_b64.txt; tail -n 7 gpurun_out/$T/r05_kernel_stats_b64.txt; head -n 12 gpurun_out/$T/r05_step_queues_b64.txt
