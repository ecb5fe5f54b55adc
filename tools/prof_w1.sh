# timeline of the one-rank sharded exchange (what the multi-GPU step costs before a byte moves): bash tools/prof_w1.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/w1
mkdir -p $O
HAMT_FORCE_DIST=1 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python3 bench.py --steps 24 --no-probes --no-cpu-baseline > $O/kt.log 2>&1
DB=$(ls $O/kt/*results.db | head -n 1)
python3 tools/prof_steps.py $DB 6 > $O/r03_w1_steps.txt
python3 tools/prof_timeline.py $DB 2 1200 2600 > $O/r03_w1_timeline.txt
rm -rf $O/kt
cat $O/r03_w1_steps.txt; tail -n 3 $O/kt.log
