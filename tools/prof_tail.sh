#!/bin/bash
# timeline of the tail of the B = 64 step (end of backward -> weight gradients -> norm -> update) for the last steps of a run
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r4t; mkdir -p $out; cd $R
timeout 600 rocprofv3 --kernel-trace --stats -d $out/kt -o kt -- python3 bench.py --steps 24 --no-probes --no-cpu-baseline > $out/kt.log 2>&1
DB=$(ls $out/kt/*results.db | head -n 1)
cd tools
for k in 2 3 4; do python3 prof_timeline.py $DB $k 1100 1400 > $out/tail_$k.txt; done
cd ..; rm -rf $out/kt; tail -60 $out/tail_2.txt
