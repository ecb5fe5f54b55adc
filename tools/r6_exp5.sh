cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6g; mkdir -p $O
python -m pytest tests -m gpu -q --durations=25 > $O/suite.txt 2>&1; tail -3 $O/suite.txt
# one-rank exchange timeline
HAMT_FORCE_DIST=1 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python3 bench.py --steps 24 --no-probes --no-cpu-baseline > $O/kt.log 2>&1
DB=$(ls $O/kt/*results.db | head -n 1)
python3 tools/prof_steps.py $DB 6 > $O/w1_steps.txt
python3 tools/prof_timeline.py $DB 2 1200 2600 > $O/w1_timeline.txt
rm -rf $O/kt
# wgrad unit size in the real step: FETCH_SIZE of the grouped launch, units of 12 / 24 / 32 tiles
for U in 12 24 32; do
  HAMT_WGRAD_UNIT_TILES=$U rocprofv3 --pmc FETCH_SIZE -d $O/pf$U -o pf --output-format csv -- python3 bench.py --steps 12 --no-probes --no-cpu-baseline > $O/pf$U.log 2>&1
  echo "== unit tiles $U" >> $O/unit_fetch.txt
  python3 tools/pmc_summary.py $(ls $O/pf$U/*counter_collection.csv | head -n 1) 3 >> $O/unit_fetch.txt
  rm -rf $O/pf$U
  HAMT_WGRAD_UNIT_TILES=$U python bench.py --no-probes --no-cpu-baseline --steps 48 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('unit $U', d['regions_ms_per_step'])" >> $O/unit_fetch.txt
done
cat $O/unit_fetch.txt | cut -c1-200
python tools/aten_sources.py > $O/aten.txt 2>&1; tail -50 $O/aten.txt | cut -c1-200
