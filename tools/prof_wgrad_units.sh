# sweep of the grouped weight-gradient unit shape (tiles pinned to one XCD together): time and L2-miss fetch per launch
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/wu
python3 -m pytest tests/test_gpu_ops.py -q -m gpu -k "wgrad" -x 2>&1 | tail -n 3 > gpurun_out/wu/tests.txt
MIX="52:768x768x5120,13:3072x768x5120,13:768x3072x5120,13:2304x768x5120"
for CFG in "0 12" "1 12" "1 16" "1 18" "1 24"; do
  set -- $CFG
  export HAMT_WGRAD_UNIT_2D=$1 HAMT_WGRAD_UNIT_TILES=$2
  echo "== 2d $1 unit tiles $2" >> gpurun_out/wu/time.txt
  python3 tools/wgrad_bench.py $MIX 2>&1 | grep TFLOP >> gpurun_out/wu/time.txt
  rocprofv3 --pmc FETCH_SIZE -d gpurun_out/wu/pf -o pf --output-format csv -- python3 tools/wgrad_bench.py $MIX > gpurun_out/wu/pf.log 2>&1
  echo "== 2d $1 unit tiles $2" >> gpurun_out/wu/fetch.txt
  python3 tools/pmc_summary.py $(ls gpurun_out/wu/pf/*counter_collection.csv | head -n 1) 1 >> gpurun_out/wu/fetch.txt
  rm -rf gpurun_out/wu/pf
done
cat gpurun_out/wu/tests.txt gpurun_out/wu/time.txt gpurun_out/wu/fetch.txt
