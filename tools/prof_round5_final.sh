cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5q
# kernel trace + stats (no probe launches in the statistics)
rocprofv3 --kernel-trace --stats -d gpurun_out/r5q/kt64 -o kt -- python3 bench.py --steps 36 --no-probes --no-cpu-baseline > gpurun_out/r5q/kt64.log 2>&1
python3 tools/prof_summary.py $(ls gpurun_out/r5q/kt64/*results.db | head -n 1) 70 > gpurun_out/r5q/r05_kernel_stats_b64.txt
python3 tools/prof_steps.py $(ls gpurun_out/r5q/kt64/*results.db | head -n 1) 6 >> gpurun_out/r5q/r05_kernel_stats_b64.txt
rocprofv3 --kernel-trace --stats -d gpurun_out/r5q/kt16 -o kt -- python3 bench.py --steps 36 --batch 16 --no-probes --no-cpu-baseline > gpurun_out/r5q/kt16.log 2>&1
python3 tools/prof_summary.py $(ls gpurun_out/r5q/kt16/*results.db | head -n 1) 50 > gpurun_out/r5q/r05_kernel_stats_b16.txt
python3 tools/prof_steps.py $(ls gpurun_out/r5q/kt16/*results.db | head -n 1) 6 >> gpurun_out/r5q/r05_kernel_stats_b16.txt
# PMC passes (separate runs, counters only)
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/r5q/pf -o pf --output-format csv -- python3 bench.py --steps 12 --no-probes --no-cpu-baseline > gpurun_out/r5q/pf.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/r5q/pw -o pw --output-format csv -- python3 bench.py --steps 12 --no-probes --no-cpu-baseline > gpurun_out/r5q/pw.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT -d gpurun_out/r5q/pm -o pm --output-format csv -- python3 bench.py --steps 12 --no-probes --no-cpu-baseline > gpurun_out/r5q/pm.log 2>&1
python3 tools/pmc_summary.py $(ls gpurun_out/r5q/pf/*counter_collection.csv | head -n 1) 40 > gpurun_out/r5q/r05_pmc_fetch_b64.txt
python3 tools/pmc_summary.py $(ls gpurun_out/r5q/pw/*counter_collection.csv | head -n 1) 40 > gpurun_out/r5q/r05_pmc_write_b64.txt
python3 tools/pmc_summary.py $(ls gpurun_out/r5q/pm/*counter_collection.csv | head -n 1) 40 > gpurun_out/r5q/r05_pmc_mfma_b64.txt
cd tools && python3 hbm_rates.py ../gpurun_out/r5q/r05_pmc_fetch_b64.txt ../gpurun_out/r5q/r05_pmc_write_b64.txt ../gpurun_out/r5q/r05_kernel_stats_b64.txt --json ../gpurun_out/r5q/r05_traffic_b64.json > ../gpurun_out/r5q/r05_hbm_rates_b64.txt; cd ..
rm -rf gpurun_out/r5q/kt64 gpurun_out/r5q/kt16 gpurun_out/r5q/pf gpurun_out/r5q/pw gpurun_out/r5q/pm
ls -la gpurun_out/r5q; head -n 30 gpurun_out/r5q/r05_hbm_rates_b64.txt; tail -n 4 gpurun_out/r5q/pf.log
python3 bench.py > gpurun_out/r5q/bench_full.log 2>&1; tail -n 1 gpurun_out/r5q/bench_full.log > gpurun_out/r5q/r05_bench_b64.json; tail -c 1500 gpurun_out/r5q/r05_bench_b64.json
python3 tools/gemm_sweep.py --batch 64 --blas > gpurun_out/r5q/r05_gemm_sweep_b64.txt 2>&1
