#!/bin/bash
# Round 6: every profile the bench line and DESIGN quote, from ONE tree, into gpurun_out/r6p (copy the r06_* files to profiles/).
#   gpurun --timeout 2400 -- 'bash tools/prof_round6.sh'
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6p; mkdir -p $O
# 1) kernel trace + stats, B = 64 and B = 16 (no probe launches in the statistics); per-step / per-queue timelines from the same trace
rocprofv3 --kernel-trace --stats -d $O/kt64 -o kt -- python3 bench.py --steps 36 --no-probes --no-cpu-baseline > $O/kt64.log 2>&1
DB=$(ls $O/kt64/*results.db | head -n 1)
python3 tools/prof_summary.py $DB 70 > $O/r06_kernel_stats_b64.txt
python3 tools/prof_steps.py $DB 6 >> $O/r06_kernel_stats_b64.txt
( cd tools; python3 prof_alone.py ../$DB 24 60 > ../$O/r06_alone_vs_concurrent.txt; python3 prof_step_queues.py ../$DB 12 --kinds > ../$O/r06_step_queues_b64.txt;
  python3 prof_step_queues.py ../$DB 1 --dump 12 > ../$O/r06_step_dump_sap.txt )
rocprofv3 --kernel-trace --stats -d $O/kt16 -o kt -- python3 bench.py --steps 36 --batch 16 --no-probes --no-cpu-baseline > $O/kt16.log 2>&1
DB16=$(ls $O/kt16/*results.db | head -n 1)
python3 tools/prof_summary.py $DB16 50 > $O/r06_kernel_stats_b16.txt
python3 tools/prof_steps.py $DB16 6 >> $O/r06_kernel_stats_b16.txt
( cd tools; python3 prof_step_queues.py ../$DB16 12 --kinds > ../$O/r06_step_queues_b16.txt )
# 2) PMC passes (separate runs, counters only)
rocprofv3 --pmc FETCH_SIZE -d $O/pf -o pf --output-format csv -- python3 bench.py --steps 12 --no-probes --no-cpu-baseline > $O/pf.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pw -o pw --output-format csv -- python3 bench.py --steps 12 --no-probes --no-cpu-baseline > $O/pw.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT -d $O/pm -o pm --output-format csv -- python3 bench.py --steps 12 --no-probes --no-cpu-baseline > $O/pm.log 2>&1
python3 tools/pmc_summary.py $(ls $O/pf/*counter_collection.csv | head -n 1) 40 > $O/r06_pmc_fetch_b64.txt
python3 tools/pmc_summary.py $(ls $O/pw/*counter_collection.csv | head -n 1) 40 > $O/r06_pmc_write_b64.txt
python3 tools/pmc_summary.py $(ls $O/pm/*counter_collection.csv | head -n 1) 40 > $O/r06_pmc_mfma_b64.txt
( cd tools; python3 hbm_rates.py ../$O/r06_pmc_fetch_b64.txt ../$O/r06_pmc_write_b64.txt ../$O/r06_kernel_stats_b64.txt --json ../$O/r06_traffic_b64.json > ../$O/r06_hbm_rates_b64.txt )
# 3) BASELINE configs 4 / 5: kernel-family tables (program directly behind `--`)
rocprofv3 --kernel-trace --stats -d $O/e2e -o e2e -- python3 tools/e2e_bench.py 1 12 graph > $O/e2e.log 2>&1
python3 tools/prof_summary.py $(ls $O/e2e/*results.db | head -n 1) 40 > $O/r06_kernel_stats_e2e_b1.txt
rocprofv3 --kernel-trace --stats -d $O/roll -o roll -- python3 tools/rollout_bench.py --reps 3 > $O/roll.log 2>&1
python3 tools/prof_summary.py $(ls $O/roll/*results.db | head -n 1) 40 > $O/r06_kernel_stats_rollout.txt
rm -rf $O/kt64 $O/kt16 $O/pf $O/pw $O/pm $O/e2e $O/roll
# 4) the bench line of the same tree, the GEMM sweep, the parity margins
python3 bench.py > $O/bench_full.log 2>&1; tail -n 1 $O/bench_full.log > $O/r06_bench_b64.json
python3 tools/gemm_sweep.py --batch 64 --blas > $O/r06_gemm_sweep_b64.txt 2>&1
python -m pytest tests/test_gpu_model.py -q -s -k "canon_b64 or dead_code or canon_multi or canon_ragged" > $O/par.log 2>&1
( echo "# round 6 (final tree): lines printed by the gated parity tests -- test_canon_b64_vs_oracle, test_unread_outputs_of_the_last_cross_layer_are_dead_code,"
  echo "# test_canon_multi_seed_margins, test_canon_ragged_vs_reference_goldens (pytest -s).  Head outputs gated at 1e-2 flat."
  grep -E "^\.?\[|^    \[|passed|failed" $O/par.log ) > $O/r06_parity_margins.txt
# 5) soak: 960 steps, B = 64, full-length and ragged batches (training sanity of the final build)
( echo "# tools/soak.py 960 steps, B = 64, bf16, hipGraph replay, 12 fixed synthetic batches per task (final build of round 6: IEEE-half dense outputs in front of the LayerNorms)"
  echo "# full-length batches (L = 80, T = 5):"; SOAK_B=64 python3 tools/soak.py 960 2>/dev/null | grep "steps:"
  echo "# ragged batches (SOAK_RAGGED=1: L ~ U[20, 80], T ~ U[0, 7]; text packed through the text-only and the cross-modal layers):"; SOAK_B=64 SOAK_RAGGED=1 python3 tools/soak.py 960 2>/dev/null | grep "steps:" ) > $O/r06_soak_b64_960steps.txt
ls -la $O; head -n 30 $O/r06_hbm_rates_b64.txt; tail -c 1500 $O/r06_bench_b64.json
