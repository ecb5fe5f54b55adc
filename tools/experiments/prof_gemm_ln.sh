# rocprofv3 evidence for the fused dense + LayerNorm kernel (csrc/gemm_ln.hip) against the two-launch path: kernel trace + one
# counter pass over tools/gemm_ln_bench.py.  Run through gpurun from the repo root: bash tools/prof_gemm_ln.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3ln
mkdir -p $O
python3 tools/gemm_ln_bench.py > $O/r03_gemm_ln_bench.txt 2>&1
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python3 tools/gemm_ln_bench.py 5120x768 11520x768 5120x3072 > $O/kt.log 2>&1
python3 tools/prof_summary.py $(ls $O/kt/*results.db | head -n 1) 12 > $O/r03_gemm_ln_kernel_stats.txt
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES -d $O/pm -o pm --output-format csv -- python3 tools/gemm_ln_bench.py 5120x768 > $O/pm.log 2>&1
python3 tools/pmc_summary.py $(ls $O/pm/*counter_collection.csv | head -n 1) 8 > $O/r03_gemm_ln_pmc.txt
rm -rf $O/kt $O/pm
cat $O/r03_gemm_ln_kernel_stats.txt; cat $O/r03_gemm_ln_pmc.txt
