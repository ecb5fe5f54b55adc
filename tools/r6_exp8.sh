cd $GRAFT_REPO_ROOT
O=gpurun_out/r6j; mkdir -p $O
python -m pytest tests/test_gpu_ops.py tests/test_vit.py -q -x > $O/ops.txt 2>&1; tail -3 $O/ops.txt
python -m pytest tests/test_gpu_model.py -q -x -k "fp32 or finetune or rollout or vlnbert or navcmt or tiny" > $O/model.txt 2>&1; tail -3 $O/model.txt
python tools/e2e_bench.py 1 12 graph 2>/dev/null | tail -3 | cut -c1-400
python tools/rollout_bench.py --reps 4 2>/dev/null | grep "ms_per_step\|eager"
python bench.py --no-probes --no-cpu-baseline --steps 48 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('bench', d['regions_ms_per_step'])"
