cd $GRAFT_REPO_ROOT
for v in "HAMT_LN_BWD_MAXB=256" "A=1"; do echo "== $v"; env $v BF16=1 python tools/ln_bench.py 2>/dev/null | grep "p=0.1"; done
for v in "A=1" "HAMT_LN_BWD_MAXB=256" "A=2" "HAMT_LN_BWD_MAXB=256"; do
  env $v python bench.py --no-probes --no-cpu-baseline --steps 48 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$v', d['regions_ms_per_step'])"
done
for v in "A=1" "HAMT_LN_BWD_MAXB=256"; do
  env $v python bench.py --no-probes --no-cpu-baseline --steps 48 --batch 16 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('B16 $v', d['regions_ms_per_step'])"
done
python -m pytest tests/test_gpu_ops.py tests/test_vit.py -q -x -k "ln or layernorm or norm or vit or block" 2>&1 | tail -3
