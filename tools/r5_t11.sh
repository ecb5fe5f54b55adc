cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5o
python3 -m pytest tests/test_gpu_model.py -q -m gpu -k "two_ranks and wrapped" 2>&1 | tail -n 4
python3 -m pytest tests/test_gpu_model.py -q -m gpu -s -k "canon_b64" > gpurun_out/r5o/b64.log 2>&1; grep "canon B=64\|outputs\]\|passed\|failed\|Error" gpurun_out/r5o/b64.log
python3 -m pytest tests/test_gpu_model.py -q -m gpu -k "canon_multi or canon_ragged or canon_pretrain or tiny" 2>&1 | tail -n 3
