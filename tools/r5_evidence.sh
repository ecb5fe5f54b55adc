# round-5 evidence files of the FINAL tree: parity margins of the gated tests (pytest -s lines) and run-to-run reproducibility
mkdir -p gpurun_out/r5x
python -m pytest tests/test_gpu_model.py -q -s -k "canon_b64 or dead_code or canon_multi or canon_ragged" > gpurun_out/r5x/par.log 2>&1
( echo "# round 5 (final tree, dead-code elimination on): lines printed by the gated parity tests -- test_canon_b64_vs_oracle, test_unread_outputs_of_the_last_cross_layer_are_dead_code,"
  echo "# test_canon_multi_seed_margins, test_canon_ragged_vs_reference_goldens (pytest -s)"
  grep -E "^\.?\[|^    \[|passed|failed" gpurun_out/r5x/par.log ) > gpurun_out/r5x/r05_parity_margins.txt
( echo "# tools/grad_bitwise_repeat.py <task> 30: forward + backward of one batch 30 times (dropout off, two streams): gradient tensors that are not bit-identical over the repetitions"
  for t in mlm sap sar sprel mrc; do echo "== $t"; python tools/grad_bitwise_repeat.py $t 30 2>/dev/null | tail -n 4; done ) > gpurun_out/r5x/r05_determinism.txt
tail -n 12 gpurun_out/r5x/r05_determinism.txt; tail -n 3 gpurun_out/r5x/r05_parity_margins.txt
