# round 6, experiment batch 1 (run on the GPU box from the repo root)
mkdir -p gpurun_out/r6c
python -m pytest tests/test_gpu_model.py -x -q -s -k "test_overlapped_grad_sync_matches_single_process_steps" > gpurun_out/r6c/ovsync.txt 2>&1
tail -3 gpurun_out/r6c/ovsync.txt
B="python bench.py --no-probes --no-cpu-baseline --steps 48"
run() { name=$1; shift; env "$@" $B > gpurun_out/r6c/$name.json 2> gpurun_out/r6c/$name.err; python - <<PY
import json
try:
    d=json.loads([l for l in open("gpurun_out/r6c/$name.json") if l.startswith("{")][-1]); print("$name", d["regions_ms_per_step"])
except Exception as e: print("$name failed", e)
PY
}
run base64 A=1
run lag64_full HAMT_OVERLAP_UPDATE=1
run lag64_256 HAMT_OVERLAP_UPDATE=1 HAMT_ADAMW_MAX_BLOCKS=256
run lag64_128 HAMT_OVERLAP_UPDATE=1 HAMT_ADAMW_MAX_BLOCKS=128
run lag64_64 HAMT_OVERLAP_UPDATE=1 HAMT_ADAMW_MAX_BLOCKS=64
run unit24 HAMT_WGRAD_UNIT_TILES=24
B="python bench.py --no-probes --no-cpu-baseline --steps 48 --batch 16"
run base16 A=1
run lag16_128 HAMT_OVERLAP_UPDATE=1 HAMT_ADAMW_MAX_BLOCKS=128
run lag16_256 HAMT_OVERLAP_UPDATE=1 HAMT_ADAMW_MAX_BLOCKS=256
