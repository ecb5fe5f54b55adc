mkdir -p gpurun_out/r6e
python -m pytest tests -m gpu -q --durations=80 > gpurun_out/r6e/suite.txt 2>&1; tail -3 gpurun_out/r6e/suite.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r6e/bench.json 2> gpurun_out/r6e/bench.err; tail -c 600 gpurun_out/r6e/bench.json
