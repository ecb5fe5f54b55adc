# A/B of two environment settings on ONE box: usage: bash tools/ab_env.sh "VAR=1 VAR2=x" rounds [bench args...]; "default" runs with no extra variable
SET="$1"; R=${2:-2}; shift; shift
for i in $(seq $R); do
  for V in default "$SET"; do
    if [ "$V" = "default" ]; then E=""; else E="$V"; fi
    env $E python3 bench.py --no-probes --no-cpu-baseline --steps 48 --regions 3 "$@" 2>/dev/null | tail -n 1 | python3 -c "import sys, json; d = json.loads(sys.stdin.read()); print('$V', d['regions_min_ms'], d['regions_ms_per_step'])"
  done
done
