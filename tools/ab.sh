# A/B of two builds of the library on ONE box (boxes differ by +-2 %): ab/libA.so and ab/libB.so alternate, N rounds of bench.py
# usage: bash tools/ab.sh [rounds] [bench args...]
R=${1:-3}; shift
for i in $(seq $R); do
  for V in A B; do
    cp ab/lib$V.so vln_hamt_amd/libhamt_hip.so
    python3 bench.py --no-probes --steps 48 --regions 3 "$@" 2>/dev/null | tail -n 1 | python3 -c "import sys, json; d = json.loads(sys.stdin.read()); print('$V', d['regions_min_ms'], d['regions_ms_per_step'])"
  done
done
