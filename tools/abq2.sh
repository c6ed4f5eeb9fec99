#!/bin/bash
# A/B on the driver's command shape (20 steps) and at 100 steps, parity gate on: tools/abq2.sh rounds lib...
cd "$(dirname "$0")/.."
N=$1; shift
for i in $(seq $N); do
  for L in "$@"; do
    for S in 20 100; do
    AISGPU_LIB=$(realpath ${L%%,*}) python bench.py --steps $S --warmup 5 --no-cpu-baseline --no-pmc --parity-receivers 4 $(echo "$L" | grep -q , && echo "${L#*,}" | tr ',' ' ') 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']; print('$L', 'steps $S ms/step', d['ms_per_step'], 'chain', r['whole_chain_frac'], 'k1 ovl', r['avg_launch_ms'], d['parity'][:9])"
    done
  done
done
