#!/bin/bash
# A/B of one library under environment settings, on the driver's command shape (20 steps) and at 100 steps, parity gate on:
#   tools/abpoll.sh rounds "VAR=v[,VAR=v]" ...      ("-" = no setting)
cd "$(dirname "$0")/.."
N=$1; shift
for i in $(seq $N); do
  for SPEC in "$@"; do
    E=""; [ "$SPEC" != "-" ] && E=$(echo "$SPEC" | tr ',' ' ')
    for S in 20 100; do
    env $E timeout 120 python bench.py --steps $S --warmup 5 --no-cpu-baseline --no-pmc --parity-receivers 4 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']; print('$SPEC', 'steps $S ms/step', d['ms_per_step'], 'chain', r['whole_chain_frac'], 'k1 ovl', r['avg_launch_ms'], d['parity'][:9])"
    done
  done
done
