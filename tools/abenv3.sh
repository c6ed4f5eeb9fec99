#!/bin/bash
# A/B of the built library under environment settings on any bench configuration, 100 steps, parity gate on:
#   tools/abenv3.sh rounds "bench args" "VAR=v[,VAR=v]"|- ...
cd "$(dirname "$0")/.."
N=$1; ARGS=$2; shift 2
for i in $(seq $N); do
  for SPEC in "$@"; do
    E=""; [ "$SPEC" != "-" ] && E=$(echo "$SPEC" | tr ',' ' ')
    env $E timeout 200 python bench.py $ARGS --steps 100 --warmup 5 --no-cpu-baseline --no-pmc --parity-receivers 4 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']; print('$ARGS', '$SPEC', 'ms/step', d['ms_per_step'], 'chain', r['whole_chain_frac'], 'front ovl', r['avg_launch_ms'], d['parity'][:9])"
  done
done
