#!/bin/bash
# A/B environment settings with the current build: tools/envab.sh "VAR=a" "VAR=b" ... (each arg is an env assignment list)
cd "$(dirname "$0")/.."
for i in 1 2; do
  for E in "$@"; do
    env $E python bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']; print('$E', 'ms/step', d['ms_per_step'], 'k1 ovl', r['avg_launch_ms'], 'k1 iso', r['isolated_launch_ms'], 'host enq', d.get('host_enqueue_ms_per_step'))"
  done
done
