#!/bin/bash
cd "$(dirname "$0")/.."
for i in 1 2; do for L in "$@"; do AISGPU_LIB=$(realpath $L) python bench.py --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']; print('$L', 'ms/step', d['ms_per_step'], 'k1 ovl', r['avg_launch_ms'], 'k1 iso', r['isolated_launch_ms'])"; done; done
