#!/bin/bash
cd "$(dirname "$0")/.."
for i in 1 2; do for R in 256 128 64 192; do
python bench.py --steps 60 --warmup 5 --no-cpu-baseline --parity-receivers 0 --receivers $R 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']; print('R=$R', 'MS/s', d['value'], 'ms/step', d['ms_per_step'], 'per-256-equivalent', round(d['ms_per_step']*256/$R,4), 'k1 ovl', r['avg_launch_ms'], 'k1 iso', r['isolated_launch_ms'])"
done; done
