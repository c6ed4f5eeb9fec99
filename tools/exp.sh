#!/bin/bash
export TMPDIR=/tmp
for r in 1 2 3; do
for A in 1 0; do
echo -n "ahead=$A: "; AISGPU_ROT_AHEAD=$A python bench.py --steps 80 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
r=d['roofline']
print(d['ms_per_step'], d['value'], 'k1', r['avg_launch_ms'], 'iso', r['isolated_launch_ms'])"; done; done
