#!/bin/bash
export TMPDIR=/tmp
for e in "A=1" "AISGPU_GL=24" "AISGPU_GL=16" "AISGPU_GL=80" "AISGPU_GL=8" "A=1"; do
echo -n "$e: "; env $e python bench.py --steps 80 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
r=d['roofline']
print(d['ms_per_step'], d['value'], 'k1', r['avg_launch_ms'], 'iso', r['isolated_launch_ms'])"; done
