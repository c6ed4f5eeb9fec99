#!/bin/bash
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
for r in 1 2 3; do
echo -n "run $r: "; python bench.py --steps 80 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
r=d['roofline']
print(d['ms_per_step'], d['value'], 'k1', r['avg_launch_ms'], 'iso', r['isolated_launch_ms'], 'frac', r['frac'])"; done
echo -n "lane: "; AISGPU_K4=lane python bench.py --steps 80 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
r=d['roofline']
print(d['ms_per_step'], d['value'], 'k1', r['avg_launch_ms'])"
