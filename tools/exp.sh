#!/bin/bash
export TMPDIR=/tmp
for r in 1 2; do
for L in base cap3 nw4 cap3nw4 cap3nw2; do
for D in 0 1; do
echo -n "$L defer=$D: "; AISGPU_DEFER_FUSED=$D AISGPU_LIB=$(realpath tools/ab/lib_$L.so) python bench.py --steps 80 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
r=d['roofline']
print(d['ms_per_step'], d['value'], 'k1', r['avg_launch_ms'], 'iso', r['isolated_launch_ms'])"; done; done; done
