#!/bin/bash
export TMPDIR=/tmp
for e in "A=1" "AISGPU_K1_PER_SIMD=3" "AISGPU_K1_PER_SIMD=2" "AISGPU_K1_PER_SIMD=3 AISGPU_K4=lane" "AISGPU_K1_PER_SIMD=2 AISGPU_K4=lane" "AISGPU_K1_PER_SIMD=1"; do echo "$e"; env $e python bench.py --no-cpu-baseline | python -c "
import sys,json
d=json.loads(sys.stdin.read())
r=d['roofline']
print(d['ms_per_step'], d['value'], 'k1', r['avg_launch_ms'], 'iso', r['isolated_launch_ms'], 'frac', r['frac'])"; done
