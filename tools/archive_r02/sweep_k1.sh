#!/bin/bash
# K1 geometry sweep on the GPU box ("threads,tile96,depth"): serial (isolated) K1 time + overlapped step time
cd "$(dirname "$0")/.."
for cfg in "64,64,1" "64,64,0"; do
  AISGPU_K1=$cfg python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']; print('K1=$cfg : ms_per_step', d['ms_per_step'], 'value', d['value'], 'k1 overlapped ms', r['avg_launch_ms'], 'isolated ms', r['isolated_launch_ms'], 'isolated frac', r['isolated_frac'])"
done
