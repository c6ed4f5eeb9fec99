#!/bin/bash
# K1 tile-size / prefetch-depth sweep on the GPU box: serial (non-overlapped) kernel times from the library's HIP events
cd "$(dirname "$0")/.."
for cfg in "256 1" "256 2" "256 3" "128 2" "128 3" "128 4" "64 4"; do
  set -- $cfg
  AISGPU_SERIAL=1 AISGPU_TILE96=$1 AISGPU_DEPTH=$2 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('P=$1 D=$2 serial : ms_per_step', d['ms_per_step'], 'k1_ms', d['roofline']['avg_launch_ms'], 'frac', d['roofline']['frac'])"
  AISGPU_TILE96=$1 AISGPU_DEPTH=$2 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('P=$1 D=$2 overlap: ms_per_step', d['ms_per_step'], 'k1_ms', d['roofline']['avg_launch_ms'], 'value', d['value'])"
done
