#!/bin/bash
# HIP maps its streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); the pipeline uses seven streams.  More queues?
cd "$(dirname "$0")/.."
run() { env $1 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --parity-receivers 4 $2 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']; print('%-44s ms/step %.4f  k1 ovl %.4f parity %s' % ('$1 $2', d['ms_per_step'], r['avg_launch_ms'], d.get('parity')[:9]))"; }
for i in 1 2; do
for q in 4 8 16; do
run GPU_MAX_HW_QUEUES=$q ""
run GPU_MAX_HW_QUEUES=$q --gpu-decode
done; done
