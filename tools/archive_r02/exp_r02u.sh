#!/bin/bash
# which of the three decoder kernels costs the step its 0.08 ms?  (skipped kernels: results wrong, parity off)
cd "$(dirname "$0")/.."
run() { env $1 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --parity-receivers 0 $2 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']; print('%-40s ms/step %.4f  k1 ovl %.4f' % ('$1 $2', d['ms_per_step'], r['avg_launch_ms']))"; }
for i in 1 2; do
run A=1 ""
run A=1 --gpu-decode
run AISGPU_K7E_SKIP=7 --gpu-decode
run AISGPU_K7E_SKIP=6 --gpu-decode
run AISGPU_K7E_SKIP=4 --gpu-decode
run AISGPU_K7E_SKIP=2 --gpu-decode
done
