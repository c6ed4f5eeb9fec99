#!/bin/bash
# is the calling thread the bottleneck with the frame decoders on the device?  host cost per step, both modes
cd "$(dirname "$0")/.."
run() { env $1 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --parity-receivers 4 $2 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']; print('%-30s ms/step %.4f  k1 ovl %.4f  host enqueue %.4f  host cost (device idle) %.4f  %s' % ('$1 $2', d['ms_per_step'], r['avg_launch_ms'], d['host_enqueue_ms_per_step'], d['host_cost_ms_per_step'], d.get('parity')[:9]))"; }
for i in 1 2; do
run A=1 ""
run A=1 --gpu-decode
done
