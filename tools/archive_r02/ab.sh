#!/bin/bash
# A/B two builds of libaisgpu.so on the same GPU box, interleaved: tools/ab.sh libA.so libB.so [rounds]
cd "$(dirname "$0")/.."
A=$1; B=$2; N=${3:-3}
for i in $(seq $N); do
  for L in "$A" "$B"; do
    AISGPU_LIB=$(realpath $L) python bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']; print('$L', 'ms/step', d['ms_per_step'], 'k1 ovl', r['avg_launch_ms'], 'k1 iso', r['isolated_launch_ms'], 'host enq', d.get('host_enqueue_ms_per_step'))"
  done
done
