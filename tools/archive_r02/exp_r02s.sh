#!/bin/bash
# frame decoders on the reserved CUs (next to the phasor recurrence) instead of among the front end's waves
cd "$(dirname "$0")/.."
run() { env $1 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --parity-receivers 4 --gpu-decode 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']; print('%-44s ms/step %.4f  k1 ovl %.4f parity %s' % ('$1', d['ms_per_step'], r['avg_launch_ms'], d.get('parity')[:9]))"; }
for i in 1 2; do
run A=1
run AISGPU_DEC_MASK=1
run "AISGPU_DEC_MASK=1 AISGPU_CUMASK=16"
run "AISGPU_DEC_MASK=1 AISGPU_CUMASK=24"
run "AISGPU_CUMASK=16"
done
