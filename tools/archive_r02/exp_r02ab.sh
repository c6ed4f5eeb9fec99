#!/bin/bash
# CUs reserved for the phasor recurrence (8 one-wave workgroups for 256 receivers): 8 (one per XCD), 4, 2 (eight SIMDs)
cd "$(dirname "$0")/.."
run() { env $1 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --parity-receivers 4 $2 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']; print('%-30s ms/step %.4f  k1 ovl %.4f %s' % ('$1 $2', d['ms_per_step'], r['avg_launch_ms'], d['parity'][:9]))"; }
for i in 1 2 3; do
run AISGPU_CUMASK=8 ""
run AISGPU_CUMASK=4 ""
run AISGPU_CUMASK=2 ""
done
run AISGPU_CUMASK=8 --gpu-decode
run AISGPU_CUMASK=2 --gpu-decode
