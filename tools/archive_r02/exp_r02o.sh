#!/bin/bash
# is the front end held back by the three-deep ring of channel buffers (it waits for the FIR kernel of block f-3)?
cd "$(dirname "$0")/.."
L=ais-catcher_amd/libaisgpu.so
run() { env AISGPU_LIB=$(realpath $1) $2 python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']; print('%-40s ms/step %.4f  k1 ovl %.4f parity %s' % ('$3', d['ms_per_step'], r['avg_launch_ms'], d.get('parity_checked')))"; }
for i in 1 2 3; do
run $L A=1 "ring of 3"
run tools/ab/nbuf4.so A=1 "ring of 4"
run tools/ab/nbuf5.so A=1 "ring of 5"
run tools/ab/nbuf4.so AISGPU_GL=80 "ring of 4, GL 80"
done
