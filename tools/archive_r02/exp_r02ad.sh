#!/bin/bash
# front-end occupancy cap (3 per SIMD by register allocation) revisited after the PhaseSearch diet: no cap (4) / cap 2
cd "$(dirname "$0")/.."
run() { env AISGPU_LIB=$(realpath $1) python bench.py --steps 60 --warmup 5 --no-cpu-baseline --parity-receivers 4 $2 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']; print('%-36s ms/step %.4f  k1 ovl %.4f iso %.4f %s' % ('$1 $2', d['ms_per_step'], r['avg_launch_ms'], r['isolated_launch_ms'], d['parity'][:9]))"; }
for i in 1 2; do
run ais-catcher_amd/libaisgpu.so ""
run tools/ab/nocap.so ""
run tools/ab/cap2.so ""
done
