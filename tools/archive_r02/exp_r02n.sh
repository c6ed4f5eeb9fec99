#!/bin/bash
# what does PhaseSearch take from the front end: its arithmetic or its memory traffic?  (variants with wrong results; parity off)
cd "$(dirname "$0")/.."
L=ais-catcher_amd/libaisgpu.so
run() { env AISGPU_LIB=$(realpath $1) $2 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --parity-receivers 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']; print('%-40s ms/step %.4f  k1 ovl %.4f' % ('$3', d['ms_per_step'], r['avg_launch_ms']))"; }
for i in 1 2; do
run $L A=1 "everything"
run tools/ab/k4nocomp.so A=1 "PhaseSearch without its arithmetic"
run tools/ab/k4noload.so A=1 "PhaseSearch without its loads"
run $L AISGPU_ABLATE=1 "no PhaseSearch at all"
done
