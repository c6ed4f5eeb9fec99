#!/bin/bash
# the frame decoders on a stream of their own (fifth stream) or behind the derotation / FIR kernel (s4) or PhaseSearch (s1)
cd "$(dirname "$0")/.."
run() { env $1 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --parity-receivers 4 $2 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']; print('%-40s ms/step %.4f  k1 ovl %.4f %s' % ('$1 $2', d['ms_per_step'], r['avg_launch_ms'], d['parity'][:9]))"; }
for i in 1 2; do
run A=1 ""
run A=1 --gpu-decode
run AISGPU_DEC_STREAM=4 --gpu-decode
run AISGPU_DEC_STREAM=1 --gpu-decode
done
