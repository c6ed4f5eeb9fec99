#!/bin/bash
# frame decoders of block f behind the derotation / FIR kernel of block f+1 on ITS stream (four active queues instead of five)
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "decoder or decoders or benchmarked or deferred" 2>&1 | tail -2
AISGPU_DEC_STREAM=4 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "decoder or decoders or benchmarked or deferred" 2>&1 | tail -2
run() { env $1 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --parity-receivers 4 $2 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']; print('%-50s ms/step %.4f  k1 ovl %.4f %s' % ('$1 $2', d['ms_per_step'], r['avg_launch_ms'], d['parity'][:9]))"; }
for i in 1 2; do
run A=1 ""
run A=1 --gpu-decode
run AISGPU_DEC_STREAM=4 --gpu-decode
run "AISGPU_DEC_STREAM=4 AISGPU_DEC_DEFER=0" --gpu-decode
done
