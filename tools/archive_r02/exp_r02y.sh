#!/bin/bash
# events bound to the front-end dispatch (hipExtLaunchKernelGGL) against event packets around it
cd "$(dirname "$0")/.."
run() { env $1 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --parity-receivers 4 $2 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']; print('%-44s ms/step %.4f  k1 ovl %.4f (%d launches) %s' % ('$1 $2', d['ms_per_step'], r['avg_launch_ms'], r['launches'], d['parity'][:9]))"; }
for i in 1 2 3; do
run AISGPU_EXT_LAUNCH=0 ""
run AISGPU_EXT_LAUNCH=1 ""
run "AISGPU_EXT_LAUNCH=0 BENCH_NO_K1_EVENTS=1" ""
run "AISGPU_EXT_LAUNCH=1 BENCH_NO_K1_EVENTS=1" ""
done
run AISGPU_EXT_LAUNCH=0 --gpu-decode
run AISGPU_EXT_LAUNCH=1 --gpu-decode
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "benchmarked or pipelined or batch" 2>&1 | tail -2
