#!/bin/bash
# PhaseSearch step: scalar EMA update (3 issue slots instead of 5), pinned v_bfe_i32 / v_add3_u32 (24 -> 20-21 slots per step)
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "phase_search or benchmarked or fallback or extreme or fused_fir or edge" 2>&1 | tail -2
run() { env AISGPU_LIB=$(realpath $1) python bench.py --steps 60 --warmup 5 --no-cpu-baseline --parity-receivers 4 $2 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']; print('%-44s ms/step %.4f  k1 ovl %.4f %s' % ('$1 $2', d['ms_per_step'], r['avg_launch_ms'], d['parity'][:9]))"; }
for i in 1 2 3; do
run tools/ab/head.so ""
run ais-catcher_amd/libaisgpu.so ""
done
run tools/ab/head.so --gpu-decode
run ais-catcher_amd/libaisgpu.so --gpu-decode
