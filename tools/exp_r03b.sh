#!/bin/bash
# round 3, experiment B: compile-time tuning after the strip (ring depth 4): PhaseSearch chunk length, derotation/FIR segment length,
# K6 occupancy hint, front-end span length.  A/B on one box, 3 rounds interleaved, the driver's command shape (20 steps) and 100 steps.
cd "$(dirname "$0")/.."
run() { # label, lib, extra args
  env AISGPU_LIB=$2 python bench.py --steps $STEPS --warmup 5 --no-cpu-baseline --parity-receivers 4 $3 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']; print('$1 | steps $STEPS ms/step', d['ms_per_step'], 'k1 ovl', r['avg_launch_ms'], 'k1 iso', r['isolated_launch_ms'], 'parity', d.get('parity_checked'), d.get('parity', '')[:9])"
}
BASE=$(realpath ais-catcher_amd/libaisgpu.so)
for STEPS in 20 100; do
for i in 1 2 3; do
  run base $BASE ""
  for v in ps1536 ps2048 gl80 gl24 k6w2; do run $v $(realpath tools/ab/$v.so) ""; done
  for t in 16 48 64; do run tps$t $BASE "--tiles-per-span $t"; done
done
done
