#!/bin/bash
# quick A/B without the parity gate (ablation builds produce wrong results): abq.sh rounds lib...
cd /root/repo
N=$1; shift
for i in $(seq $N); do
  for L in "$@"; do
    AISGPU_LIB=$(realpath $L) python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-pmc --parity-receivers 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']; print('$L', 'ms/step', d['ms_per_step'], 'k1 ovl', r['avg_launch_ms'], 'k1 iso', r['isolated_launch_ms'])"
  done
done
