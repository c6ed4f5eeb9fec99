#!/bin/bash
# A/B of builds on BASELINE configs[2] (6 MSPS ModelChallenger): tools/abc3.sh rounds lib...
cd "$(dirname "$0")/.."
N=$1; shift
for i in $(seq $N); do
  for L in "$@"; do
    AISGPU_LIB=$(realpath $L) python bench.py --config 3 --steps 40 --warmup 5 --no-cpu-baseline --no-pmc --parity-receivers 4 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']; print('$L', 'ms/step', d['ms_per_step'], 'frac', r['whole_chain_frac'], 'pre ovl', r['avg_launch_ms'], 'host', d['host_cost_ms_per_step'], d['parity'][:9])"
  done
done
