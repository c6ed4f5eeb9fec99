#!/bin/bash
# A/B of library options (AISGPU_<KEY>=v in the environment, forwarded by ais-catcher_amd/gpu.py) on any bench configuration, interleaved:
#   tools/abopt.sh rounds "bench args" "VAR=v[,VAR=v]"|- ...          e.g.  tools/abopt.sh 3 "--config 3" - AISGPU_US_K1=0
cd "$(dirname "$0")/.."
N=$1; ARGS=$2; shift 2
for i in $(seq $N); do
  for SPEC in "$@"; do
    E=""; [ "$SPEC" != "-" ] && E=$(echo "$SPEC" | tr ',' ' ')
    for S in ${STEPS:-20 100}; do
    env $E timeout 300 python bench.py $ARGS --steps $S --warmup 5 --no-cpu-baseline --no-pmc --no-other-configs --parity-receivers ${PARITY:-4} 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']; print('$SPEC', 'steps $S ms/step', d['ms_per_step'], 'value', d['value'], 'chain', r['whole_chain_frac'], 'kernel', r['avg_launch_ms'], d['parity'][:9])"
    done
  done
done
