#!/bin/bash
# A/B builds x environment settings on one GPU box: tools/abenv.sh rounds "lib.so[,VAR=v,...]" ...   (STEPS=, PARITY=, ABARGS="--config 3" in the environment)
cd "$(dirname "$0")/.."
N=$1; shift
for i in $(seq $N); do
  for SPEC in "$@"; do
    L=${SPEC%%,*}; E=""; [ "$SPEC" != "$L" ] && E=$(echo "${SPEC#*,}" | tr ',' ' ')
    env $E AISGPU_LIB=$(realpath $L) python bench.py --steps ${STEPS:-40} --warmup 5 --no-cpu-baseline --no-pmc --no-other-configs --parity-receivers ${PARITY:-4} $ABARGS 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']; print('$SPEC', 'ms/step', d['ms_per_step'], 'k1 ovl', r['avg_launch_ms'], 'k1 iso', r['isolated_launch_ms'], 'host', d.get('host_cost_ms_per_step'), 'parity', d.get('parity_checked'), d.get('parity', '')[:9])"
  done
done
