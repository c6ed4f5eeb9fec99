#!/bin/bash
cd "$(dirname "$0")/.."
for i in 1 2 3; do for L in tools/ab/k7old.so ais-catcher_amd/libaisgpu.so; do
AISGPU_LIB=$(realpath $L) python bench.py --steps 60 --no-cpu-baseline --parity-receivers 4 --gpu-decode 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L gpu-decode', d['value'], d['ms_per_step'], d['parity'][:9])"
done; done
