#!/bin/bash
# device frame decoders: k7e_sim with the word-parallel frame evaluator against the per-symbol one (tools/ab/k7old2.so)
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "decoder or decoders or benchmarked or deferred" 2>&1 | tail -3
for i in 1 2; do for L in tools/ab/k7ring2.so ais-catcher_amd/libaisgpu.so; do
AISGPU_LIB=$(realpath $L) python bench.py --steps 60 --no-cpu-baseline --parity-receivers 4 --gpu-decode 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L gpu-decode', d['value'], d['ms_per_step'], d['parity'][:9])"
done; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/p7 -o k7 -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --no-cpu-baseline --parity-receivers 0 --gpu-decode > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(find /tmp/p7 -name '*.db' | head -1) | grep -E 'k7|k1_dpp|k4_|k3_|Name' | head -20
