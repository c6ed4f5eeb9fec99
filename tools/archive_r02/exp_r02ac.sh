#!/bin/bash
# front-end DMA loads with immediate offsets (one address pair and one M0 per four pieces instead of one 64-bit add and one M0 per piece)
cd "$(dirname "$0")/.."
AISGPU_LIB=$(realpath tools/ab/k1imm.so) timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -o timeout=60 -k "golden or benchmarked or spectral or cu8 or ladder or rates or 256 or tiling or edge" 2>&1 | tail -2
run() { env AISGPU_LIB=$(realpath $1) python bench.py --steps 60 --warmup 5 --no-cpu-baseline --parity-receivers 4 $2 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']; print('%-36s ms/step %.4f  k1 ovl %.4f iso %.4f %s' % ('$1 $2', d['ms_per_step'], r['avg_launch_ms'], r['isolated_launch_ms'], d['parity'][:9]))"; }
for i in 1 2 3; do
run ais-catcher_amd/libaisgpu.so ""
run tools/ab/k1imm.so ""
done
