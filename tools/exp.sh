#!/bin/bash
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
for r in 1 2; do
for D in 0 1; do
echo -n "defer=$D: "; AISGPU_DEFER_FUSED=$D python bench.py --steps 80 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
r=d['roofline']
print(d['ms_per_step'], d['value'], 'k1', r['avg_launch_ms'], 'iso', r['isolated_launch_ms'])"; done; done
AISGPU_SERIAL=1 rocprofv3 --kernel-trace --stats -d /tmp/pp -o res -- python bench.py --no-cpu-baseline --steps 8 > /dev/null 2>&1
python tools/rocprof_summary.py $(find /tmp/pp -name "*.db" | head -1) | grep -v "at::native\|rocclr" | head -8 | cut -c1-150
