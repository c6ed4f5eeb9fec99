#!/bin/bash
# round 3, experiment A: (1) does hipExtAnyOrderLaunch overlap launches on gfx950, (2) instruction-class issue costs,
# (3) the CPU baseline's host (cgroup limit, thread scaling), (4) A/B of the any-order front-end launch in the pipeline.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
echo "== host: $(hostname)  $(date -u +%FT%TZ)"
echo "== microbench_anyorder"; timeout 120 tools/bin/mb_anyorder
echo "== microbench_issue"; timeout 300 tools/bin/mb_issue
} > gpurun_out/r03_expA_micro.txt 2>&1
timeout 400 python tools/cpu_probe.py 16 > gpurun_out/r03_expA_cpu.txt 2>&1
{
echo "== A/B any-order (20-step lines, the driver's command shape: --steps 20 --warmup 5)"
for i in 1 2 3; do
  for SPEC in "base" "AISGPU_ANYORDER=1" "AISGPU_ANYORDER=1 AISGPU_LIB=$(realpath tools/ab/nbuf4.so)" "AISGPU_LIB=$(realpath tools/ab/nbuf4.so)"; do
    E="$SPEC"; [ "$SPEC" = base ] && E="X=1"
    env $E python bench.py --steps 20 --warmup 5 --no-cpu-baseline --parity-receivers 8 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']; print('$SPEC'.replace('$(realpath tools/ab)/',''), '| ms/step', d['ms_per_step'], 'k1 ovl', r['avg_launch_ms'], 'k1 iso', r['isolated_launch_ms'], 'parity', d.get('parity_checked'), d.get('parity', '')[:9])"
  done
done
echo "== 100-step lines"
for i in 1 2; do
  for SPEC in "base" "AISGPU_ANYORDER=1" "AISGPU_ANYORDER=1 AISGPU_LIB=$(realpath tools/ab/nbuf4.so)"; do
    E="$SPEC"; [ "$SPEC" = base ] && E="X=1"
    env $E python bench.py --steps 100 --warmup 5 --no-cpu-baseline --parity-receivers 4 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']; print('$SPEC'.replace('$(realpath tools/ab)/',''), '| ms/step', d['ms_per_step'], 'k1 ovl', r['avg_launch_ms'], 'k1 iso', r['isolated_launch_ms'], 'parity', d.get('parity_checked'), d.get('parity', '')[:9])"
  done
done
} > gpurun_out/r03_expA_ab.txt 2>&1
tail -n 40 gpurun_out/r03_expA_micro.txt gpurun_out/r03_expA_ab.txt
