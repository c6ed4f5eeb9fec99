#!/bin/bash
# round-2 experiment A: where does the step time come from?  ablations (results invalid, parity off) and PhaseSearch variants
cd "$(dirname "$0")/.."
run() { # label, env...
  L=$1; shift
  env "$@" python bench.py --steps 40 --warmup 5 --no-cpu-baseline --parity-receivers 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']; print('%-44s ms/step %.4f  k1 ovl %.4f  k1 iso %.4f' % ('$L', d['ms_per_step'], r['avg_launch_ms'], r['isolated_launch_ms']))"
}
for i in 1 2; do
run "baseline" A=1
run "no PhaseSearch (ablate 1)" AISGPU_ABLATE=1
run "no derot/FIR (ablate 2)" AISGPU_ABLATE=2
run "no PS, no derot/FIR (ablate 3)" AISGPU_ABLATE=3
run "front end only (ablate 7)" AISGPU_ABLATE=7
run "no front end (ablate 8)" AISGPU_ABLATE=8
run "PhaseSearch only (ablate 14)" AISGPU_ABLATE=14
run "lane variant" AISGPU_K4=lane
run "lane variant prio 1" AISGPU_K4=lane AISGPU_PS_PRIO=1
run "lane variant prio 3" AISGPU_K4=lane AISGPU_PS_PRIO=3
run "lane variant cl 256" AISGPU_K4=lane AISGPU_PS_CL=256
run "lane variant cl 256 prio 2" AISGPU_K4=lane AISGPU_PS_CL=256 AISGPU_PS_PRIO=2
done
