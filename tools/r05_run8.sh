cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { python bench.py --steps 20 --warmup 5 --preroll $1 --no-cpu-baseline --no-pmc --parity-receivers 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']; print('preroll $1 idle $2', 'ms/step', d['ms_per_step'], 'chain', r['whole_chain_frac'], 'k1 ovl', r['avg_launch_ms'], 'k1 iso', r['isolated_launch_ms'])"; }
{
for i in 1 2; do
  sleep 15; run 40 15
  sleep 15; run 400 15
  sleep 15; run 2000 15
  sleep 15; run 8000 15
done
run 40 0; run 40 0; run 2000 0
rocm-smi --showclocks 2>/dev/null | head -20
} > gpurun_out/r05_t7_preroll.txt 2>&1
cat gpurun_out/r05_t7_preroll.txt
