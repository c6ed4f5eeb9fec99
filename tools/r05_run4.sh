cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { # lib env steps
  env $2 AISGPU_LIB=$1 timeout 120 python bench.py --steps $3 --warmup 5 --no-cpu-baseline --no-pmc --parity-receivers 4 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']; print('$1 $2', 'steps $3 ms/step', d['ms_per_step'], 'chain', r['whole_chain_frac'], 'k1 ovl', r['avg_launch_ms'], d['parity'][:9])"
}
D=$PWD/ais-catcher_amd/libaisgpu.so
for i in 1 2; do
for L in $D $PWD/tools/ab/c1536.so $PWD/tools/ab/c2048.so; do
  for E in AISGPU_K46=1 AISGPU_K46=0; do run $L $E 20; run $L $E 100; done
done; done > gpurun_out/r05_t3_chunks.txt 2>&1
cat gpurun_out/r05_t3_chunks.txt
