#!/bin/bash
# per-kernel rocprofv3 stats of two libaisgpu builds on the same box: tools/prof_ab.sh libA.so libB.so
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
R=$PWD
for L in "$@"; do
  tag=$(basename $L .so)
  rm -rf gpurun_out/prof_$tag
  AISGPU_LIB=$(realpath $L) rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$tag -o res -- python $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline > gpurun_out/prof_$tag.log 2>&1
  db=$(find gpurun_out/prof_$tag -name "*.db" | head -1)
  python tools/rocprof_summary.py $db > gpurun_out/stats_$tag.txt
  tail -1 gpurun_out/prof_$tag.log | cut -c1-200
done
