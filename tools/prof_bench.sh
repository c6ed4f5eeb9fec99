#!/bin/bash
# rocprofv3 kernel stats of a bench.py run:  tools/prof_bench.sh TAG "bench args" [ENV=v ...]   -> gpurun_out/TAG_kernel_stats.txt
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
R=$PWD; TAG=$1; ARGS=$2; shift 2
mkdir -p gpurun_out
rm -rf gpurun_out/prof_$TAG
(cd /tmp && env "$@" rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o res -- python $R/bench.py $ARGS --no-cpu-baseline --no-pmc --no-other-configs > $R/gpurun_out/prof_$TAG.log 2>&1)
DB=$(find gpurun_out/prof_$TAG -name "*.db" | head -1)
python tools/rocprof_summary.py $DB > gpurun_out/${TAG}_kernel_stats.txt
rm -rf gpurun_out/prof_$TAG
cat gpurun_out/${TAG}_kernel_stats.txt | cut -c1-200 | head -${LINES_:-30}
