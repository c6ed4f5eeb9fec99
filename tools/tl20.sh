#!/bin/bash
# kernel timeline of the driver's command shape (20 steps) under rocprofv3: tools/tl20.sh TAG [bench args]  ->  gpurun_out/TAG_tl20.txt (+ kernel stats)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
R=$PWD; TAG=$1; shift
(cd /tmp && rocprofv3 --kernel-trace -d $R/gpurun_out/prof_tl -o res -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --parity-receivers 0 "$@" > $R/gpurun_out/prof_tl.log 2>&1)
DB=$(find gpurun_out/prof_tl -name "*.db" | head -1)
python tools/timeline_dump.py $DB > gpurun_out/${TAG}_tl20.txt
python tools/rocprof_summary.py $DB > gpurun_out/${TAG}_tl20_stats.txt
rm -rf gpurun_out/prof_tl
