#!/bin/bash
# usage: tools/prof_k.sh TAG [env assignments...]: rocprofv3 kernel stats of the bench, overlapped and serial, into gpurun_out/TAG_*.txt
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
R=$PWD
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
rm -rf gpurun_out/prof_$TAG gpurun_out/profs_$TAG
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o res -- python $R/bench.py --no-cpu-baseline --steps 60 > gpurun_out/${TAG}_bench.log 2>&1
python tools/rocprof_summary.py $(find gpurun_out/prof_$TAG -name "*.db" | head -1) | grep -v "at::native\|rocclr" > gpurun_out/${TAG}_stats.txt
python tools/timeline2.py $(find gpurun_out/prof_$TAG -name "*.db" | head -1) 30 2 > gpurun_out/${TAG}_timeline.txt
AISGPU_SERIAL=1 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/profs_$TAG -o res -- python $R/bench.py --no-cpu-baseline --steps 8 > gpurun_out/${TAG}_serial.log 2>&1
python tools/rocprof_summary.py $(find gpurun_out/profs_$TAG -name "*.db" | head -1) | grep -v "at::native\|rocclr" > gpurun_out/${TAG}_serial_stats.txt
rm -rf gpurun_out/prof_$TAG gpurun_out/profs_$TAG
tail -1 gpurun_out/${TAG}_bench.log | cut -c1-200
cut -c1-150 gpurun_out/${TAG}_stats.txt
echo; cut -c1-150 gpurun_out/${TAG}_serial_stats.txt
echo; cat gpurun_out/${TAG}_timeline.txt
