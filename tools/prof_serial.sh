#!/bin/bash
# per-kernel times with every kernel alone (serial mode), current build + env: tools/prof_serial.sh [ENV=..] ...
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
R=$PWD
rm -rf /tmp/prof_serial
env "$@" AISGPU_SERIAL=1 rocprofv3 --kernel-trace --stats -d /tmp/prof_serial -o res -- python $R/bench.py --no-cpu-baseline --steps 8 --parity-receivers 0 > /tmp/prof_serial.log 2>&1
python tools/rocprof_summary.py $(find /tmp/prof_serial -name "*.db" | head -1) | grep -E "aisk|kernel " | head -14
