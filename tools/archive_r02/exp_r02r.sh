#!/bin/bash
# per-block timeline with the frame decoders on the device
cd /tmp && export TMPDIR=/tmp
AISGPU_DEC_STREAM=4 rocprofv3 --kernel-trace --stats -d /tmp/p8 -o k7 -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --no-cpu-baseline --parity-receivers 0 --gpu-decode > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/timeline2.py $(find /tmp/p8 -name '*.db' | head -1) 14 4
