#!/bin/bash
cd "$(dirname "$0")/.."
L=ais-catcher_amd/libaisgpu.so
bash tools/abenv.sh 2 "$L" "$L,AISGPU_STREAM_PRIO=99:-1:99" "$L,AISGPU_STREAM_PRIO=99:-1:-1" "$L,AISGPU_STREAM_PRIO=1:-1:-1" "$L,AISGPU_STREAM_PRIO=1:99:99" "$L,AISGPU_STREAM_PRIO=0:0:0" "$L,AISGPU_STREAM_PRIO=-1:1:1"
