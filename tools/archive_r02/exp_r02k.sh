#!/bin/bash
cd "$(dirname "$0")/.."
L=ais-catcher_amd/libaisgpu.so
bash tools/abenv.sh 3 "$L" "$L,AISGPU_TPS=32,AISGPU_STREAM_PRIO=1:99:99" "$L,AISGPU_STREAM_PRIO=1:99:99" "$L,AISGPU_TPS=32,AISGPU_STREAM_PRIO=1:99:99,AISGPU_CUMASK=0" "$L,AISGPU_STREAM_PRIO=1:0:0"
