#!/bin/bash
cd "$(dirname "$0")/.."
L=ais-catcher_amd/libaisgpu.so
bash tools/abenv.sh 2 "$L" "$L,AISGPU_TPS=32" "$L,AISGPU_TPS=48" "tools/ab/chunk2048.so" "tools/ab/nocap.so" "$L,AISGPU_STREAM_PRIO=1:99:99" "tools/ab/chunk2048.so,AISGPU_STREAM_PRIO=1:99:99"
