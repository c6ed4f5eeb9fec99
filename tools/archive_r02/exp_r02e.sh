#!/bin/bash
cd "$(dirname "$0")/.."
L=ais-catcher_amd/libaisgpu.so
bash tools/abenv.sh 3 "$L,AISGPU_K4_STREAMS=1" "tools/ab/k4v24b4.so,AISGPU_K4_STREAMS=1" "tools/ab/k4v24.so,AISGPU_K4_STREAMS=1" "tools/ab/k4v16b4.so,AISGPU_K4_STREAMS=1" "tools/ab/k4v20b4.so,AISGPU_K4_STREAMS=1"
