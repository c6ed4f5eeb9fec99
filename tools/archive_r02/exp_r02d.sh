#!/bin/bash
cd "$(dirname "$0")/.."
L=ais-catcher_amd/libaisgpu.so
bash tools/abenv.sh 3 "$L,AISGPU_K4_STREAMS=1" "$L" "$L,AISGPU_PS_PRIO=1"
