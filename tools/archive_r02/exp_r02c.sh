#!/bin/bash
cd "$(dirname "$0")/.."
L=ais-catcher_amd/libaisgpu.so
bash tools/abenv.sh 3 "$L,AISGPU_K46=0" "$L" "tools/ab/nt.so" "tools/ab/nt.so,AISGPU_K46=0"
