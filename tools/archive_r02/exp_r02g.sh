#!/bin/bash
cd "$(dirname "$0")/.."
L=ais-catcher_amd/libaisgpu.so
bash tools/abenv.sh 3 "$L" "tools/ab/nt1.so" "tools/ab/aux3.so" "tools/ab/aux16.so" "tools/ab/aux17.so" "tools/ab/aux18.so" "tools/ab/aux19.so"
