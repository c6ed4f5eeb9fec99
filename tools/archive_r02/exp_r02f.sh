#!/bin/bash
cd "$(dirname "$0")/.."
L=ais-catcher_amd/libaisgpu.so
bash tools/abenv.sh 4 "$L" "tools/ab/nt1.so" "tools/ab/nt2.so" "tools/ab/nt3.so" "tools/ab/nt4.so"
