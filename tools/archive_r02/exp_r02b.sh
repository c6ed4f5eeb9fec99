#!/bin/bash
cd "$(dirname "$0")/.."
L=ais-catcher_amd/libaisgpu.so
bash tools/abenv.sh 2 "$L,AISGPU_ELIDE_WAITS=0" "$L" "$L,BENCH_NO_K1_EVENTS=1" "tools/ab/nt.so" "$L,AISGPU_ABLATE=7" "$L,AISGPU_ABLATE=7,BENCH_NO_K1_EVENTS=1" "$L,AISGPU_ABLATE=7,AISGPU_ELIDE_WAITS=0" "tools/ab/nt.so,AISGPU_ABLATE=7"
