#!/bin/bash
cd "$(dirname "$0")/.."
bash tools/abenv.sh 3 "tools/ab/k7old.so" "ais-catcher_amd/libaisgpu.so" "ais-catcher_amd/libaisgpu.so,AISGPU_GL=80" "ais-catcher_amd/libaisgpu.so,AISGPU_GL=24"
