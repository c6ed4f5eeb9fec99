#!/bin/bash
# A/B of bench_paths lines under environment settings: tools/abpaths.sh rounds "substring of the path's name" "VAR=v[,VAR=v]"|- ...
cd "$(dirname "$0")/.."
N=$1; ONLY=$2; shift 2
for i in $(seq $N); do
  for SPEC in "$@"; do
    E=""; [ "$SPEC" != "-" ] && E=$(echo "$SPEC" | tr ',' ' ')
    env $E BENCH_PATHS_ONLY="$ONLY" timeout 300 python tools/bench_paths.py 2>/dev/null | grep "MS/s" | sed "s#^#[$SPEC] #" | cut -c1-190
  done
done
