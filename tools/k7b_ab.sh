#!/bin/bash
# A/B of K7b builds (ModelBase, sampler + decoder on the device): tools/k7b_ab.sh name=lib.so ...   (base = the in-tree library)
# per build: the bench line with one stream replicated and with 256 distinct receivers
cd "$(dirname "$0")/.."
export BENCH_PATHS_ONLY="SimplePLL"
for v in base "$@"; do
  name=${v%%=*}; lib=${v#*=}
  if [ "$v" = base ]; then unset AISGPU_LIB; else export AISGPU_LIB=$PWD/$lib; fi
  echo "== $name"
  python tools/bench_paths.py 2>&1 | grep -i "SimplePLL" | cut -c52-140
  BENCH_PATHS_DISTINCT=1 python tools/bench_paths.py 2>&1 | grep -i "SimplePLL" | cut -c52-160
done
