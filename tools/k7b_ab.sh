#!/bin/bash
# A/B of K7b builds on distinct receivers: tools/k7b_ab.sh name=lib.so ...   (base = the in-tree library)
cd "$(dirname "$0")/.."
export BENCH_PATHS_ONLY="SimplePLL" BENCH_PATHS_DISTINCT=1
for v in base "$@"; do
  name=${v%%=*}; lib=${v#*=}
  if [ "$v" = base ]; then unset AISGPU_LIB; else export AISGPU_LIB=$PWD/$lib; fi
  echo "== $name"
  python tools/bench_paths.py 2>&1 | grep -i "SimplePLL"
  DISTINCT=1 tools/prof_path.sh base_dec_$name "model=gpu.MODEL_BASE, gpu_decode=True" | grep -E "k7b|k1_dpp|k5_f"
done
