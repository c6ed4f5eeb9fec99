#!/bin/bash
# The measurement set that is committed under profiles/ (run on the GPU box): bench line, per-kernel rocprofv3 stats of the
# same command, serial-mode per-kernel stats, SQ counters and HBM traffic of the front-end kernel.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$PWD
TAG=${1:-r01_v5}
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_bench -o res -- python $R/bench.py --no-cpu-baseline > gpurun_out/prof_bench.log 2>&1
python tools/rocprof_summary.py $(find gpurun_out/prof_bench -name "*.db" | head -1) > gpurun_out/${TAG}_bench_kernel_stats.txt
python tools/timeline2.py $(find gpurun_out/prof_bench -name "*.db" | head -1) 12 3 > gpurun_out/${TAG}_overlap_timeline.txt
AISGPU_SERIAL=1 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_serial -o res -- python $R/bench.py --no-cpu-baseline --steps 8 > gpurun_out/prof_serial.log 2>&1
python tools/rocprof_summary.py $(find gpurun_out/prof_serial -name "*.db" | head -1) > gpurun_out/${TAG}_serial_kernel_stats.txt
GRAFT_REPO_ROOT=$R ./tools/pmc_traffic.sh > gpurun_out/pmc_traffic.log 2>&1
GRAFT_REPO_ROOT=$R ./tools/pmc_k1.sh > gpurun_out/${TAG}_pmc_sq.txt 2>&1
cp gpurun_out/pmc_traffic.json gpurun_out/${TAG}_pmc_traffic_k1.json 2>/dev/null
rm -rf gpurun_out/prof_bench gpurun_out/prof_serial gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/pmc_sq1 gpurun_out/pmc_sq2
tail -1 gpurun_out/${TAG}_bench.json | cut -c1-400
