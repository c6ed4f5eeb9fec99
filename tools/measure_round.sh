#!/bin/bash
# The measurement set that is committed under profiles/ (run on the GPU box): HBM traffic of the front-end kernel (PMC passes),
# then the bench line WITH that same-session traffic figure, per-kernel rocprofv3 stats of the same command, serial-mode
# per-kernel stats, SQ counters.  usage: tools/measure_round.sh TAG [COMMIT]
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$PWD
TAG=${1:-r02_v1}
echo "${2:-unknown}" > gpurun_out/.commit
GRAFT_REPO_ROOT=$R ./tools/pmc_traffic.sh > gpurun_out/pmc_traffic.log 2>&1
cp gpurun_out/pmc_traffic.json gpurun_out/${TAG}_pmc_traffic_k1.json 2>/dev/null
BENCH_TRAFFIC_JSON=$R/gpurun_out/${TAG}_pmc_traffic_k1.json python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_bench -o res -- python $R/bench.py --no-cpu-baseline > gpurun_out/prof_bench.log 2>&1
python tools/rocprof_summary.py $(find gpurun_out/prof_bench -name "*.db" | head -1) > gpurun_out/${TAG}_bench_kernel_stats.txt
python tools/timeline2.py $(find gpurun_out/prof_bench -name "*.db" | head -1) 12 3 > gpurun_out/${TAG}_overlap_timeline.txt
AISGPU_SERIAL=1 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_serial -o res -- python $R/bench.py --no-cpu-baseline --steps 8 --parity-receivers 0 > gpurun_out/prof_serial.log 2>&1
python tools/rocprof_summary.py $(find gpurun_out/prof_serial -name "*.db" | head -1) > gpurun_out/${TAG}_serial_kernel_stats.txt
GRAFT_REPO_ROOT=$R ./tools/pmc_k1.sh > gpurun_out/${TAG}_pmc_sq.txt 2>&1
GRAFT_REPO_ROOT=$R ./tools/pmc_traffic_all.sh --parity-receivers 0 > gpurun_out/${TAG}_pmc_traffic_all_kernels.txt 2>&1
rm -rf gpurun_out/prof_bench gpurun_out/prof_serial gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/pmc_sq1 gpurun_out/pmc_sq2
tail -1 gpurun_out/${TAG}_bench.json | cut -c1-600
