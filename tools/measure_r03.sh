#!/bin/bash
# Round-3 measurement set (run on the GPU box; outputs under gpurun_out/, copied to profiles/ by hand): usage tools/measure_r03.sh TAG COMMIT
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$PWD; TAG=${1:-r03_v2}
echo "${2:-unknown}" > gpurun_out/.commit
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err                        # default: 100 steps, own PMC passes, cpu_baseline
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_bench_steps20.json 2>/dev/null   # the driver's command shape
python bench.py --config 2 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > gpurun_out/${TAG}_bench_config2.json 2>/dev/null
python bench.py --config 3 --steps 20 --warmup 5 --cpu-seconds 10 > gpurun_out/${TAG}_bench_config3.json 2>/dev/null
python bench.py --gpu-decode --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_bench_gpu_decode.json 2>/dev/null
(cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_bench -o res -- python $R/bench.py --no-cpu-baseline --no-pmc > $R/gpurun_out/prof_bench.log 2>&1)
python tools/rocprof_summary.py $(find gpurun_out/prof_bench -name "*.db" | head -1) > gpurun_out/${TAG}_bench_kernel_stats.txt
python tools/timeline2.py $(find gpurun_out/prof_bench -name "*.db" | head -1) 12 3 > gpurun_out/${TAG}_overlap_timeline.txt
(cd /tmp && AISGPU_SERIAL=1 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_serial -o res -- python $R/bench.py --no-cpu-baseline --no-pmc --steps 8 --parity-receivers 0 > $R/gpurun_out/prof_serial.log 2>&1)
python tools/rocprof_summary.py $(find gpurun_out/prof_serial -name "*.db" | head -1) > gpurun_out/${TAG}_serial_kernel_stats.txt
GRAFT_REPO_ROOT=$R ./tools/pmc_k1.sh > gpurun_out/${TAG}_pmc_sq.txt 2>&1
GRAFT_REPO_ROOT=$R ./tools/pmc_traffic_all.sh --parity-receivers 0 --no-pmc > gpurun_out/${TAG}_pmc_traffic_all_kernels.txt 2>&1
python tools/bench_paths.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_paths.txt
BENCH_PATHS_DISTINCT=1 BENCH_PATHS_ONLY="on the device" python tools/bench_paths.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_paths_distinct_receivers.txt   # the device decoders on 256 DISTINCT receivers (divergence)
tools/bin/mb_issue > gpurun_out/${TAG}_mb_issue.txt 2>&1
python __graft_entry__.py smoke 2>&1 | grep -v amdgpu.ids | tail -2 > gpurun_out/${TAG}_smoke.txt
rm -rf gpurun_out/prof_bench gpurun_out/prof_serial gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/pmc_sq1 gpurun_out/pmc_sq2
tail -1 gpurun_out/${TAG}_bench.json | cut -c1-900; tail -1 gpurun_out/${TAG}_bench_steps20.json | cut -c1-300
