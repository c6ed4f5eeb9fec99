#!/bin/bash
# Round-5 measurement set (run on the GPU box; outputs under gpurun_out/, copied to profiles/ by hand): usage tools/measure_r05.sh TAG COMMIT
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$PWD; TAG=${1:-r05_v1}
echo "${2:-unknown}" > gpurun_out/.commit
python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_gpu_tests.log 2>&1; tail -2 gpurun_out/${TAG}_gpu_tests.log
python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_steps20.json 2> gpurun_out/${TAG}_bench.err      # the driver's command: own PMC passes, other_configs, cpu_baseline
python bench.py --no-cpu-baseline --no-other-configs > gpurun_out/${TAG}_bench.json 2>/dev/null                    # the default 100 steps
python bench.py --config 2 --no-cpu-baseline --no-pmc > gpurun_out/${TAG}_bench_config2_steps100.json 2>/dev/null
python bench.py --config 3 --no-cpu-baseline --no-pmc > gpurun_out/${TAG}_bench_config3_steps100.json 2>/dev/null
python bench.py --gpu-decode --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_bench_gpu_decode.json 2>/dev/null
(cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_bench -o res -- python $R/bench.py --no-cpu-baseline --no-pmc > $R/gpurun_out/prof_bench.log 2>&1)
DB=$(find gpurun_out/prof_bench -name "*.db" | head -1)
python tools/rocprof_summary.py $DB > gpurun_out/${TAG}_bench_kernel_stats.txt
python tools/timeline2.py $DB 12 3 > gpurun_out/${TAG}_overlap_timeline.txt
(cd /tmp && AISGPU_SERIAL=1 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_serial -o res -- python $R/bench.py --no-cpu-baseline --no-pmc --steps 8 --parity-receivers 0 > $R/gpurun_out/prof_serial.log 2>&1)
python tools/rocprof_summary.py $(find gpurun_out/prof_serial -name "*.db" | head -1) > gpurun_out/${TAG}_serial_kernel_stats.txt
tools/tl20.sh ${TAG}                                                                                # kernel timeline + stats of the 20-step shape
GRAFT_REPO_ROOT=$R ./tools/pmc_traffic_all.sh --parity-receivers 0 --no-pmc > gpurun_out/${TAG}_pmc_traffic_all_kernels.txt 2>&1
python tools/bench_paths.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_paths.txt
BENCH_PATHS_DISTINCT=1 BENCH_PATHS_ONLY="on the device" python tools/bench_paths.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_paths_distinct_receivers.txt
{ for RX in 1024 2048; do BENCH_PATHS_DISTINCT=1 BENCH_PATHS_ONLY="EngineV2" python tools/bench_paths.py $RX 2>&1 | grep -v amdgpu.ids | grep -v "^parity gate"; done; python tools/v2_end_to_end.py 16 4; } > gpurun_out/${TAG}_v2_engine.txt 2>&1
DISTINCT=1 tools/prof_path.sh ${TAG}_v2dev "model=gpu.MODEL_V2, gpu_decode=True" 6 256 > /dev/null 2>&1; cp gpurun_out/prof_${TAG}_v2dev.txt gpurun_out/${TAG}_v2_engine_kernel_stats.txt
python __graft_entry__.py smoke 2>&1 | grep -v amdgpu.ids | tail -2 > gpurun_out/${TAG}_smoke.txt
rm -rf gpurun_out/prof_bench gpurun_out/prof_serial
tail -1 gpurun_out/${TAG}_bench_steps20.json | cut -c1-600; tail -1 gpurun_out/${TAG}_bench.json | cut -c1-300
