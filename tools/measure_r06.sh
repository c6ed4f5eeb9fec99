#!/bin/bash
# Round-6 measurement set (run on the GPU box; outputs under gpurun_out/, copied to profiles/ by hand): usage tools/measure_r06.sh TAG COMMIT
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$PWD; TAG=${1:-r06_v1}
echo "${2:-unknown}" > gpurun_out/.commit
python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_gpu_tests.log 2>&1; tail -2 gpurun_out/${TAG}_gpu_tests.log
python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_steps20.json 2> gpurun_out/${TAG}_bench.err      # the driver's command: own PMC passes, other_configs (configs[1], configs[2], CU8), cpu_baseline
python bench.py --no-cpu-baseline --no-other-configs > gpurun_out/${TAG}_bench.json 2>/dev/null                    # the default 100 steps
python bench.py --config 2 --no-cpu-baseline --no-pmc > gpurun_out/${TAG}_bench_config2_steps100.json 2>/dev/null
python bench.py --config 3 --no-cpu-baseline --no-pmc > gpurun_out/${TAG}_bench_config3_steps100.json 2>/dev/null
python bench.py --config 5 --no-cpu-baseline --no-pmc > gpurun_out/${TAG}_bench_config5_cu8_steps100.json 2>/dev/null
python bench.py --gpu-decode --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_bench_gpu_decode.json 2>/dev/null
(cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_bench -o res -- python $R/bench.py --no-cpu-baseline --no-pmc --no-other-configs > $R/gpurun_out/prof_bench.log 2>&1)
DB=$(find gpurun_out/prof_bench -name "*.db" | head -1)
python tools/rocprof_summary.py $DB > gpurun_out/${TAG}_bench_kernel_stats.txt
python tools/timeline2.py $DB 12 3 > gpurun_out/${TAG}_overlap_timeline.txt
(cd /tmp && AISGPU_SERIAL=1 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_serial -o res -- python $R/bench.py --no-cpu-baseline --no-pmc --no-other-configs --steps 8 --parity-receivers 0 > $R/gpurun_out/prof_serial.log 2>&1)
python tools/rocprof_summary.py $(find gpurun_out/prof_serial -name "*.db" | head -1) > gpurun_out/${TAG}_serial_kernel_stats.txt
LINES_=20 tools/prof_bench.sh ${TAG}_config3 "--config 3 --steps 40 --parity-receivers 0" > /dev/null 2>&1
LINES_=20 tools/prof_bench.sh ${TAG}_config3_serial "--config 3 --steps 8 --parity-receivers 0" AISGPU_SERIAL=1 > /dev/null 2>&1
LINES_=20 tools/prof_bench.sh ${TAG}_config5_cu8 "--config 5 --steps 40 --parity-receivers 0" > /dev/null 2>&1
LINES_=20 tools/prof_bench.sh ${TAG}_config5_cu8_serial "--config 5 --steps 8 --parity-receivers 0" AISGPU_SERIAL=1 > /dev/null 2>&1
tools/tl20.sh ${TAG} --no-other-configs                                                             # kernel timeline + stats of the 20-step shape
GRAFT_REPO_ROOT=$R ./tools/pmc_traffic_all.sh --parity-receivers 0 --no-pmc --no-other-configs > gpurun_out/${TAG}_pmc_traffic_all_kernels.txt 2>&1
python tools/bench_paths.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_paths.txt
BENCH_PATHS_DISTINCT=1 BENCH_PATHS_ONLY="on the device" python tools/bench_paths.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_paths_distinct_receivers.txt
DISTINCT=1 LINES_OUT=12 tools/prof_path.sh ${TAG}_v2_engine "model=gpu.MODEL_V2, gpu_decode=True" 8 > /dev/null 2>&1; mv gpurun_out/prof_${TAG}_v2_engine.txt gpurun_out/${TAG}_v2_engine_kernel_stats.txt   # ModelEngineV2's engine on the device: its kernels alone
for r in 64 384 512 1024; do BENCH_PATHS_DISTINCT=1 BENCH_PATHS_ONLY="whole engine" python tools/bench_paths.py $r 2>&1 | grep "MS/s"; done > gpurun_out/${TAG}_v2_engine_batch_sizes.txt
python __graft_entry__.py smoke 2>&1 | grep -v amdgpu.ids | tail -2 > gpurun_out/${TAG}_smoke.txt
rm -rf gpurun_out/prof_bench gpurun_out/prof_serial gpurun_out/prof_${TAG}_*.log
tail -1 gpurun_out/${TAG}_bench_steps20.json | cut -c1-600; tail -1 gpurun_out/${TAG}_bench.json | cut -c1-300
