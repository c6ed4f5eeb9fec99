cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
tools/abpoll.sh 3 - AISGPU_K46=0 > gpurun_out/r05_t2_ab.txt 2>&1
R=$PWD
(cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_b -o res -- python $R/bench.py --no-cpu-baseline --no-pmc --steps 40 > /tmp/prof_b.log 2>&1)
python tools/rocprof_summary.py $(find /tmp/prof_b -name "*.db" | head -1) > gpurun_out/r05_t2_k46_kernel_stats.txt
tools/prof_serial.sh > gpurun_out/r05_t2_k46_serial.txt 2>&1
cat gpurun_out/r05_t2_ab.txt; head -20 gpurun_out/r05_t2_k46_kernel_stats.txt; cat gpurun_out/r05_t2_k46_serial.txt
