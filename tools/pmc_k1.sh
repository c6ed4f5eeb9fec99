#!/bin/bash
# PMC passes for the front-end kernel (serial mode so kernels do not overlap). Output: gpurun_out/pmc_*/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { # name, counters...
  name=$1; shift
  AISGPU_SERIAL=1 timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $R/gpurun_out/pmc_$name -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --parity-receivers 0 > $R/gpurun_out/pmc_$name.log 2>&1
  python $R/tools/rocprof_summary.py $R/gpurun_out/pmc_$name/p_results.db 2>&1 | grep -E "aisk" | head -80
}
run sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVES
