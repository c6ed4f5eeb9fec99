cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "one_workgroup or fused_backend or fallback or resampled_ladders or 256" > gpurun_out/r05_t8_tests.log 2>&1; tail -6 gpurun_out/r05_t8_tests.log
tools/abpoll.sh 3 - AISGPU_PS_SPLIT=0 > gpurun_out/r05_t8_ab.txt 2>&1; cat gpurun_out/r05_t8_ab.txt
