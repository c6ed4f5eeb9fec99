cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "one_workgroup or fused_backend_1536k" > gpurun_out/r05_t1_tests.log 2>&1; tail -15 gpurun_out/r05_t1_tests.log
