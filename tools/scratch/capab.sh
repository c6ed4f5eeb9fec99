for v in cap1536 cap768 cap256; do echo == $v; AISGPU_LIB=$PWD/tools/ab/$v.so DISTINCT=1 tools/prof_path.sh bd_$v "model=gpu.MODEL_BASE, gpu_decode=True" | grep "k7b"; done
