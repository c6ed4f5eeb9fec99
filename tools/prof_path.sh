#!/bin/bash
# rocprofv3 kernel stats of one configuration of the library: tools/prof_path.sh OUTNAME "<python kwargs of gpu.AisGpu>" [steps] [R]
# e.g. tools/prof_path.sh base_dec "model=gpu.MODEL_BASE, gpu_decode=True"     (RATE=6000000 in the environment: another sample rate)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
R=$PWD; OUT=$1; KW=$2; STEPS=${3:-12}; NRX=${4:-256}; RATE=${RATE:-1536000}
rm -rf /tmp/prof_$OUT
cat > /tmp/prof_$OUT.py <<PY
import sys
sys.path.insert(0, "$R")
import numpy as np, torch, _pkg
_pkg.load()
from ais_catcher_amd import gpu, synth
B = 786432
x = synth.receiver_stream(B * 2, sample_rate=$RATE, receiver_id=7)
dev = torch.from_numpy(np.ascontiguousarray(x.view(np.float32).reshape(2, B, 2))).cuda()
data = dev.unsqueeze(1).expand(2, $NRX, B, 2).contiguous()
g = gpu.AisGpu(sample_rate=$RATE, n_receivers=$NRX, block_len=B, $KW)
for i in range(8 + $STEPS):
    g.submit_device(data[i & 1].data_ptr(), B); g.run()
g.sync(); g.close()
PY
(cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_$OUT -o res -- python /tmp/prof_$OUT.py > /tmp/prof_$OUT.log 2>&1)
mkdir -p gpurun_out
python tools/rocprof_summary.py $(find /tmp/prof_$OUT -name "*.db" | head -1) > gpurun_out/prof_$OUT.txt
grep -E "aisk|kernel " gpurun_out/prof_$OUT.txt | cut -c1-200 | head -20
