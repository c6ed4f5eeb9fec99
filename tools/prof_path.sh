#!/bin/bash
# rocprofv3 kernel stats of one configuration of the library: tools/prof_path.sh OUTNAME "<python kwargs of gpu.AisGpu>" [steps] [R]
# e.g. tools/prof_path.sh base_dec "model=gpu.MODEL_BASE, gpu_decode=True"     (RATE=6000000 in the environment: another sample rate;
# BLOCK=196608: another block length; DISTINCT=1: the bench's batch of distinct receivers instead of 256 copies of one stream; LIB=path: another build of libaisgpu.so)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
R=$PWD; OUT=$1; KW=$2; STEPS=${3:-12}; NRX=${4:-256}; RATE=${RATE:-1536000}; DISTINCT=${DISTINCT:-0}; BLOCK=${BLOCK:-786432}
rm -rf /tmp/prof_$OUT
cat > /tmp/prof_$OUT.py <<PY
import sys
sys.path.insert(0, "$R")
import numpy as np, torch, _pkg
_pkg.load()
from ais_catcher_amd import gpu, synth, workload
B = $BLOCK
if $DISTINCT:
    data = workload.resident_batch(torch, $NRX, 2, sample_rate=$RATE)
else:
    x = synth.receiver_stream(B * 2, sample_rate=$RATE, receiver_id=7)
    dev = torch.from_numpy(np.ascontiguousarray(x.view(np.float32).reshape(2, B, 2))).cuda()
    data = dev.unsqueeze(1).expand(2, $NRX, B, 2).contiguous()
g = gpu.AisGpu(sample_rate=$RATE, n_receivers=$NRX, block_len=B, $KW)
for i in range(8 + $STEPS):
    g.submit_device(data[i & 1].data_ptr(), B); g.run()
g.sync(); g.close()
PY
# PMC="SQ_INSTS_VALU SQ_WAVE_CYCLES": a counter pass instead of the timing pass (kernels serialised: AISGPU_SERIAL=1)
if [ -n "$PMC" ]; then
  (cd /tmp && AISGPU_SERIAL=1 rocprofv3 --kernel-trace --pmc $PMC -d /tmp/prof_$OUT -o res -- python /tmp/prof_$OUT.py > /tmp/prof_$OUT.log 2>&1)
else
  (cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_$OUT -o res -- python /tmp/prof_$OUT.py > /tmp/prof_$OUT.log 2>&1)
fi
mkdir -p gpurun_out
python tools/rocprof_summary.py $(find /tmp/prof_$OUT -name "*.db" | head -1) > gpurun_out/prof_$OUT.txt
[ -n "$TIMELINE" ] && python tools/timeline.py $(find /tmp/prof_$OUT -name "*.db" | head -1) $TIMELINE > gpurun_out/prof_${OUT}_timeline.txt   # TIMELINE="skip count"
grep -E "aisk|kernel " gpurun_out/prof_$OUT.txt | cut -c1-200 | head -${LINES_OUT:-20}
