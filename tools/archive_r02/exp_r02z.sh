#!/bin/bash
# speculative warm-up length of the chunk-parallel PhaseSearchEMA: step time and how often the exact fallback runs
cd "$(dirname "$0")/.."
for w in 256 192 160 128 96; do
AISGPU_PS_WARM=$w python - <<PY
import os, sys, time
sys.path.insert(0, '.')
import torch, _pkg
_pkg.load()
from ais_catcher_amd import gpu, workload
R, BLOCK, NB = 256, 786432, 6
data = workload.resident_batch(torch, R, NB, seed=3, unique=8, block=BLOCK)
g = gpu.AisGpu(sample_rate=1536000, n_receivers=R, block_len=BLOCK)
for i in range(12):
    g.submit_device(data[i % NB].data_ptr(), BLOCK); g.run()
g.sync()
t0 = time.perf_counter()
N = 60
for i in range(N):
    g.submit_device(data[i % NB].data_ptr(), BLOCK); g.run()
g.sync()
dt = (time.perf_counter() - t0) / N * 1e3
print('warm %3s symbols: %.4f ms per step, %d fallback workgroups in %d blocks x 640 workgroups' % (os.environ['AISGPU_PS_WARM'], dt, g.ps_fallbacks(), N + 12))
PY
done
