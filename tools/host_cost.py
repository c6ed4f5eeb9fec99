#!/usr/bin/env python3
"""Host-side cost of one aisgpu_run() call (no back-pressure: the device is idle and the host at most 2 blocks ahead)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import _pkg
_pkg.load()
from ais_catcher_amd import gpu
R, BLOCK = 256, 786432
x = torch.zeros((R, BLOCK, 2), dtype=torch.float32, device="cuda")
torch.cuda.synchronize()
g = gpu.AisGpu(sample_rate=1536000, n_receivers=R, block_len=BLOCK)
for _ in range(4):
    g.submit_device(x.data_ptr(), BLOCK); g.run()
g.sync()
res = []
for rep in range(5):
    ts = []
    for k in range(2):
        t0 = time.perf_counter()
        g.submit_device(x.data_ptr(), BLOCK); g.run()
        ts.append((time.perf_counter() - t0) * 1e3)
    g.sync()
    res.append(ts)
print("host ms per run() call [1st, 2nd] x5:", [[round(v, 3) for v in t] for t in res])
