#!/usr/bin/env python3
"""Events / runs per decoder of the event-driven frame decoders on the bench signal (AISGPU_K7E_STATS=1 prints them at every
sync_outputs): python tools/k7e_stats.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["AISGPU_K7E_STATS"] = "1"
import torch  # noqa: E402
import bench  # noqa: E402
import _pkg  # noqa: E402
_pkg.load()
from ais_catcher_amd import gpu  # noqa: E402

R = 256
data = bench.make_resident_input(torch, R, 2, seed=0)
g = gpu.AisGpu(sample_rate=bench.RATE, n_receivers=R, block_len=bench.BLOCK, gpu_decode=True)
for i in range(4):
    g.submit_device(data[i % 2].data_ptr(), bench.BLOCK)
    g.run()
    g.sync_outputs()
    print("block", i, "frames", len(g.frames()))
g.close()
