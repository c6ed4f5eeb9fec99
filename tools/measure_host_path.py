#!/usr/bin/env python3
"""Secondary rates for DESIGN.md section 9 (never the bench `value`):
  (1) PCIe-inclusive chain rate: blocks handed over as HOST buffers through aisgpu_submit (pageable -> pinned -> H2D)
  (2) end-to-end single receiver incl. host decode: ModelDefaultGPU.receive (C++ host: GPU chain + 10 decoders + NMEA)
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import _pkg  # noqa: E402

_pkg.load()
from ais_catcher_amd import gpu, host, synth  # noqa: E402

BLOCK = 786432


def pcie_inclusive(R=64, steps=6):
    x = synth.receiver_stream(BLOCK, receiver_id=1)
    g = gpu.AisGpu(n_receivers=R, block_len=BLOCK)
    for r in range(R):
        g.submit(r, x)
    g.run()
    g.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        for r in range(R):
            g.submit(r, x)
        g.run()
    g.sync()
    dt = time.perf_counter() - t0
    g.close()
    return R * BLOCK * steps / dt / 1e6


def end_to_end_single(nblocks=8):
    x = synth.receiver_stream(BLOCK * nblocks, receiver_id=2)
    m = host.ModelDefaultGPU(block_len=BLOCK)
    m.receive(x[:BLOCK])
    t0 = time.perf_counter()
    for b in range(1, nblocks):
        m.receive(x[b * BLOCK:(b + 1) * BLOCK])
    dt = time.perf_counter() - t0
    n = len(m.nmea())
    m.close()
    return BLOCK * (nblocks - 1) / dt / 1e6, n


if __name__ == "__main__":
    print("PCIe-inclusive (64 receivers, host buffers via aisgpu_submit): %.0f MS/s" % pcie_inclusive())
    r, n = end_to_end_single()
    print("single receiver end-to-end incl. D2H + host decode (ModelDefaultGPU): %.1f MS/s, %d NMEA lines" % (r, n))
