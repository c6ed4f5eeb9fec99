#!/usr/bin/env python3
"""Secondary rates for DESIGN.md section 9 (never the bench `value`): blocks handed over as HOST buffers.
  (1) PCIe-inclusive chain rate through the C ABI alone: aisgpu_submit (pageable -> pinned -> H2D) x R, aisgpu_run, no decoding
  (2) the real multi-receiver flow: R receiver threads -> GpuBatch -> host decoders (or device decoders) -> NMEA, classic
      (launch, wait, decode) and pipelined (decode block f-1 and copy block f+1 in while the device runs f)
  (3) end-to-end single receiver incl. host decode
usage: tools/measure_host_path.py [R]
"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import _pkg  # noqa: E402

_pkg.load()
from ais_catcher_amd import gpu, host, synth  # noqa: E402

BLOCK = 786432


def pcie_inclusive(R=64, steps=6, threads=16):
    x = synth.receiver_stream(BLOCK, receiver_id=1)
    g = gpu.AisGpu(n_receivers=R, block_len=BLOCK)

    def fill():
        def part(t):
            for r in range(t, R, threads):
                g.submit(r, x)
        th = [threading.Thread(target=part, args=(t,)) for t in range(threads)]
        [t.start() for t in th]
        [t.join() for t in th]
    fill()
    g.run()
    g.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        fill()
        g.run()
    g.sync()
    dt = time.perf_counter() - t0
    g.close()
    return R * BLOCK * steps / dt / 1e6


def batch_flow(R, nblocks, pipelined, gpu_decode):
    xs = [synth.receiver_stream(BLOCK * 2, receiver_id=10 + u) for u in range(4)]
    batch = host.Batch(n_receivers=R, block_len=BLOCK, gpu_decode=gpu_decode)
    batch.set_timeout(0)
    if pipelined:
        batch.set_pipelined(True)
    models = [host.ModelDefaultGPU(block_len=BLOCK, batch=batch, rx=r) for r in range(R)]
    t_start = [0.0]
    bar = threading.Barrier(R + 1)

    def run(r):
        x = xs[r % 4]
        models[r].receive(x[:BLOCK])      # warm-up block (allocations, first launch)
        bar.wait()
        for b in range(nblocks):
            models[r].receive(x[(b & 1) * BLOCK:((b & 1) + 1) * BLOCK])
        if pipelined:
            models[r].flush()

    th = [threading.Thread(target=run, args=(r,)) for r in range(R)]
    [t.start() for t in th]
    bar.wait()
    t0 = time.perf_counter()
    [t.join() for t in th]
    dt = time.perf_counter() - t0
    n = sum(len(m.nmea()) for m in models)
    for m in models:
        m.close()
    batch.close()
    return R * BLOCK * nblocks / dt / 1e6, n


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
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    print("PCIe-inclusive, C ABI only (%d receivers, host buffers via aisgpu_submit from 16 threads, no decoding): %.0f MS/s" % (R, pcie_inclusive(R)))
    for pipelined in (False, True):
        for dec in (False, True):
            r, n = batch_flow(R, 6, pipelined, dec)
            print("%d receiver threads -> GpuBatch -> %s decoders -> NMEA, %s: %.0f MS/s (%d NMEA lines)"
                  % (R, "device" if dec else "host", "pipelined" if pipelined else "classic", r, n))
    r, n = end_to_end_single()
    print("single receiver end-to-end incl. D2H + host decode (ModelDefaultGPU): %.1f MS/s, %d NMEA lines" % (r, n))
