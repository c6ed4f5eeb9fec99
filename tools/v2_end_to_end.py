#!/usr/bin/env python3
"""ModelEngineV2 end to end (input block in, NMEA out) through the host model: the engine on the host (the device hands over the 48 kHz
channels) against the engine on the device (AISGPU_FLAG_GPU_DECODE: frames back).  One receiver = one host thread, like the
reference's device threads; R receivers share a batch.  usage: tools/v2_end_to_end.py [R] [blocks]"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import _pkg  # noqa: E402

_pkg.load()
from ais_catcher_amd import host, synth  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 16
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 6
B = 786432
xs = [synth.receiver_stream(B * 2, receiver_id=300 + r, gap_slots=(0, 2)) for r in range(min(R, 8))]
for dec in (False, True):
    host.reset_sequence()
    batch = host.Batch(n_receivers=R, block_len=B, model=11, gpu_decode=dec) if R > 1 else None
    ms = [host.ModelEngineV2GPU(block_len=B, batch=batch, rx=r, gpu_decode=dec) for r in range(R)]

    def work(r):
        x = xs[r % len(xs)]
        for b in range(NB):
            ms[r].receive(x[(b & 1) * B:((b & 1) + 1) * B])
    th = [threading.Thread(target=work, args=(r,)) for r in range(R)]
    t0 = time.perf_counter()
    [t.start() for t in th]
    [t.join() for t in th]
    dt = time.perf_counter() - t0
    n = sum(len(m.nmea()) for m in ms)
    print("engine on the %s: %d receivers x %d blocks of %d samples in %.3f s = %.1f MS/s end to end (%d NMEA lines; %d host threads, cgroup quota applies)"
          % ("device" if dec else "host  ", R, NB, B, dt, R * NB * B / dt / 1e6, n, R), flush=True)
    for m in ms:
        m.close()
    if batch:
        batch.close()
