#!/usr/bin/env python3
"""tools/cpu_probe.py -- what the CPU baseline of bench.py sees on this host: topology, cgroup CPU limit, and the reference
chain's thread scaling with pinned threads (bench.cpu_baseline) and, for comparison, with unpinned ones."""
import ctypes
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import bench  # noqa: E402

for cmd in (["nproc"], ["sh", "-c", "lscpu | egrep 'Model name|Socket|Core|Thread|NUMA|MHz|L2|L3'"],
            ["sh", "-c", "cat /sys/fs/cgroup/cpu.max /sys/fs/cgroup/cpu.stat 2>/dev/null; cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null"],
            ["sh", "-c", "grep -E 'MemTotal|MemAvailable' /proc/meminfo; cat /proc/loadavg"]):
    print("$", " ".join(cmd))
    print(subprocess.run(cmd, capture_output=True, text=True).stdout)
print(json.dumps(bench.cpu_baseline(float(sys.argv[1]) if len(sys.argv) > 1 else 16.0), indent=1))

# unpinned threads at a few counts (what round 2 measured, minus Python)
import checkers  # noqa: E402
import _pkg  # noqa: E402
_pkg.load()
from ais_catcher_amd import synth  # noqa: E402
lib = ctypes.CDLL(os.path.join(checkers.ORACLE_DIR, "_ref", "libaisref_fast.so"))
lib.ref_bench_threads.restype = ctypes.c_double
lib.ref_bench_threads.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p] + [ctypes.c_int] * 3 + [ctypes.c_void_p, ctypes.c_double, ctypes.c_void_p]
x = np.ascontiguousarray(synth.receiver_stream(bench.BLOCK * 4, receiver_id=4242)).astype(np.complex64)
ncpu = len(os.sched_getaffinity(0))
for n in sorted(set([min(ncpu, c) for c in (16, 32, 64, 128, 256)])):
    counts = np.zeros(n, np.int64)
    dt = lib.ref_bench_threads(2, bench.RATE, 1, x.ctypes.data, 4, bench.BLOCK * 8, n, None, 1.5, counts.ctypes.data)
    print("unpinned %3d threads: %8.1f MS/s  (%.1f per thread, min/max blocks per thread %d/%d)" % (n, counts.sum() * bench.BLOCK / dt / 1e6,
          counts.sum() * bench.BLOCK / dt / 1e6 / n, counts.min(), counts.max()))
