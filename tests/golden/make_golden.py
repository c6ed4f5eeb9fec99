"""Generate tests/golden/*.npz from the COMPILED REFERENCE (oracle/_ref/libaisref_strict.so).

Run in the build container (needs /root/reference):  python tests/golden/make_golden.py
The fixtures pin the oracle and the HIP path to outputs of the reference itself: the reference's own
ModelDefault (strict IEEE flags) is driven with fixed 131072-sample CU8 blocks (the RTL-SDR block size,
Device/RTLSDR.h:57) and every stage's output is recorded by recorder sinks (oracle/ref_harness.cpp).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import _pkg  # noqa: E402

_pkg.load()
from ais_catcher_amd import synth  # noqa: E402
import checkers  # noqa: E402


def main():
    checkers.build_ref()
    block, nblocks = 131072, 3
    x = synth.receiver_stream(block * nblocks, receiver_id=7, gap_slots=(1, 2), type5_every=3)
    cu8 = synth.to_cu8(x)
    r = checkers.Ref(model=2, rate=1536000, fmt="cu8", taps=True, kind="strict")
    r.feed_blocks(cu8, block)
    out = dict(cu8=cu8, block_len=np.int64(block), nmea=np.array("\n".join(r.nmea())))
    for w in range(6):
        out["tap%d" % w] = r.tap(w)
    out["ppm_a"], out["ppm_b"] = r.tap_ppm(2), r.tap_ppm(3)
    for ch in range(2):
        for j in range(5):
            b, l, i = r.bits(ch, j)
            out["bits_%d_%d" % (ch, j)] = b.astype(np.int8)
            out["lvl_%d_%d" % (ch, j)] = l
    lvl, ppm = r.msg_meta()
    out["msg_level"], out["msg_ppm"] = lvl, ppm
    path = os.path.join(HERE, "modeldefault_cu8_1536k.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(r.nmea()), "NMEA lines")
    print("\n".join(r.nmea()))


if __name__ == "__main__":
    main()
