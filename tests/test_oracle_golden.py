"""CPU: the plain-C oracle (oracle/ais_oracle.c) against the committed golden fixtures, i.e. against
recorded outputs of the compiled reference itself (tests/golden/make_golden.py).  Bit-exact."""
import numpy as np

import checkers


def _run(golden, fmt):
    cu8 = golden["cu8"]
    block = int(golden["block_len"])
    o = checkers.Oracle(model=2, rate=1536000, fmt=fmt, taps=True)
    if fmt == "cu8":
        o.feed_blocks(cu8, block)
    else:  # the CU8 -> float conversion is exact, so the CF32 path must give identical results
        x = ((cu8.astype(np.int32) - 128) / np.float32(128.0)).astype(np.float32).view(np.complex64)
        o.feed_blocks(x, block)
    return o


def test_oracle_matches_golden_cu8(golden):
    o = _run(golden, "cu8")
    for w in range(6):
        assert np.array_equal(o.tap(w).view(np.float32), golden["tap%d" % w].view(np.float32)), "tap %d" % w
    assert np.array_equal(o.tap_ppm(2), golden["ppm_a"]) and np.array_equal(o.tap_ppm(3), golden["ppm_b"])
    for ch in range(2):
        for j in range(5):
            b, l, _ = o.bits(ch, j)
            assert np.array_equal(b.astype(np.int8), golden["bits_%d_%d" % (ch, j)])
            assert np.array_equal(l, golden["lvl_%d_%d" % (ch, j)])
    assert o.nmea() == str(golden["nmea"]).split("\n")
    lvl, ppm = o.msg_meta()
    assert np.array_equal(lvl, golden["msg_level"]) and np.array_equal(ppm, golden["msg_ppm"])


def test_oracle_matches_golden_cf32(golden):
    o = _run(golden, "cf32")
    for w in range(6):
        assert np.array_equal(o.tap(w).view(np.float32), golden["tap%d" % w].view(np.float32)), "tap %d" % w
    assert o.nmea() == str(golden["nmea"]).split("\n")


def test_golden_contains_pinned_payloads(golden):
    # payloads whose decoded fields the reference pins in python/tests/test_decode.py:12-17
    lines = str(golden["nmea"]).split("\n")
    assert any("15MgK45P3@G?fl0E`JbR0OwT0@MS" in l for l in lines)
    assert any(l.startswith("!AIVDM,2,1,") for l in lines) and any(l.startswith("!AIVDM,2,2,") for l in lines)
