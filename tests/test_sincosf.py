"""CPU: the sinf / cosf restatement used by the device V2 engine (std::polar of V2::FreqOffset::Derotate) equals the host libm bit for bit."""
import os
import subprocess
import tempfile

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _has_fma():
    try:
        return " fma " in open("/proc/cpuinfo").read()
    except OSError:
        return False


@pytest.mark.skipif(not _has_fma(), reason="glibc selects its non-FMA sinf / cosf on this CPU; the device restates the FMA variant (the GPU hosts' CPUs have FMA)")
def test_glibc_sincosf_restatement_matches_host_libm():
    """glibc picks its sinf / cosf by CPU (ifunc): on FMA CPUs the variant whose a * b + c are fused.  That is the one the device
    restates (with two rounded operations instead, 5 of 2 x 10^7 results differ in the last bit for |x| up to 100)."""
    flags = ["-ffp-contract=off", "-DUSE_FMA", "-mfma"]
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "sc")
        subprocess.check_call(["gcc", "-O2", "-fno-fast-math"] + flags + [os.path.join(HERE, "sincosf_restatement.c"), "-o", exe, "-lm"])
        out = subprocess.run([exe], capture_output=True).stdout.decode()
    assert "mismatch 0 / 20000000" in out, out
