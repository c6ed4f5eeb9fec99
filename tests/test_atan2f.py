"""CPU: the atan2f restatement used by the FM-branch kernel equals the host libm's atan2f bit for bit."""
import os
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))


def test_fdlibm_restatement_matches_host_libm():
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "at")
        subprocess.check_call(["gcc", "-O2", "-fno-fast-math", "-ffp-contract=off", os.path.join(HERE, "atan2f_restatement.c"), "-o", exe, "-lm"])
        out = subprocess.check_output([exe]).decode()
    assert "mismatch 0 / 10000000" in out, out
