"""The word-parallel frame evaluator of the device decoders (ais-catcher_amd/csrc/dec_core.h: dec_run_frame, what k7e_sim runs
inside a frame) against the symbol-by-symbol step of the same header, on random streams cut into random blocks -- the device
header built for the host (tests/dec_core_fuzz.cpp).  The step's own semantics (reference Marine/AIS.h:82-181, Marine/AIS.cpp:33-142)
are pinned by the GPU parity tests against the reference's decoders; this test needs no GPU."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def fuzz_binary(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("dec_core") / "dec_core_fuzz")
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(ROOT, "ais-catcher_amd", "csrc"),
                    os.path.join(ROOT, "tests", "dec_core_fuzz.cpp"), "-o", out], check=True)
    return out


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_frame_evaluator_equals_the_step(fuzz_binary, seed):
    r = subprocess.run([fuzz_binary, "250000", str(seed)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all equal" in r.stdout
    # the generator reaches what it is meant to reach: completed messages, abandoned frames, frames that cross blocks
    runs, messages, other, crossings = (int(v) for v in re.findall(r"\d+", r.stdout)[:4])
    assert runs == 250000 and messages > 10000 and other > 100000 and crossings > 50000, r.stdout


def test_lazy_frame_evaluator_equals_the_step(fuzz_binary):
    """The evaluator as K7b's boundary tasks call it: CRC register and seven-bit tail not kept up to date from word to word
    (dec_run_frame<LAZY>), formed by dec_fix_crc_tail() when the symbol-by-symbol step takes over."""
    r = subprocess.run([fuzz_binary, "250000", "5", "lazy"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all equal" in r.stdout


def test_idle_decoder_step_equals_the_step(fuzz_binary):
    """dec_step_idle -- the branch-free TRAINING / STARTFLAG step kv2_engine takes while no decoder of a channel is inside a frame --
    against dec_step on random states, field by field (incl. the frame registers a step that opens a frame clears)."""
    r = subprocess.run([fuzz_binary, "2000000", "7", "idle"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    steps, opened, flagged = (int(v) for v in re.findall(r"\d+", r.stdout)[:3])
    assert "all equal" in r.stdout and steps == 2000000 and opened > 10000 and flagged > 10000, r.stdout


def test_lean_decoder_pair_equals_the_step(fuzz_binary):
    """dec_lean_idle / dec_lean_data -- the decoder of kv2_engine's two-wave form (round 6: no CRC register kept per symbol, the rare
    events of a frame behind one branch) -- against dec_step, symbol by symbol over streams of noise, preambles, flags and stuffed payloads."""
    r = subprocess.run([fuzz_binary, "20000", "11", "lean"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    steps, opened, messages, abandoned = (int(v) for v in re.findall(r"\d+", r.stdout)[:4])
    assert "all equal" in r.stdout and steps > 5000000 and opened > 20000 and messages > 5000 and abandoned > 10000, r.stdout


@pytest.fixture(scope="module")
def scan_binary(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("dec_scan") / "dec_scan_fuzz")
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(ROOT, "ais-catcher_amd", "csrc"),
                    os.path.join(ROOT, "tests", "dec_scan_fuzz.cpp"), "-o", out], check=True)
    return out


@pytest.mark.parametrize("seed", [1, 2])
def test_segmented_candidate_scan_equals_the_row_scan(scan_binary, seed):
    """k7e_scan's sixteen segments per row (dec_scan_words + the combination of the lanes) against the scan as one loop."""
    r = subprocess.run([scan_binary, "40000", str(seed)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    rows, events, failures, runs, over = (int(v) for v in re.findall(r"\d+", r.stdout)[:5])
    assert "all equal" in r.stdout and rows == 40000 and failures > 100000 and runs > 100000 and over > 100, r.stdout


@pytest.fixture(scope="module")
def mesh_binary(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("dec_mesh") / "dec_mesh_fuzz")
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(ROOT, "ais-catcher_amd", "csrc"),
                    os.path.join(ROOT, "tests", "dec_mesh_fuzz.cpp"), "-o", out], check=True)
    return out


@pytest.mark.parametrize("seed", [1, 2])
def test_event_driven_decoders_equal_the_stepped_mesh(mesh_binary, seed):
    """The whole method of the device frame decoders -- candidate scan, one run per possible frame, the walk that decides in the
    reference's order which runs happened and what a sibling's Reset cuts short -- against five / ten decoders stepped symbol by
    symbol with their Reset mesh, over multi-block streams with decision errors (tests/dec_mesh_fuzz.cpp)."""
    r = subprocess.run([mesh_binary, "5000", str(seed)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    streams, blocks, messages, carried = (int(v) for v in re.findall(r"\d+", r.stdout)[:4])
    assert "all equal" in r.stdout and streams == 5000 and blocks > 10000 and messages > 5000 and carried > 20000, r.stdout
