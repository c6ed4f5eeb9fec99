"""CPU: the host logic behind the GPU chain (GpuChain::replay -> 2 x 5 AIS::Decoder with Reset mesh -> NMEA,
ais-catcher_amd/host/) fed with the checker's symbol decisions must print exactly the reference's NMEA text,
levels and ppm, message for message (golden fixture + seeded streams vs oracle/_ref)."""
import numpy as np
import pytest

import checkers
from ais_catcher_amd import host, synth


def _replay_blocks(m, chk, L, nblocks):
    """Feed the decisions recorded by a checker in the GPU's block structure (A's block, then B's)."""
    bits = [[chk.bits(ch, j) for j in range(5)] for ch in range(2)]
    ppm = [chk.tap_ppm(2), chk.tap_ppm(3)]
    W = L // 512
    for b in range(nblocks):
        g0, g1 = (b * L) // 5, ((b + 1) * L) // 5
        for ch in range(2):
            b5 = np.stack([bits[ch][j][0][g0:g1] for j in range(5)])
            m.replay(ch, g0, b * L, b5, bits[ch][0][1][g0:g1], ppm[ch][b * W:(b + 1) * W])


def test_replay_of_golden_decisions_gives_golden_nmea(golden):
    host.reset_sequence()
    m = host.ModelDefaultGPU(detached=True)
    block = int(golden["block_len"])
    L = block // 32
    W = L // 512
    for b in range(3):
        g0, g1 = (b * L) // 5, ((b + 1) * L) // 5
        for ch in range(2):
            b5 = np.stack([golden["bits_%d_%d" % (ch, j)][g0:g1] for j in range(5)])
            m.replay(ch, g0, b * L, b5, golden["lvl_%d_0" % ch][g0:g1], golden["ppm_" + "ab"[ch]][b * W:(b + 1) * W])
    assert m.nmea() == str(golden["nmea"]).split("\n")
    lvl, ppm = m.msg_meta()
    assert np.array_equal(lvl, golden["msg_level"]) and np.array_equal(ppm, golden["msg_ppm"])


@pytest.mark.parametrize("block,nblocks,rid", [(786432, 4, 0), (131072, 10, 1), (16384, 60, 2)])
def test_replay_matches_checker_on_streams(block, nblocks, rid):
    x = synth.receiver_stream(block * nblocks, receiver_id=rid, type5_every=3, gap_slots=(1, 3))
    chk = checkers.Ref(taps=True) if checkers.have_ref() else checkers.Oracle(taps=True)
    chk.feed_blocks(x, block)
    host.reset_sequence()
    m = host.ModelDefaultGPU(detached=True)
    _replay_blocks(m, chk, block // 32, nblocks)
    assert m.nmea() == chk.nmea() and len(chk.nmea()) >= 3
    a, b = m.msg_meta(), chk.msg_meta()
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_random_bits_never_crash_and_match_oracle_decoder():
    """Noise decisions exercise the abort paths (cannotBeValid, max frame length, CRC failures)."""
    rng = np.random.default_rng(5)
    x = (0.3 * (rng.standard_normal(131072 * 6) + 1j * rng.standard_normal(131072 * 6))).astype(np.complex64)
    chk = checkers.Oracle(taps=True)
    chk.feed_blocks(x, 131072)
    m = host.ModelDefaultGPU(detached=True)
    _replay_blocks(m, chk, 4096, 6)
    assert m.nmea() == chk.nmea()


def test_challenger_replay_matches_checker():
    """ModelChallenger wiring (20 decoders, throttled interleave of the coherent and the FM branch) fed with the
    checker's decisions == the checker's NMEA."""
    block, nblocks = 131072, 10
    x = synth.receiver_stream(block * nblocks, receiver_id=7, gap_slots=(1, 2), type5_every=4)
    chk = checkers.Ref(model=4, taps=True) if checkers.have_ref() else checkers.Oracle(model=4, taps=True)
    chk.feed_blocks(x, block)
    host.reset_sequence()
    m = host.ModelChallengerGPU(detached=True)
    L = block // 32
    W = L // 512
    bits = [[chk.bits(ch, j) for j in range(5)] for ch in range(2)]
    ppm = [chk.tap_ppm(2), chk.tap_ppm(3)]
    fm = []
    for ch in range(2):  # FM decoder j saw the samples N = j (mod 5): interleave them back into one stream
        per = [chk.bits(ch, j, 1)[0] for j in range(5)]
        n = min(len(p) for p in per) * 5
        f = np.zeros(n, np.float32)
        for j in range(5):
            f[j::5] = per[j][:n // 5]
        fm.append(f)
    for b in range(nblocks):
        g0, g1 = (b * L) // 5, ((b + 1) * L) // 5
        for ch in range(2):
            b5 = np.stack([bits[ch][j][0][g0:g1] for j in range(5)])
            m.replay(ch, g0, b * L, b5, bits[ch][0][1][g0:g1], ppm[ch][b * W:(b + 1) * W], fm=fm[ch][b * L:(b + 1) * L])
    assert m.nmea() == chk.nmea() and len(chk.nmea()) >= 5
    a, c = m.msg_meta(), chk.msg_meta()
    assert np.array_equal(a[1], c[1])


def test_model_base_replay_matches_checker():
    """ModelBase wiring (SimplePLL sampler switched fast/slow by its decoder) fed with the sign of the checker's filtered
    discriminator == the checker's NMEA: the sampler and the feedback loop are host logic."""
    block, nblocks = 131072, 12
    x = synth.receiver_stream(block * nblocks, receiver_id=9, gap_slots=(1, 2), type5_every=4)
    chk = checkers.Ref(model=1, taps=True) if checkers.have_ref() else checkers.Oracle(model=1, taps=True)
    chk.feed_blocks(x, block)
    host.reset_sequence()
    m = host.ModelBaseGPU(detached=True)
    L = block // 32
    W = L // 512
    fm = [chk.bits(ch, 0, 1)[0] for ch in range(2)]
    empty5 = np.zeros((5, 0), np.float32)
    for b in range(nblocks):
        for ch in range(2):
            m.replay(ch, (b * L) // 5, b * L, empty5, np.zeros(0, np.float32), np.zeros(W, np.float32), fm=fm[ch][b * L:(b + 1) * L])
    assert m.nmea() == chk.nmea() and len(chk.nmea()) >= 3
    a, c = m.msg_meta(), chk.msg_meta()
    assert np.array_equal(a[0], c[0]) and np.array_equal(a[1], c[1])


@pytest.mark.parametrize("block,nblocks,rid,kw", [(131072, 24, 51, {}), (16384, 96, 52, {"gap_slots": (0, 1)}), (786432, 4, 53, {"type5_every": 3})])
def test_v2_engine_host_logic_matches_checker(block, nblocks, rid, kw):
    """ModelEngineV2 (-m 11): the host engine (ais-catcher_amd/host/v2_engine.cpp) fed with the 48 kHz channels a checker
    recorded, in the block structure the device delivers them (channel A's block, then channel B's)."""
    x = synth.receiver_stream(block * nblocks, receiver_id=rid, **kw)
    chk = checkers.Ref(model=11, taps=True) if checkers.have_ref() else checkers.Oracle(model=11, taps=True)
    chk.feed_blocks(x, block)
    host.reset_sequence()
    m = host.ModelEngineV2GPU(block_len=block, detached=True)
    L = block // 32
    ta, tb = chk.tap(0), chk.tap(1)
    for b in range(nblocks):
        m.feed48(0, ta[b * L:(b + 1) * L])
        m.feed48(1, tb[b * L:(b + 1) * L])
    assert m.nmea() == chk.nmea() and len(chk.nmea()) >= 3
    a, c = m.msg_meta(), chk.msg_meta()
    assert np.array_equal(a[0], c[0]) and np.array_equal(a[1], c[1])
