"""CPU: oracle/ais_oracle.c against the compiled reference (oracle/_ref/libaisref_strict.so) on seeded
synthetic streams -- every tap, every hard bit, every level/ppm and the NMEA text, bit-exact.
Skipped where the compiled reference is absent (it is built from /root/reference by oracle/Makefile)."""
import numpy as np
import pytest

import checkers
from ais_catcher_amd import synth

pytestmark = pytest.mark.skipif(not checkers.have_ref("strict"), reason="oracle/_ref not built")


def _compare(model, rate, fmt, block, nblocks, rid, fm=False, dsk=False, ps_ema=True, ma=False, afc_wide=True, droop=True, fp_ds=False, **kw):
    x = synth.receiver_stream(block * nblocks, sample_rate=rate, receiver_id=rid, **kw)
    data = {"cu8": synth.to_cu8, "cs8": synth.to_cs8, "cs16": synth.to_cs16, "cf32": lambda v: v}[fmt](x)
    o = checkers.Oracle(model=model, rate=rate, fmt=fmt, taps=True, dsk=dsk, ps_ema=ps_ema, ma=ma, afc_wide=afc_wide, droop=droop, fp_ds=fp_ds)
    r = checkers.Ref(model=model, rate=rate, fmt=fmt, taps=True, dsk=dsk, ps_ema=ps_ema, ma=ma, afc_wide=afc_wide, droop=droop, fp_ds=fp_ds)
    o.feed_blocks(data, block)
    r.feed_blocks(data, block)
    for w in range(6):
        a, b = o.tap(w), r.tap(w)
        assert len(a) == len(b) and len(a) > 0
        assert np.array_equal(a.view(np.float32), b.view(np.float32)), "tap %d" % w
    for w in (2, 3):
        assert np.array_equal(o.tap_ppm(w), r.tap_ppm(w))
    for ch in range(2):
        for j in range(5):
            for f in ((0, 1) if fm else (0,)):
                ob, rb = o.bits(ch, j, f), r.bits(ch, j, f)
                assert len(ob[0]) == len(rb[0]) > 0
                for p, q in zip(ob, rb):
                    assert np.array_equal(p, q)
    assert o.nmea() == r.nmea()
    ol, rl = o.msg_meta(), r.msg_meta()
    assert np.array_equal(ol[0], rl[0]) and np.array_equal(ol[1], rl[1])
    return r.nmea()


def test_default_cf32_reference_block():
    lines = _compare(2, 1536000, "cf32", 786432, 3, rid=0, type5_every=4)
    assert len(lines) >= 10


def test_default_cu8_rtl_block():
    assert len(_compare(2, 1536000, "cu8", 131072, 8, rid=1)) >= 3


@pytest.mark.parametrize("fmt", ["cs8", "cs16"])
def test_default_signed_integer_formats(fmt):
    """Util::ConvertRAW's CS8 / CS16 branches (Utilities/StreamHelpers.cpp:91-106, Convert.cpp:266-286)"""
    assert len(_compare(2, 1536000, fmt, 131072, 8, rid=12)) >= 3


@pytest.mark.parametrize("model,rate,fmt,block,nblocks,afc_wide,droop,extra", [
    (2, 1536000, "cf32", 131072, 8, False, True, {}), (2, 1536000, "cf32", 131072, 8, True, False, {}), (2, 1536000, "cu8", 131072, 8, False, False, {}),
    (2, 768000, "cf32", 65536, 12, False, False, {}), (2, 6000000, "cf32", 786432, 4, False, False, {}), (2, 3072000, "cf32", 262144, 6, True, False, {}),
    (4, 1536000, "cf32", 131072, 8, False, False, {"fm": True}), (4, 6000000, "cf32", 786432, 4, False, True, {"fm": True}),
    (2, 1536000, "cu8", 131072, 8, True, False, {"fp_ds": True}), (2, 192000, "cf32", 16384, 16, False, False, {}), (2, 250000, "cf32", 24576 * 3, 6, False, False, {})])
def test_afc_wide_off_and_droop_off(model, rate, fmt, block, nblocks, afc_wide, droop, extra):
    """`-go AFC_WIDE off` (Model.cpp:536-540, 586-588: the CGF searches its 36 candidates around bin 0 instead of around the
    widest 133-bin energy window) and `-go DROOP off` (Model.cpp:384-386: every ladder wired without FilterComplex3Tap,
    Model.cpp:162-327), alone and together, on the direct, the resampled, the fixed-point and the decimate-by-3 ladders."""
    lines = _compare(model, rate, fmt, block, nblocks, rid=40, afc_wide=afc_wide, droop=droop, gap_slots=(1, 2), **extra)
    assert len(lines) >= 1


def test_default_small_blocks():
    _compare(2, 1536000, "cf32", 16384, 24, rid=2, gap_slots=(1, 1))


@pytest.mark.parametrize("rate", [768000, 384000, 192000, 3072000])
def test_default_other_ladders(rate):
    _compare(2, rate, "cf32", 512 * (rate // 48000) * 8, 3, rid=3, gap_slots=(1, 2))


def test_challenger_cf32():
    _compare(4, 1536000, "cf32", 131072, 8, rid=4, fm=True)


def test_default_6msps_upsampled():
    # 6,000,000 S/s -> 6.144M bucket with Upsample 125/128 (Model.cpp:183-189)
    lines = _compare(2, 6000000, "cf32", 786432, 4, rid=5)
    assert len(lines) >= 1


@pytest.mark.parametrize("rate,dsk,block", [(250000, False, 24576 * 3), (240000, False, 24576 * 2), (500000, True, 49152 * 2),
                                            (1000000, True, 98304 * 2), (2000000, True, 196608), (250000, False, 20000)])
def test_rates_upsampled_into_a_decimate_by_3_bucket(rate, dsk, block):
    # convert >> DS2.. >> US >> DSK >> ROT (Model.cpp:213-219, 254-259, 284-289, 311-313): 250k / 240k -> 288k; with `-go DSK on`
    # 500k -> 576k, 1 MSPS -> 1152k, 2 MSPS -> 2304k (instead of the next 2^k bucket); also an input block that is not a
    # whole number of the filter's 8192-sample output blocks
    lines = _compare(2, rate, "cf32", block, 6, rid=8, dsk=dsk, gap_slots=(1, 2))
    assert len(lines) >= 1


@pytest.mark.parametrize("rate,block", [(48000, 512 * 40), (96000, 1024 * 40), (192000, 2048 * 40), (40000, 512 * 40), (150000, 2048 * 30),
                                        (12000, 512 * 8), (48000, 3000)])
def test_channel_mode_x(rate, block):
    """`-c X` (Receiver.cpp:87-98, Model.cpp:35-107): one channel, already centred, 12k .. 192k -- convert >> [US] >> [DS2_2] >>
    [DS2_1] >> [FDC] >> FCIC5_a; the messages carry channel 'X'."""
    x = synth.receiver_stream(block * 8, sample_rate=rate, receiver_id=77, gap_slots=(1, 2), single_channel=True)
    o = checkers.Oracle(model=2, rate=rate, taps=True, mode_x=True)
    r = checkers.Ref(model=2, rate=rate, taps=True, mode_x=True)
    o.feed_blocks(x, block)
    r.feed_blocks(x, block)
    for w in (0, 2, 4):
        assert len(o.tap(w)) == len(r.tap(w)) > 0 and np.array_equal(o.tap(w).view(np.float32), r.tap(w).view(np.float32))
    for j in range(5):
        for a, b in zip(o.bits(0, j), r.bits(0, j)):
            assert np.array_equal(a, b)
    assert o.nmea() == r.nmea()
    if rate >= 40000:
        assert len(r.nmea()) >= 5 and all(l.split(",")[4] == "X" for l in r.nmea())


@pytest.mark.parametrize("rate,block", [(96000, 1024 * 48), (150000, 2048 * 30), (120000, 2048 * 24), (96000, 1024 * 7)])
def test_lowest_rates(rate, block):
    # 96 kSPS: convert >> ROT (Model.cpp:332-334); 96k < rate < 192k: US >> DS2_1 >> FDC(-0.8) >> ROT (Model.cpp:323-329)
    lines = _compare(2, rate, "cf32", block, 8, rid=9, gap_slots=(1, 2))
    assert len(lines) >= 1


@pytest.mark.parametrize("rate", [300000, 350000])
def test_default_rates_upsampled_into_the_384k_bucket(rate):
    # 288k < rate < 384k: Upsample on the converted input itself, then DS2_2, DS2_1, FDC(-1.1) (Model.cpp:295-301)
    lines = _compare(2, rate, "cf32", 4096 * 24, 5, rid=7, gap_slots=(1, 2))
    assert len(lines) >= 1


@pytest.mark.parametrize("rate", [10000000, 8000000])
def test_default_rates_above_6144k_upsampled(rate):
    # 10 MSPS (Airspy R2) / 8 MSPS -> 12.288M bucket: DS2_7 .. DS2_3, Upsample, DS2_2, DS2_1, FDC(-2.0) (Model.cpp:166-172)
    lines = _compare(2, rate, "cf32", 512 * 256 * 6, 4, rid=6, gap_slots=(1, 2))
    assert len(lines) >= 1


def test_strict_and_shipped_builds_decode_the_same():
    if not checkers.have_ref("fast"):
        pytest.skip("fast build missing")
    x = synth.receiver_stream(786432 * 3, receiver_id=11)
    a = checkers.Ref(kind="strict")
    b = checkers.Ref(kind="fast")
    a.feed_blocks(x, 786432)
    b.feed_blocks(x, 786432)
    assert sorted(a.nmea()) == sorted(b.nmea()) and len(a.nmea()) > 5


def test_decimate_by_3_ladder_288k():
    """DownsampleKFilter (DSP.cpp:160-189): the 288 kHz bucket needs no option; Rotate then runs on its 8192-sample blocks."""
    lines = _compare(2, 288000, "cf32", 24576 * 4, 6, rid=21, gap_slots=(1, 2))
    assert len(lines) >= 3


@pytest.mark.parametrize("rate", [576000, 1152000, 2304000])
def test_decimate_by_3_ladders_with_dsk(rate):
    _compare(2, rate, "cf32", 24576 * (rate // 288000) * 2, 6, rid=22, dsk=True, gap_slots=(1, 2))


def test_dsk_input_blocks_not_aligned_to_its_output_blocks():
    """input blocks that are not a multiple of 3 * 8192 samples: the phase carry (idx_in) and the partial output block"""
    _compare(2, 288000, "cf32", 40000, 12, rid=23, gap_slots=(1, 1))


@pytest.mark.parametrize("model, rate, fmt, block", [(2, 1536000, "cf32", 131072), (2, 1536000, "cu8", 100000), (2, 768000, "cs16", 65536),
                                                    (2, 2304000, "cf32", 196608), (4, 1536000, "cf32", 131072), (2, 250000, "cf32", 50000),
                                                    (2, 2400000, "cu8", 204800)])
def test_moving_average_downsampler(model, rate, fmt, block):
    """`-go MA on` (Model.cpp:122-126, DSP.cpp:60-82): integrate-and-dump to 96 kHz in blocks of 8192, then Rotate; integer and
    fractional ratios, input blocks that are not aligned to its output blocks."""
    lines = _compare(model, rate, fmt, block, 12 if block < 150000 else 8, rid=61, ma=True, fm=model == 4, gap_slots=(1, 2))
    assert len(lines) >= 2


def test_phase_search_boxcar():
    """`-go PS_EMA off`: Demod::PhaseSearch (Demod.cpp:103-170) instead of PhaseSearchEMA"""
    lines = _compare(2, 1536000, "cf32", 131072, 8, rid=24, ps_ema=False)
    assert len(lines) >= 3


def _compare_base(rate, fmt, block, nblocks, rid, **kw):
    """ModelBase (-m 1): FM discriminator -> 37-tap filter -> SimplePLL (fast/slow switched by the decoder) -> one decoder."""
    x = synth.receiver_stream(block * nblocks, sample_rate=rate, receiver_id=rid, **kw)
    data = synth.to_cu8(x) if fmt == "cu8" else x
    o = checkers.Oracle(model=1, rate=rate, fmt=fmt, taps=True)
    r = checkers.Ref(model=1, rate=rate, fmt=fmt, taps=True)
    o.feed_blocks(data, block)
    r.feed_blocks(data, block)
    for w in (0, 1):
        assert np.array_equal(o.tap(w).view(np.float32), r.tap(w).view(np.float32))
    for ch in range(2):
        fo, fr = o.bits(ch, 0, 1)[0], r.bits(ch, 0, 1)[0]      # filtered discriminator, every 48 kHz sample
        assert len(fo) == len(fr) > 0 and np.array_equal(fo, fr)
        so, sr = o.bits(ch, 0, 0), r.bits(ch, 0, 0)            # what the sampler hands to the decoder
        assert len(so[0]) == len(sr[0]) > 0 and np.array_equal(so[0], sr[0])
    assert o.nmea() == r.nmea()
    ol, rl = o.msg_meta(), r.msg_meta()
    assert np.array_equal(ol[0], rl[0]) and np.array_equal(ol[1], rl[1])
    return r.nmea()


def test_model_base_fm_receiver():
    assert len(_compare_base(1536000, "cf32", 131072, 12, rid=51, type5_every=4)) >= 3
    _compare_base(1536000, "cu8", 16384, 40, rid=52)


@pytest.mark.parametrize("rate,fmt,block", [(1536000, "cf32", 131072), (1536000, "cu8", 131072), (768000, "cf32", 65536)])
def test_model_standard(rate, fmt, block):
    """ModelStandard (-m 0, Model.cpp:484-518): FM discriminator -> 37-tap filter -> Deinterleave(5) -> five decoders with
    their Reset mesh; what every decoder receives and the NMEA text."""
    x = synth.receiver_stream(block * 10, sample_rate=rate, receiver_id=31, gap_slots=(1, 2), type5_every=4)
    data = synth.to_cu8(x) if fmt == "cu8" else x
    o = checkers.Oracle(model=0, rate=rate, fmt=fmt, taps=True)
    r = checkers.Ref(model=0, rate=rate, fmt=fmt, taps=True)
    o.feed_blocks(data, block)
    r.feed_blocks(data, block)
    for w in (0, 1):
        assert np.array_equal(o.tap(w).view(np.float32), r.tap(w).view(np.float32))
    for ch in range(2):
        for j in range(5):
            ob, rb = o.bits(ch, j, 1), r.bits(ch, j, 1)
            assert len(ob[0]) == len(rb[0]) > 0
            assert np.array_equal(ob[0], rb[0]) and np.array_equal(ob[2], rb[2])
    assert o.nmea() == r.nmea() and len(r.nmea()) >= 3
    ol, rl = o.msg_meta(), r.msg_meta()
    assert np.array_equal(ol[0], rl[0]) and np.array_equal(ol[1], rl[1])


def test_fixed_point_ladder_fp_ds():
    """`-go FP_DS on`: Downsample16_CU8 (DSP.cpp:499-651) in front of the rotator instead of the four float CIC5 stages"""
    block = 131072
    x = synth.to_cu8(synth.receiver_stream(block * 8, receiver_id=33, gap_slots=(0, 2)))
    o = checkers.Oracle(fmt="cu8", taps=True, fp_ds=True)
    r = checkers.Ref(fmt="cu8", taps=True, fp_ds=True)
    plain = checkers.Ref(fmt="cu8", taps=True)
    for c in (o, r, plain):
        c.feed_blocks(x, block)
    for w in range(6):
        assert np.array_equal(o.tap(w).view(np.float32), r.tap(w).view(np.float32)), "tap %d" % w
    assert not np.array_equal(r.tap(0).view(np.float32), plain.tap(0).view(np.float32))  # the truncating ladder is a different filter
    for ch in range(2):
        for j in range(5):
            for p, q in zip(o.bits(ch, j), r.bits(ch, j)):
                assert np.array_equal(p, q)
    assert o.nmea() == r.nmea() and len(r.nmea()) >= 3


@pytest.mark.parametrize("rate,fmt,block,rid,kw", [(1536000, "cf32", 131072, 41, {}), (1536000, "cf32", 16384, 42, {"gap_slots": (0, 1)}),
                                                  (1536000, "cu8", 786432, 43, {"type5_every": 3}), (768000, "cf32", 65536, 44, {}),
                                                  (6000000, "cf32", 786432, 45, {})])
def test_model_engine_v2(rate, fmt, block, rid, kw):
    """ModelEngineV2 (-m 11, Model.cpp:440-463): oracle/ais_oracle_v2.inc against V2::Engine of the compiled reference --
    the 48 kHz channels, the NMEA text, the levels and the ppm of every message."""
    n = max(6, 786432 * 4 // block)
    x = synth.receiver_stream(block * n, sample_rate=rate, receiver_id=rid, **kw)
    data = synth.to_cu8(x) if fmt == "cu8" else x
    o = checkers.Oracle(model=11, rate=rate, fmt=fmt, taps=True)
    r = checkers.Ref(model=11, rate=rate, fmt=fmt, taps=True)
    o.feed_blocks(data, block)
    r.feed_blocks(data, block)
    for w in (0, 1):
        assert np.array_equal(o.tap(w).view(np.float32), r.tap(w).view(np.float32))
    assert o.nmea() == r.nmea() and len(r.nmea()) >= 3
    ol, rl = o.msg_meta(), r.msg_meta()
    assert np.array_equal(ol[0], rl[0]) and np.array_equal(ol[1], rl[1])


@pytest.mark.parametrize("model,rate,fmt,block", [(1, 1536000, "cf32", 131072), (0, 768000, "cu8", 65536), (4, 1536000, "cf32", 131072),
                                                  (4, 6000000, "cf32", 786432)])
def test_fm_receiver_float_taps(model, rate, fmt, block):
    """Demod::FM output (taps 6/7) and Filter(Receiver) output (taps 8/9) of the FM receivers -- ModelBase, ModelStandard and
    the FM branch of ModelChallenger (Model.cpp:431-432, 500-503, 638-639): the restatement's floats == the reference's."""
    x = synth.receiver_stream(block * 5, sample_rate=rate, receiver_id=140 + model, gap_slots=(1, 2))
    data = synth.to_cu8(x) if fmt == "cu8" else x
    o = checkers.Oracle(model=model, rate=rate, fmt=fmt, taps=True)
    r = checkers.Ref(model=model, rate=rate, fmt=fmt, taps=True)
    o.feed_blocks(data, block)
    r.feed_blocks(data, block)
    for w in (6, 7, 8, 9):
        a, b = o.tapf(w), r.tapf(w)
        assert len(a) == len(b) > 0 and np.array_equal(a.view(np.uint32), b.view(np.uint32)), "tap %d" % w
    assert np.any(o.tapf(6) != 0)
