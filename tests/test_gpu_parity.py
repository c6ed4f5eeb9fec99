"""GPU (-m gpu): the HIP chain, called through the C ABI (libaisgpu.so), against the oracle and the
golden fixtures recorded from the compiled reference.  Everything is compared BIT-EXACT: the float
taps are required to be identical binary32 values (north_star tolerance 1e-5 rel is therefore met
with margin 0), hard bits / levels / ppm identical.
"""
import numpy as np
import pytest

import checkers
from ais_catcher_amd import gpu, synth

pytestmark = pytest.mark.gpu

try:  # torch (used for resident device buffers) bundles its own HIP runtime: let it initialise first
    import torch
    if torch.cuda.is_available():
        torch.cuda.init()
except Exception:  # pragma: no cover
    torch = None


_FMT = {"cu8": gpu.FMT_CU8, "cf32": gpu.FMT_CF32, "cs8": gpu.FMT_CS8, "cs16": gpu.FMT_CS16}


def _feq(a, b):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    return a.shape == b.shape and np.array_equal(a.view(np.float32), b.view(np.float32))


def _run_gpu_vs_oracle(streams, rate, fmt, block, nblocks, **kw):
    """streams: list of per-receiver arrays.  Compares every block's taps and outputs per receiver."""
    R = len(streams)
    okw = dict(dsk=kw.get("dsk", False), ps_ema=kw.get("ps_ema", True), fp_ds=kw.get("fp_ds", False), ma=kw.get("ma", False),
               afc_wide=kw.get("afc_wide", True), droop=kw.get("droop", True))
    g = gpu.AisGpu(sample_rate=rate, n_receivers=R, block_len=block,
                   input_format=_FMT[fmt], taps=True, **kw)
    per = 1 if fmt == "cf32" else 2
    oracles = [checkers.Oracle(model=2, rate=rate, fmt=fmt, taps=True, **okw) for _ in range(R)]
    for o, x in zip(oracles, streams):
        o.feed_blocks(x, block)
    otaps = [[o.tap(w) for w in range(6)] for o in oracles]
    oppm = [[o.tap_ppm(2), o.tap_ppm(3)] for o in oracles]
    obits = [[[o.bits(ch, j) for j in range(5)] for ch in range(2)] for o in oracles]
    L = block // (rate // 48000)
    W = L // 512
    gdone = [[0, 0] for _ in range(R)]
    for b in range(nblocks):
        for r in range(R):
            g.submit(r, streams[r][b * block * per:(b + 1) * block * per])
        g.run()
        g.sync_outputs()
        for r in range(R):
            for w in (0, 1, 2, 3):
                assert _feq(g.tap(w, r), otaps[r][w][b * L:(b + 1) * L]), "block %d rx %d tap %d" % (b, r, w)
            for ch in range(2):
                out = g.fetch(r, ch)
                assert out["first_sample48"] == b * L
                assert out["first_group"] == gdone[r][ch]
                n = out["n_groups"]
                assert n == ((b + 1) * L) // 5 - (b * L) // 5
                g0 = gdone[r][ch]
                fir = g.tap(4 + ch, r)
                assert _feq(fir, otaps[r][4 + ch][5 * g0:5 * (g0 + n)]), "block %d rx %d FIR ch %d" % (b, r, ch)
                for j in range(5):
                    ob, ol, oi = obits[r][ch][j]
                    assert np.array_equal(out["bits"][j], ob[g0:g0 + n]), "bits b%d r%d c%d j%d" % (b, r, ch, j)
                    assert np.array_equal(oi[g0:g0 + n], 5 * (g0 + np.arange(n)) + j)
                    assert _feq(out["lvl"], ol[g0:g0 + n]), "lvl b%d r%d c%d" % (b, r, ch)
                assert _feq(out["ppm"], oppm[r][ch][b * W:(b + 1) * W]), "ppm b%d r%d c%d" % (b, r, ch)
                gdone[r][ch] += n
    g.close()


def test_golden_cu8(golden):
    """3 RTL-size CU8 blocks: GPU taps/bits/levels/ppm == recorded outputs of the compiled reference."""
    cu8, block = golden["cu8"], int(golden["block_len"])
    L = block // 32
    g = gpu.AisGpu(n_receivers=1, block_len=block, input_format=gpu.FMT_CU8, taps=True)
    gd = [0, 0]
    for b in range(3):
        g.submit(0, cu8[b * block * 2:(b + 1) * block * 2])
        g.run()
        g.sync_outputs()
        for w in range(4):
            assert _feq(g.tap(w), golden["tap%d" % w][b * L:(b + 1) * L]), "tap %d block %d" % (w, b)
        for ch in range(2):
            out = g.fetch(0, ch)
            n, g0 = out["n_groups"], gd[ch]
            assert _feq(g.tap(4 + ch), golden["tap%d" % (4 + ch)][5 * g0:5 * (g0 + n)])
            for j in range(5):
                assert np.array_equal(out["bits"][j].astype(np.int8), golden["bits_%d_%d" % (ch, j)][g0:g0 + n])
            assert _feq(out["lvl"], golden["lvl_%d_0" % ch][g0:g0 + n])
            assert _feq(out["ppm"], golden["ppm_" + "ab"[ch]][b * 8:(b + 1) * 8])
            gd[ch] += n
    g.close()


def test_cf32_reference_block_two_receivers():
    xs = [synth.receiver_stream(786432 * 3, receiver_id=r, type5_every=5) for r in (0, 1)]
    _run_gpu_vs_oracle(xs, 1536000, "cf32", 786432, 3)


@pytest.mark.parametrize("fmt,rate,block", [("cs8", 1536000, 131072), ("cs16", 1536000, 131072), ("cs16", 768000, 65536),
                                            ("cs8", 192000, 16384), ("cs16", 3072000, 262144), ("cs8", 6000000, 786432),
                                            ("cs16", 288000, 49152)])
def test_signed_integer_formats(fmt, rate, block):
    """CS8 / CS16 input (Util::ConvertRAW, Utilities/StreamHelpers.cpp:91-106 + Convert.cpp:266-286) converted inside the
    front end: direct ladders, a pre-decimated one, the resampled one and a decimate-by-3 one."""
    x = synth.receiver_stream(block * 3, sample_rate=rate, receiver_id=61, gap_slots=(0, 2))
    data = synth.to_cs8(x) if fmt == "cs8" else synth.to_cs16(x)
    if rate == 6000000:
        _run_multi_sub(data, rate, block, 3, fmt=fmt)
    else:
        _run_gpu_vs_oracle([data], rate, fmt, block, 3)


def test_fixed_point_ladder_fp_ds():
    """`-go FP_DS on` (-F): 1536 kSPS CU8 through Downsample16_CU8, the SWAR fixed-point CIC5 ladder (DSP.cpp:499-651),
    evaluated with packed 16-bit fields in the front end's registers; taps, hard bits and the NMEA text."""
    from ais_catcher_amd import host
    xs = [synth.to_cu8(synth.receiver_stream(131072 * 6, receiver_id=64 + r, gap_slots=(0, 2))) for r in range(2)]
    _run_gpu_vs_oracle(xs, 1536000, "cu8", 131072, 6, fp_ds=True)
    _run_gpu_vs_oracle(xs[:1], 1536000, "cu8", 16384, 12, fp_ds=True)
    edge = np.zeros(2 * 16384 * 4, np.uint8)
    edge[0::4] = 255
    edge[1::7] = 1
    _run_gpu_vs_oracle([edge], 1536000, "cu8", 16384, 4, fp_ds=True)
    chk = checkers.Ref(fmt="cu8", fp_ds=True) if checkers.have_ref() else checkers.Oracle(fmt="cu8", fp_ds=True)
    chk.feed_blocks(xs[0], 131072)
    host.reset_sequence()
    m = host.ModelDefaultGPU(block_len=131072, input_format=gpu.FMT_CU8, fp_ds=True)
    for b in range(6):
        m.receive(xs[0][b * 262144:(b + 1) * 262144])
    assert m.nmea() == chk.nmea() and len(chk.nmea()) >= 2


def test_cu8_equals_cf32_path():
    x = synth.receiver_stream(131072 * 4, receiver_id=3, gap_slots=(1, 2))
    _run_gpu_vs_oracle([synth.to_cu8(x)], 1536000, "cu8", 131072, 4)


@pytest.mark.parametrize("tps", [1, 5, 24])
def test_span_tiling_is_invisible(tps):
    """Time tiling (spans + warm-up tile) must not change a single bit."""
    x = synth.receiver_stream(98304 * 3, receiver_id=4, gap_slots=(1, 1))
    _run_gpu_vs_oracle([x], 1536000, "cf32", 98304, 3, tiles_per_span=tps)


@pytest.mark.parametrize("rate", [768000, 384000, 192000])
def test_other_ladders(rate):
    block = 512 * (rate // 48000) * 6
    x = synth.receiver_stream(block * 3, sample_rate=rate, receiver_id=5, gap_slots=(1, 2))
    _run_gpu_vs_oracle([x], rate, "cf32", block, 3)
    _run_gpu_vs_oracle([synth.to_cu8(x)], rate, "cu8", block, 3)


def test_edge_inputs():
    """All-zero, DC, full-scale alternating and a huge-dynamic-range burst: still bit-exact."""
    n = 16384 * 4
    rng = np.random.default_rng(1)
    zero = np.zeros(n, np.complex64)
    dc = np.full(n, 0.25 - 0.5j, np.complex64)
    alt = (np.where(np.arange(n) % 2 == 0, 1.0, -1.0) * (1 + 1j)).astype(np.complex64)
    wild = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    wild[:n // 2] *= np.float32(1e-18)
    wild[n // 2:] *= np.float32(3e4)
    _run_gpu_vs_oracle([zero, dc, alt, wild], 1536000, "cf32", 16384, 4)
    cu = np.zeros(2 * n, np.uint8)
    cu[0::4] = 255
    _run_gpu_vs_oracle([cu], 1536000, "cu8", 16384, 4)


def test_edge_inputs_through_the_chunked_phase_search():
    """Blocks long enough for the chunk-parallel PhaseSearchEMA (several chunks per block, speculative warm-ups that are verified):
    exact zeros (t == +-0 must count as "not > 0"), DC, alternating full scale, 10^22 dynamic range."""
    n = 131072 * 3
    rng = np.random.default_rng(11)
    zero = np.zeros(n, np.complex64)
    dc = np.full(n, 0.25 - 0.5j, np.complex64)
    alt = (np.where(np.arange(n) % 2 == 0, 1.0, -1.0) * (1 + 1j)).astype(np.complex64)
    wild = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    wild[:n // 3] *= np.float32(1e-18)
    wild[n // 3:2 * n // 3] *= np.float32(3e4)
    half = zero.copy()
    half[n // 2:] = wild[n // 2:]
    _run_gpu_vs_oracle([zero, dc, alt, wild, half], 1536000, "cf32", 131072, 3)


def test_noise_only_long():
    rng = np.random.default_rng(2)
    x = (0.05 * (rng.standard_normal(786432 * 2) + 1j * rng.standard_normal(786432 * 2))).astype(np.complex64)
    _run_gpu_vs_oracle([x], 1536000, "cf32", 786432, 2)


def test_full_batch_256_resident_matches_single():
    """BASELINE config 4 geometry (256 receivers x 786,432 samples, resident in HBM): every receiver of
    a batch fed the same stream must produce identical outputs, equal to the oracle's (size-independent
    property: batch invariance + checksum)."""
    import torch
    R, block = 256, 786432
    x = synth.receiver_stream(block * 2, receiver_id=9)
    o = checkers.Oracle(taps=True)
    o.feed_blocks(x, block)
    g = gpu.AisGpu(n_receivers=R, block_len=block, taps=True)
    gd = 0
    for b in range(2):
        one = torch.from_numpy(x[b * block:(b + 1) * block].view(np.float32).copy()).cuda()
        batch = one.unsqueeze(0).expand(R, -1).contiguous()
        torch.cuda.synchronize()
        g.submit_device(batch.data_ptr(), block)
        g.run()
        g.sync_outputs()
        L = block // 32
        ref = [g.fetch(0, ch) for ch in range(2)]
        for ch in range(2):
            n = ref[ch]["n_groups"]
            for j in range(5):
                assert np.array_equal(ref[ch]["bits"][j], o.bits(ch, j)[0][gd:gd + n])
            assert _feq(ref[ch]["lvl"], o.bits(ch, 0)[1][gd:gd + n])
        assert _feq(g.tap(0, 0), o.tap(0)[b * L:(b + 1) * L])
        for r in (1, 17, 128, 255):
            assert _feq(g.tap(0, r), g.tap(0, 0)) and _feq(g.tap(3, r), g.tap(3, 0))
        for r in range(1, R):
            for ch in range(2):
                out = g.fetch(r, ch)
                assert np.array_equal(out["bits"], ref[ch]["bits"]) and _feq(out["lvl"], ref[ch]["lvl"])
                assert _feq(out["ppm"], ref[ch]["ppm"])
        gd += ref[0]["n_groups"]
        del batch, one
    g.close()


def test_nmea_end_to_end_single_receiver():
    """ModelDefaultGPU (C++ host: GpuChain -> AIS::Decoder x10 -> NMEA) == the checker's NMEA, line for line."""
    from ais_catcher_amd import host
    block, nblocks = 786432, 4
    x = synth.receiver_stream(block * nblocks, receiver_id=12, type5_every=4)
    chk = checkers.Ref() if checkers.have_ref() else checkers.Oracle()
    chk.feed_blocks(x, block)
    host.reset_sequence()
    m = host.ModelDefaultGPU(block_len=block)
    for b in range(nblocks):
        m.receive(x[b * block:(b + 1) * block])
    assert m.nmea() == chk.nmea() and len(chk.nmea()) >= 10
    a, c = m.msg_meta(), chk.msg_meta()
    assert np.array_equal(a[0], c[0]) and np.array_equal(a[1], c[1])
    m.close()


def test_nmea_end_to_end_cu8_file_blocks():
    """BASELINE config 1 geometry: CU8 RAW-file blocks of 3,145,728 IQ samples (Device/FileRAW.h:43)."""
    from ais_catcher_amd import host
    block, nblocks = 3145728, 2
    x = synth.to_cu8(synth.receiver_stream(block * nblocks, receiver_id=13))
    chk = checkers.Ref(fmt="cu8") if checkers.have_ref() else checkers.Oracle(fmt="cu8")
    chk.feed_blocks(x, block)
    host.reset_sequence()
    m = host.ModelDefaultGPU(block_len=block, input_format=gpu.FMT_CU8)
    for b in range(nblocks):
        m.receive(x[b * block * 2:(b + 1) * block * 2])
    assert m.nmea() == chk.nmea() and len(chk.nmea()) >= 20
    m.close()


def test_nmea_batched_receivers_on_threads():
    """Four receivers sharing one GPU context, each driven from its own thread like the reference's device
    threads (Device/FileRAW.cpp:205-206); per-receiver NMEA lists equal the checker's."""
    import threading
    from ais_catcher_amd import host
    R, block, nblocks = 4, 131072, 6
    xs = [synth.receiver_stream(block * nblocks, receiver_id=30 + r, gap_slots=(1, 2)) for r in range(R)]
    want = []
    for x in xs:
        c = checkers.Oracle()
        c.feed_blocks(x, block)
        want.append(c.nmea())
    batch = host.Batch(n_receivers=R, block_len=block)
    models = [host.ModelDefaultGPU(block_len=block, batch=batch, rx=r) for r in range(R)]

    def run(r):
        for b in range(nblocks):
            models[r].receive(xs[r][b * block:(b + 1) * block])

    th = [threading.Thread(target=run, args=(r,)) for r in range(R)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for r in range(R):
        assert models[r].nmea() == want[r] and len(want[r]) >= 2
        models[r].close()
    batch.close()


@pytest.mark.parametrize("env", [{"AISGPU_PS_WARM": "16"}, {"AISGPU_PS_WARM": "64"}, {"AISGPU_PS_SEQUENTIAL": "1"}, {"AISGPU_SERIAL": "1"},
                                 {"AISGPU_SERIAL": "1", "AISGPU_PS_WARM": "16"}])
def test_phase_search_fallback_and_variants(env, monkeypatch):
    """The chunk-parallel PhaseSearchEMA verifies its speculative warm-ups; with a 16-symbol warm-up the check
    must fail and the sequential fallback must still deliver bit-exact decisions.  Also: the plain sequential
    kernel and the single-stream (non-overlapped) schedule.  (AISGPU_<KEY>: test hooks, forwarded by the Python wrapper to
    aisgpu_set_option -- the library reads no environment variable.)"""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    xs = [synth.receiver_stream(786432 * 2, receiver_id=40 + r) for r in range(2)]
    _run_gpu_vs_oracle(xs, 1536000, "cf32", 786432, 2)


@pytest.mark.parametrize("warm, expect_fallbacks", [(None, False), ("16", True)])
def test_phase_search_fallback_counter(warm, expect_fallbacks, monkeypatch):
    """aisgpu_ps_fallbacks(): with the default warm-up of 256 symbols no workgroup of a burst-carrying batch needs the exact
    sequential kernel (a warm-up of 192 already costs one per block of 640, 160 a dozen per hundred -- profiles/r02_expZ.txt);
    a 16-symbol warm-up sends practically all of them through it.  The decisions are the oracle's either way."""
    if warm:
        monkeypatch.setenv("AISGPU_PS_WARM", warm)
    R, block, nb = 8, 786432, 3
    xs = [synth.receiver_stream(block * nb, receiver_id=70 + r) for r in range(R)]
    g = gpu.AisGpu(sample_rate=1536000, n_receivers=R, block_len=block)
    os_ = []
    for r in range(R):
        o = checkers.Oracle(model=2, rate=1536000, fmt="cf32", taps=True)
        o.feed_blocks(xs[r], block)
        os_.append(o)
    got = [[[] for _ in range(2)] for _ in range(R)]
    for b in range(nb):
        for r in range(R):
            g.submit(r, xs[r][b * block:(b + 1) * block])
        g.run()
        g.sync_outputs()
        for r in range(R):
            for ch in range(2):
                got[r][ch].append(g.fetch(r, ch)["bits"])
    n = g.ps_fallbacks()
    g.close()
    for r in range(R):
        for ch in range(2):
            have = np.concatenate(got[r][ch], axis=1)
            for j in range(5):
                want = os_[r].bits(ch, j)[0]
                n_cmp = min(len(want), have.shape[1])
                assert n_cmp > 14000 and np.array_equal(have[j, :n_cmp], want[:n_cmp])
    assert (n > 0) if expect_fallbacks else (n == 0), n


def _run_multi_sub(x, rate, block, nblocks, fmt="cf32", dsk=False):
    """Rates whose ladder contains the resampler or a pre-decimation pass: compare every completed downstream
    block (there can be 1 or 2 per input block) with the oracle's stream."""
    g = gpu.AisGpu(sample_rate=rate, n_receivers=1, block_len=block,
                   input_format=_FMT[fmt], taps=True, dsk=dsk)
    per = 1 if fmt == "cf32" else 2
    o = checkers.Oracle(model=2, rate=rate, fmt=fmt, taps=True, dsk=dsk)
    o.feed_blocks(x, block)
    otap = [o.tap(w) for w in range(6)]
    oppm = [o.tap_ppm(2), o.tap_ppm(3)]
    obits = [[o.bits(ch, j) for j in range(5)] for ch in range(2)]
    n48 = 0
    gd = 0
    subs = []
    for b in range(nblocks):
        g.submit(0, x[b * block * per:(b + 1) * block * per])
        g.run()
        g.sync_outputs()
        ns = g.out_count()
        subs.append(ns)
        for s in range(ns):
            outs = [g.fetch(0, ch, s) for ch in range(2)]
            L = outs[0]["n_windows"] * 512
            assert outs[0]["first_sample48"] == n48
            n = outs[0]["n_groups"]
            for ch in range(2):
                for j in range(5):
                    assert np.array_equal(outs[ch]["bits"][j], obits[ch][j][0][gd:gd + n]), "bits blk %d sub %d ch %d j %d" % (b, s, ch, j)
                assert _feq(outs[ch]["lvl"], obits[ch][0][1][gd:gd + n])
                assert _feq(outs[ch]["ppm"], oppm[ch][n48 // 512:(n48 + L) // 512])
            if s == ns - 1:  # float taps are readable for the last downstream block of a run
                for w in range(4):
                    assert _feq(g.tap(w), otap[w][n48:n48 + L]), "tap %d blk %d" % (w, b)
            n48 += L
            gd += n
    assert n48 > 0 and len(otap[0]) >= n48
    g.close()
    return subs


def test_6msps_resampled_ladder():
    """BASELINE config 3 front end: 6,000,000 S/s -> 4 x CIC5 -> Upsample 125/128 -> DS2_2 -> DS2_1 -> FDC(-2.0) -> ...
    (Model.cpp:183-189).  One input block completes 1 or 2 downstream blocks."""
    block = 786432
    x = synth.receiver_stream(block * 5, sample_rate=6000000, receiver_id=50)
    subs = _run_multi_sub(x, 6000000, block, 5)
    assert set(subs) <= {1, 2} and sum(subs) == (5 * 49152 * 128 // 125) // 49152


@pytest.mark.parametrize("rate", [3072000, 6144000, 12288000])
def test_deep_pure_ladders(rate):
    block = 512 * (rate // 48000) * 4
    x = synth.receiver_stream(block * 3, sample_rate=rate, receiver_id=51, gap_slots=(1, 1))
    _run_multi_sub(x, rate, block, 3)
    _run_multi_sub(synth.to_cu8(x), rate, block, 3, fmt="cu8")


@pytest.mark.parametrize("rate", [3072000, 6144000])
def test_deep_ladders_in_one_pass_smallest_blocks(rate):
    """Five / six CIC5 stages in the front-end waves (k1_dpp<5>, <6>: tiles of 2,048 / 4,096 samples) with blocks of ONE 512-sample
    window (16 tiles: every span is a single window, the warm-up tile of every block is the previous block's last tile) and of three
    windows, three receivers of which one is silence and one is reversed; CF32 (LDS-DMA tiles) and CS16 (register path)."""
    for windows, nb in ((1, 14), (3, 5)):
        block = 512 * (rate // 48000) * windows
        x = synth.receiver_stream(block * nb, sample_rate=rate, receiver_id=54 + windows, gap_slots=(0, 1))
        _run_outputs_vs_oracle([x, np.zeros_like(x), x[::-1].copy()], rate, "cf32", block, nb)
        _run_outputs_vs_oracle([synth.to_cs16(x)], rate, "cs16", block, nb)


@pytest.mark.parametrize("rate", [2000000, 1000000, 2400000])
def test_other_resampled_rates(rate):
    """Rates whose resampler increment is NOT exactly representable: the host replays the float accumulation."""
    bucket = 3072000 if rate > 1536000 else 1536000
    block = 512 * (bucket // 48000) * 8
    x = synth.receiver_stream(block * 4, sample_rate=rate, receiver_id=52, gap_slots=(1, 2))
    _run_multi_sub(x, rate, block, 4)


@pytest.mark.parametrize("rate,block,fmt", [(48000, 512 * 40, "cf32"), (96000, 1024 * 40, "cf32"), (192000, 2048 * 40, "cu8"),
                                            (40000, 512 * 40, "cf32"), (150000, 2048 * 30, "cs16"), (25000, 512 * 16, "cf32"), (12000, 512 * 8, "cf32"), (16000, 512 * 12, "cu8"), (20000, 512 * 16, "cf32")])
def test_channel_mode_x(rate, block, fmt):
    """`-c X` (Model.cpp:35-107): one already centred channel at 48 / 96 / 192 kSPS or resampled into the next of these; the
    single-channel front end K1x feeds channel A, channel B stays silent (below 24 kSPS one input block completes up to four
    downstream blocks: their outputs are copied out as they complete).  48 kHz tap, hard bits, levels, ppm per downstream
    block, then NMEA (channel letter X) through the host model."""
    from ais_catcher_amd import host
    nblocks = 8
    x = synth.receiver_stream(block * nblocks, sample_rate=rate, receiver_id=77, gap_slots=(1, 2), single_channel=True)
    data = {"cu8": synth.to_cu8, "cs16": synth.to_cs16, "cf32": lambda v: v}[fmt](x)
    per = 1 if fmt == "cf32" else 2
    o = checkers.Oracle(model=2, rate=rate, fmt=fmt, taps=True, mode_x=True)
    o.feed_blocks(data, block)
    g = gpu.AisGpu(sample_rate=rate, n_receivers=1, block_len=block, input_format=_FMT[fmt], taps=True, mode_x=True)
    otap, oppm = o.tap(0), o.tap_ppm(2)
    obits = [o.bits(0, j) for j in range(5)]
    n48 = gd = 0
    for b in range(nblocks):
        g.submit(0, data[b * block * per:(b + 1) * block * per])
        g.run()
        g.sync_outputs()
        ns = g.out_count()
        for s_ in range(ns):
            out = g.fetch(0, 0, s_)
            L, n = out["n_windows"] * 512, out["n_groups"]
            assert out["first_sample48"] == n48
            for j in range(5):
                assert np.array_equal(out["bits"][j], obits[j][0][gd:gd + n]), "bits blk %d sub %d j %d" % (b, s_, j)
            assert _feq(out["lvl"], obits[0][1][gd:gd + n]) and _feq(out["ppm"], oppm[n48 // 512:(n48 + L) // 512])
            if s_ == ns - 1:
                assert _feq(g.tap(0), otap[n48:n48 + L])
            silent = g.fetch(0, 1, s_)
            assert not np.any(silent["lvl"])  # channel B: nothing
            n48 += L
            gd += n
    g.close()
    if rate >= 40000:  # (below that the synthetic bursts are under-sampled for the decoders; the chain is still compared above)
        chk = checkers.Ref(model=2, rate=rate, fmt=fmt, mode_x=True) if checkers.have_ref() else o
        if chk is not o:
            chk.feed_blocks(data, block)
        host.reset_sequence()
        m = host.ModelDefaultGPU(sample_rate=rate, block_len=block, input_format=_FMT[fmt], ch1="X", ch2="X", mode_x=True)
        for b in range(nblocks):
            m.receive(data[b * block * per:(b + 1) * block * per])
        assert m.nmea() == chk.nmea() and len(chk.nmea()) >= 10
        m.close()


@pytest.mark.parametrize("rate,block,nblocks,R,gpu_decode", [(96000, 1024 * 48, 4, 3, False), (192000, 2048 * 24, 5, 5, False), (40000, 512 * 40, 6, 3, False),
                                                           (16000, 512 * 12, 8, 2, False), (96000, 1024 * 48, 4, 3, True), (150000, 2048 * 30, 5, 4, True)])
def test_channel_mode_x_receivers_packed(rate, block, nblocks, R, gpu_decode):
    """Round 6: in channel mode X the receivers of a batch are packed -- chain r of everything behind the 48 kHz channels IS receiver r,
    instead of every receiver dragging a silent channel B through the back end.  Distinct receivers (odd counts: one silent row pads
    the batch for the kernels that take channels in pairs), the fused default path, every downstream block (up to four per input block
    below 24 kSPS): hard bits, levels, ppm per receiver against the oracle; aisgpu_fetch(rx, 1) hands out silence; with the frame
    decoders on the device the frames carry the right receiver and print the oracle's NMEA (channel letter X)."""
    from ais_catcher_amd import host
    xs = [synth.receiver_stream(block * nblocks, sample_rate=rate, receiver_id=380 + r, gap_slots=(1, 2), single_channel=True) for r in range(R)]
    oracles = []
    for x in xs:
        o = checkers.Oracle(model=2, rate=rate, fmt="cf32", taps=True, mode_x=True)
        o.feed_blocks(x, block)
        oracles.append(o)
    g = gpu.AisGpu(sample_rate=rate, n_receivers=R, block_len=block, mode_x=True, gpu_decode=gpu_decode)
    hms = [host.ModelDefaultGPU(sample_rate=rate, detached=True, gpu_decode=True, ch1="X", ch2="X", mode_x=True) for _ in range(R)] if gpu_decode else []
    gd, wd = [0] * R, [0] * R
    for b in range(nblocks):
        for r in range(R):
            g.submit(r, xs[r][b * block:(b + 1) * block])
        g.run()
        g.sync_outputs()
        for s_ in range(g.out_count()):
            for r in range(R):
                out, o = g.fetch(r, 0, s_), oracles[r]
                n, W = out["n_groups"], out["n_windows"]
                assert out["first_group"] == gd[r]
                for j in range(5):
                    assert np.array_equal(out["bits"][j], o.bits(0, j)[0][gd[r]:gd[r] + n]), "bits b%d s%d r%d j%d" % (b, s_, r, j)
                assert _feq(out["lvl"], o.bits(0, 0)[1][gd[r]:gd[r] + n]) and _feq(out["ppm"], o.tap_ppm(2)[wd[r]:wd[r] + W])
                silent = g.fetch(r, 1, s_)
                assert silent["n_groups"] == n and not np.any(silent["lvl"]) and not any(np.any(silent["bits"][j] > 0) for j in range(5))
                gd[r] += n
                wd[r] += W
        if gpu_decode:
            for f in g.frames():
                assert 0 <= f["rx"] < R and f["ch"] == 0
                hms[f["rx"]].frame(f)
    g.close()
    assert gd[0] > 0
    if gpu_decode:
        strip = lambda ls: [",".join(f for i, f in enumerate(l.split("*")[0].split(",")) if i != 3) for l in ls]
        for r in range(R):
            assert strip(hms[r].nmea()) == strip(oracles[r].nmea()) and len(oracles[r].nmea()) >= 3, "receiver %d" % r
            hms[r].close()


@pytest.mark.parametrize("hook,block,fmt,R,rate", [(0, 1024 * 48, "cf32", 3, 96000), (4, 1024 * 48, "cf32", 3, 96000), (8, 1024 * 44, "cf32", 2, 96000), (2, 1024 * 48, "cf32", 2, 96000),
                                                  (0, 1024 * 24, "cu8", 2, 96000), (8, 1024 * 24, "cs16", 3, 96000), (4, 1024 * 1, "cf32", 2, 96000),
                                                  (4, 512 * 48, "cf32", 3, 48000), (0, 512 * 40, "cs16", 2, 48000), (2, 512 * 48, "cf32", 2, 48000),
                                                  (8, 2048 * 24, "cf32", 2, 192000), (0, 2048 * 16, "cu8", 3, 192000), (4, 2048 * 2, "cf32", 2, 192000)])
def test_channel_mode_x_wave_front_end(hook, block, fmt, R, rate, monkeypatch):
    """Round 6, late: at 96 kSPS the mode-X front end is k1x_wave -- one wave per span of 1,024-sample tiles, Downsample2CIC5 / droop /
    FilterCIC5 in registers with DPP halos, a warm-up tile of which only the last 128 samples are read (in front of a block: the library's
    look-back).  Small batches get one-tile spans; the test hook k1u_spw forces spans of 4 / 8 tiles (tile-to-tile shadow registers,
    a last span that is shorter: 44 and 48 tiles per block; a block of ONE tile), 2 forces the workgroup form it replaced.  CF32 read in
    place and CU8 / CS16 through the converted copy; 48 kHz samples are compared through hard bits, levels and ppm of every block.
    Where a block is an even number of windows the waves also finish their span with the spectral analysis (wave_fft_tail, the row's
    windows in pairs) instead of k2_fft_search_win behind the front end: ppm of every window."""
    if hook:
        monkeypatch.setenv("AISGPU_K1U_SPW", str(hook))
    nblocks = 5  # (rate 48000 / 192000 -- round 6, last: the same waves without a CIC5 stage / with two)
    per = 1 if fmt == "cf32" else 2
    conv = {"cu8": synth.to_cu8, "cs16": synth.to_cs16, "cf32": lambda v: v}[fmt]
    xs = [conv(synth.receiver_stream(block * nblocks, sample_rate=rate, receiver_id=610 + r, gap_slots=(1, 2), single_channel=True)) for r in range(R)]
    oracles = []
    for x in xs:
        o = checkers.Oracle(model=2, rate=rate, fmt=fmt, taps=True, mode_x=True)
        o.feed_blocks(x, block)
        oracles.append(o)
    g = gpu.AisGpu(sample_rate=rate, n_receivers=R, block_len=block, input_format=_FMT[fmt], mode_x=True)
    gd, wd = [0] * R, [0] * R
    for b in range(nblocks):
        for r in range(R):
            g.submit(r, xs[r][b * block * per:(b + 1) * block * per])
        g.run()
        g.sync_outputs()
        for s_ in range(g.out_count()):
            for r in range(R):
                out, o = g.fetch(r, 0, s_), oracles[r]
                n, W = out["n_groups"], out["n_windows"]
                for j in range(5):
                    assert np.array_equal(out["bits"][j], o.bits(0, j)[0][gd[r]:gd[r] + n]), "bits b%d s%d r%d j%d" % (b, s_, r, j)
                assert _feq(out["lvl"], o.bits(0, 0)[1][gd[r]:gd[r] + n]) and _feq(out["ppm"], o.tap_ppm(2)[wd[r]:wd[r] + W])
                gd[r] += n
                wd[r] += W
    g.close()
    assert gd[0] > 0


@pytest.mark.parametrize("rate,block,fmt", [(96000, 1024 * 48, "cf32"), (96000, 1024 * 24, "cu8"), (150000, 2048 * 30, "cf32"),
                                            (120000, 2048 * 24, "cs8"), (96000, 1024, "cf32")])
def test_lowest_rates(rate, block, fmt):
    """96 kSPS (no decimation in front of Rotate, Model.cpp:332-334) and rates resampled into the 192k bucket (one CIC5 stage
    behind the resampler, Model.cpp:323-329)."""
    nblocks = 6 if block > 1024 else 200
    x = synth.receiver_stream(block * nblocks, sample_rate=rate, receiver_id=56, gap_slots=(1, 2))
    x = {"cu8": synth.to_cu8, "cs8": synth.to_cs8, "cf32": lambda v: v}[fmt](x)
    _run_multi_sub(x, rate, block, nblocks, fmt=fmt)


@pytest.mark.parametrize("rate,fmt", [(300000, "cf32"), (350000, "cf32"), (300000, "cu8"), (300000, "cs16")])
def test_resampled_rates_into_the_384k_bucket(rate, fmt):
    """288k < rate < 384k: the resampler works on the converted input itself (no CIC5 stage in front, Model.cpp:295-301)."""
    block = 4096 * 24
    x = synth.receiver_stream(block * 5, sample_rate=rate, receiver_id=54, gap_slots=(1, 2))
    if fmt == "cu8":
        x = synth.to_cu8(x)
    elif fmt == "cs16":
        x = synth.to_cs16(x)
    _run_multi_sub(x, rate, block, 5, fmt=fmt)


@pytest.mark.parametrize("rate,fmt", [(10000000, "cf32"), (8000000, "cf32"), (10000000, "cu8")])
def test_resampled_rates_above_6144k(rate, fmt):
    """10 MSPS (Airspy R2) / 8 MSPS: bucket 12288k, FIVE CIC5 stages in front of the resampler (Model.cpp:166-172) -- two
    pre-decimation passes (1 + 4 stages) in rounds 3-4, ONE pass of five stages (k1_dpp<5, FMT, PRE>) since round 5 -- then Upsample -> DS2_2 -> DS2_1 -> FDC(-2.0)."""
    block = 512 * 256 * 6
    x = synth.receiver_stream(block * 4, sample_rate=rate, receiver_id=53, gap_slots=(1, 2))
    if fmt == "cu8":
        x = synth.to_cu8(x)
    _run_multi_sub(x, rate, block, 4, fmt=fmt)


def test_challenger_fm_branch_bits_and_nmea():
    """AIS::ModelChallenger at 1536 kSPS: FM-branch decisions (atan2f discriminator -> 37-tap FIR -> sign) and the
    end-to-end NMEA of the 20-decoder wiring == the checker (Model.cpp:601-678)."""
    from ais_catcher_amd import host
    block, nblocks = 131072, 8
    x = synth.receiver_stream(block * nblocks, receiver_id=60, gap_slots=(1, 2), type5_every=4)
    chk = checkers.Ref(model=4, taps=True) if checkers.have_ref() else checkers.Oracle(model=4, taps=True)
    chk.feed_blocks(x, block)
    L = block // 32
    g = gpu.AisGpu(n_receivers=1, block_len=block, model=gpu.MODEL_CHALLENGER)
    for b in range(nblocks):
        g.submit(0, x[b * block:(b + 1) * block])
        g.run()
        g.sync_outputs()
        for ch in range(2):
            out = g.fetch(0, ch)
            for j in range(5):
                f = chk.bits(ch, j, 1)[0]  # FM decoder j saw samples N = j (mod 5)
                N = np.arange(b * L, (b + 1) * L)
                sel = N[N % 5 == j]
                assert np.array_equal(out["fm_bits"][sel - b * L], (f[sel // 5] > 0).astype(np.uint8)), "fm blk %d ch %d j %d" % (b, ch, j)
    g.close()
    host.reset_sequence()
    m = host.ModelChallengerGPU(block_len=block)
    for b in range(nblocks):
        m.receive(x[b * block:(b + 1) * block])
    assert m.nmea() == chk.nmea() and len(chk.nmea()) >= 4
    m.close()


def test_config3_6msps_challenger_nmea():
    """BASELINE configs[2]: 6 MSPS input, deeper CIC5 ladder + resampler, ModelChallenger -- NMEA identical to the checker."""
    from ais_catcher_amd import host
    block, nblocks = 786432, 6
    x = synth.receiver_stream(block * nblocks, sample_rate=6000000, receiver_id=61, type5_every=5)
    chk = checkers.Ref(model=4, rate=6000000) if checkers.have_ref() else checkers.Oracle(model=4, rate=6000000)
    chk.feed_blocks(x, block)
    host.reset_sequence()
    m = host.ModelChallengerGPU(sample_rate=6000000, block_len=block)
    for b in range(nblocks):
        m.receive(x[b * block:(b + 1) * block])
    assert len(chk.nmea()) >= 5
    assert m.nmea() == chk.nmea()
    a, c = m.msg_meta(), chk.msg_meta()
    assert np.array_equal(a[1], c[1])
    m.close()


@pytest.mark.parametrize("env", [{}, {"AISGPU_SERIAL": "1"}])
def test_config3_stream_placements(env, monkeypatch):
    """The 6 MSPS ladder's kernels on their streams (resampler front end on the downstream stream with the previous flush's second
    half deferred behind it, the FM bits regrouped in front of PhaseSearch) and everything on one stream (`serial`): the messages are
    the checker's in both, also across the drain at the end of the stream (the deferred second half)."""
    from ais_catcher_amd import host
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    block, nblocks = 786432, 5
    x = synth.receiver_stream(block * nblocks, sample_rate=6000000, receiver_id=62, type5_every=5)
    chk = checkers.Ref(model=4, rate=6000000) if checkers.have_ref() else checkers.Oracle(model=4, rate=6000000)
    chk.feed_blocks(x, block)
    host.reset_sequence()
    m = host.ModelChallengerGPU(sample_rate=6000000, block_len=block)
    for b in range(nblocks):
        m.receive(x[b * block:(b + 1) * block])
    assert len(chk.nmea()) >= 4
    assert m.nmea() == chk.nmea()
    m.close()


def test_challenger_fm_branch_inside_the_fir_kernel():
    """ModelChallenger's FM branch inside the derotation / FIR kernel: the FM decisions of every sample and the NMEA of the twenty
    decoders (host and device) are the checker's, over eight blocks (the carried tail of 40 derotated samples)."""
    from ais_catcher_amd import host
    block, nblocks = 131072, 8
    x = synth.receiver_stream(block * nblocks, receiver_id=65, gap_slots=(1, 2), type5_every=4)
    chk = checkers.Ref(model=4, taps=True) if checkers.have_ref() else checkers.Oracle(model=4, taps=True)
    chk.feed_blocks(x, block)
    L = block // 32
    g = gpu.AisGpu(n_receivers=1, block_len=block, model=gpu.MODEL_CHALLENGER)
    for b in range(nblocks):
        g.submit(0, x[b * block:(b + 1) * block])
        g.run()
        g.sync_outputs()
        for ch in range(2):
            out = g.fetch(0, ch)
            for j in range(5):
                f = chk.bits(ch, j, 1)[0]
                N = np.arange(b * L, (b + 1) * L)
                sel = N[N % 5 == j]
                assert np.array_equal(out["fm_bits"][sel - b * L], (f[sel // 5] > 0).astype(np.uint8)), "fm blk %d ch %d j %d" % (b, ch, j)
    g.close()
    for dec in (False, True):
        host.reset_sequence()
        m = host.ModelChallengerGPU(block_len=block, gpu_decode=dec) if dec else host.ModelChallengerGPU(block_len=block)
        for b in range(nblocks):
            m.receive(x[b * block:(b + 1) * block])
        assert m.nmea() == chk.nmea() and len(chk.nmea()) >= 4
        m.close()


@pytest.mark.parametrize("rate,block,nblocks", [(6000000, 786432, 14), (2400000, 393216, 14), (250000, 49152, 20)])
def test_resampled_ladders_over_many_blocks(rate, block, nblocks):
    """The resampled ladders keep rings -- six pre-decimated input blocks, eight sets of resampler tables, two sets of everything behind
    the 48 kHz channels, the second half of a flush deferred behind the next flush's resampler front end: a run long enough for every
    ring to wrap more than once; hard bits, levels and ppm of every downstream block against the oracle."""
    x = synth.receiver_stream(block * nblocks, sample_rate=rate, receiver_id=66, gap_slots=(0, 2))
    _run_outputs_vs_oracle([x], rate, "cf32", block, nblocks)


@pytest.mark.parametrize("spw", [2, 4, 8])
@pytest.mark.parametrize("rate,block,nblocks", [(6000000, 786432, 6), (2400000, 393216, 6), (10000000, 786432, 5), (1000000, 512 * 32 * 3, 8),
                                                (150000, 512 * 4 * 6, 8), (6000000, 512 * 128 * 3, 8)])
def test_resampler_front_end_span_walk(spw, rate, block, nblocks, monkeypatch):
    """k1u_resample_frontend walking 2 / 4 / 8 consecutive spans per workgroup (register prefetch one span ahead, LDS-only barriers,
    the next span's samples staged over the later stages' buffers).  launch_k1u only chooses such walks for batches of ~200 receivers
    and more; the test hook "k1u_spw" forces them for three receivers -- also on flushes whose span count is not a multiple of the
    walk (blocks of three 512-sample windows = 12 spans: the walk falls back to 4), with two and with one stage behind the resampler."""
    monkeypatch.setenv("AISGPU_K1U_SPW", str(spw))
    monkeypatch.setenv("AISGPU_US_K1", "0")  # (round 6: by default the ladders with two stages behind the resampler end in k1_dpp<2, 5, false> instead)
    xs = [synth.receiver_stream(block * nblocks, sample_rate=rate, receiver_id=360 + r, gap_slots=(0, 2)) for r in range(3)]
    _run_outputs_vs_oracle(xs, rate, "cf32", block, nblocks)


@pytest.mark.parametrize("us_k1", ["1", "0"])
@pytest.mark.parametrize("rate,block,nblocks,fmt", [(6000000, 786432, 6, "cf32"), (2400000, 393216, 6, "cu8"), (10000000, 786432, 5, "cf32"), (1000000, 512 * 32 * 3, 8, "cs16"),
                                                    (5000000, 512 * 128, 14, "cf32"), (700000, 512 * 16 * 5, 9, "cf32")])
def test_resampled_ladder_tail_in_the_front_end_waves(us_k1, rate, block, nblocks, fmt, monkeypatch):
    """Round 6: Upsample -> DS2_2 -> DS2_1 -> FDC -> Rotate -> DS2_a/b -> FilterCIC5 (Model.cpp:163-189, DSP.cpp:192-212) as one-wave
    workgroups of the front-end kernel -- k1_dpp<2, 5, false>: a lane computes its four Upsample outputs from the pre-decimated
    stream (tables two tiles ahead, samples one tile ahead), the two stages run in registers, the spectral analysis rides at the end
    of the waves -- against the oracle, and the same configurations through the form of rounds 3-5 (k1u_resample_frontend +
    k2_fft_search_win, test hook us_k1 = 0).  Flushes that straddle input blocks (every one at these ratios), one-window flushes
    (5 MSPS with 65,536-sample blocks: 16 tiles, one span), the first flush's halo in front of the stream, integer input formats."""
    monkeypatch.setenv("AISGPU_US_K1", us_k1)
    conv = {"cu8": synth.to_cu8, "cs16": synth.to_cs16, "cf32": lambda v: v}[fmt]
    xs = [conv(synth.receiver_stream(block * nblocks, sample_rate=rate, receiver_id=370 + r, gap_slots=(0, 2))) for r in range(3)]
    _run_outputs_vs_oracle(xs, rate, fmt, block, nblocks)


def test_fft_bin_magnitude_matches_hypot_restatement():
    """The FFT-bin magnitude routine (double sqrt without the denormal rescaling) against the glibc-equivalent
    hypotf restatement on 16M inputs: random bit patterns over the exponent range IQ data can reach, IQ-like
    values, zeros and denormals."""
    import ctypes
    lib = gpu.load()
    rng = np.random.default_rng(99)
    n = 1 << 22
    parts = []
    # uniform random mantissas, exponents -60..+20, both signs
    e = rng.integers(-60, 21, size=(n, 2))
    m = rng.random((n, 2), dtype=np.float32) + 1.0
    sgn = rng.choice(np.float32([-1, 1]), size=(n, 2))
    parts.append((np.ldexp(m, e).astype(np.float32) * sgn))
    parts.append(rng.standard_normal((n, 2), dtype=np.float32))
    parts.append((rng.integers(-128, 128, size=(n, 2)).astype(np.float32) / 128.0) ** 2)
    sp = rng.standard_normal((n, 2), dtype=np.float32) * np.float32(1e-3)
    sp[::7, 0] = 0.0
    sp[::11, 1] = 0.0
    sp[::13] = 0.0
    sp[1::13] = np.float32(1e-42)  # denormal
    sp[2::13, 0] = np.float32(3e-39)
    parts.append(sp)
    for x in parts:
        x = np.ascontiguousarray(x, np.float32)
        bad = lib.aisgpu_selftest(0, 0, x.ctypes.data_as(ctypes.c_void_p), x.shape[0])
        assert bad == 0


@pytest.mark.parametrize("rate,dsk", [(288000, False), (576000, True), (1152000, True), (2304000, True)])
def test_decimate_by_3_ladders(rate, dsk):
    """DownsampleKFilter ladders (a15): 288k needs no option, the others `-go DSK on`; CF32 and CU8."""
    block = 24576 * (rate // 288000) * 2
    x = synth.receiver_stream(block * 3, sample_rate=rate, receiver_id=31, gap_slots=(1, 2))
    _run_gpu_vs_oracle([x], rate, "cf32", block, 3, dsk=dsk)
    _run_gpu_vs_oracle([synth.to_cu8(x)], rate, "cu8", block, 3, dsk=dsk)


@pytest.mark.parametrize("hook,rate,dsk,fmt,blk", [(4, 288000, False, "cf32", 24576 * 2), (8, 288000, False, "cs16", 24576 * 2), (8, 288000, False, "cf32", 24576 * 3),
                                                  (4, 1152000, True, "cf32", 24576 * 8), (2, 288000, False, "cf32", 24576 * 2), (4, 288000, False, "cu8", 24576)])
def test_decimate_by_3_wave_front_end(hook, rate, dsk, fmt, blk, monkeypatch):
    """Round 6, late: the decimate-by-3 tail without a resampler in front is k1k_wave -- one wave per span of 1,536-sample tiles, the
    26-tap DownsampleKFilter from a 48-sample window per lane out of LDS (swizzled by the DMA), Rotate, DS2_a / DS2_b and FilterCIC5 in
    registers with DPP halos (FilterCIC5's reach two lanes back), a warm-up tile of which only the last 128 samples are read.  Small
    batches get one-tile spans; the test hook k1u_spw forces spans of 4 / 8 tiles (tile-to-tile look-back in LDS and shadow registers;
    a block is a multiple of 16 tiles), 2 forces the workgroup form it replaced.  Two distinct receivers, every tap (the 48 kHz
    channels themselves), bits, levels, ppm; CF32 read in place, CS16 / CU8 through the converted copy, 1152 kSPS behind its
    pre-decimation pass.  Taps path (the front end alone) and default path (the spectral analysis at the end of the same waves)."""
    monkeypatch.setenv("AISGPU_K1U_SPW", str(hook))
    nb = 4
    conv = {"cu8": synth.to_cu8, "cs16": synth.to_cs16, "cf32": lambda v: v}[fmt]
    xs = [conv(synth.receiver_stream(blk * nb, sample_rate=rate, receiver_id=640 + r, gap_slots=(1, 2))) for r in range(2)]
    _run_gpu_vs_oracle(xs, rate, fmt, blk, nb, dsk=dsk)
    # the default path: the waves also finish their span with the spectral analysis of the windows they have written (wave_fft_tail)
    # instead of k2_fft_search_win behind the front end -- ppm of every window, bits, levels
    _run_outputs_vs_oracle(xs, rate, fmt, blk, nb, dsk=dsk)


@pytest.mark.parametrize("hook,fmt,blk", [(0, "cf32", 1024 * 16), (4, "cf32", 1024 * 24), (8, "cs16", 1024 * 16), (2, "cf32", 1024 * 16), (4, "cu8", 1024)])
def test_dual_channel_96k_wave_front_end(hook, fmt, blk, monkeypatch):
    """96 kSPS dual-channel input, the ladder's last bucket (convert >> ROT >> DS2_a/b >> FCIC5, Model.cpp:332-334): k1k_wave without the
    filter (round 6, last) -- 512-sample tiles, a lane's eight samples straight out of the swizzled tile.  Span lengths by the hook k1u_spw
    (0: by batch size; 4 / 8; 2: k1u_resample_frontend<0>, the workgroup form), a one-window block, CF32 in place and CS16 / CU8 through
    the converted copy; taps path (the 48 kHz channels themselves) and default path (spectral analysis at the end of the waves)."""
    if hook:
        monkeypatch.setenv("AISGPU_K1U_SPW", str(hook))
    rate, nb = 96000, 5
    conv = {"cu8": synth.to_cu8, "cs16": synth.to_cs16, "cf32": lambda v: v}[fmt]
    xs = [conv(synth.receiver_stream(blk * nb, sample_rate=rate, receiver_id=660 + r, gap_slots=(1, 2))) for r in range(2)]
    _run_gpu_vs_oracle(xs, rate, fmt, blk, nb)
    _run_outputs_vs_oracle(xs, rate, fmt, blk, nb)


@pytest.mark.parametrize("rate, fmt, block, nblocks", [(1536000, "cf32", 131072, 8), (1536000, "cu8", 131072 * 3, 3), (768000, "cs16", 65536, 8),
                                                       (2304000, "cf32", 196608, 6), (2400000, "cu8", 204800, 6), (192000, "cs8", 16384 * 2, 6)])
def test_moving_average_downsampler(rate, fmt, block, nblocks):
    """`-go MA on` (Model.cpp:122-126, DSP.cpp:60-82): integrate-and-dump to 96 kHz in front of Rotate (which then works on the
    downsampler's 8192-sample blocks); every tap, bit, level and ppm against the oracle (== the compiled reference with the key
    set, tests/test_oracle_vs_ref.py::test_moving_average_downsampler)."""
    x = synth.receiver_stream(block * nblocks, sample_rate=rate, receiver_id=63, gap_slots=(1, 2))
    data = {"cu8": synth.to_cu8, "cs8": synth.to_cs8, "cs16": synth.to_cs16, "cf32": lambda v: v}[fmt](x)
    _run_gpu_vs_oracle([data], rate, fmt, block, nblocks, ma=True)


@pytest.mark.parametrize("model", [2, 4])
def test_moving_average_downsampler_messages(model):
    """... and the NMEA lines in the reference's order (channel A / B alternate every 4096 samples at 48 kHz, as on the
    decimate-by-3 ladders), through the host mirror and -- where the reference is compiled -- through the reference-side binding
    with SetKey(KEY_SETTING_MA, "ON")."""
    from ais_catcher_amd import host
    rate, block, nblocks = 1536000, 786432, 4
    x = synth.receiver_stream(block * nblocks, sample_rate=rate, receiver_id=64, gap_slots=(0, 1), type5_every=4)
    chk = checkers.Ref(model=model, rate=rate, ma=True) if checkers.have_ref() else checkers.Oracle(model=model, rate=rate, ma=True)
    chk.feed_blocks(x, block)
    want = chk.nmea()
    assert len(want) >= 6 and len(set(l.split(",")[4] for l in want)) == 2
    host.reset_sequence()
    cls = {2: host.ModelDefaultGPU, 4: host.ModelChallengerGPU}[model]
    m = cls(sample_rate=rate, block_len=block, ma=True)
    for b in range(nblocks):
        m.receive(x[b * block:(b + 1) * block])
    assert m.nmea() == want
    if model == 2:  # ... and with the frame decoders on the device (frames sorted into the same 4096-sample slices)
        host.reset_sequence()
        md = cls(sample_rate=rate, block_len=block, ma=True, gpu_decode=True)
        for b in range(nblocks):
            md.receive(x[b * block:(b + 1) * block])
        assert md.nmea() == want
    if checkers.have_refgpu():
        r = checkers.RefGpu(model=12 if model == 2 else 14, rate=rate, ma=True)
        r.feed_blocks(x, block)
        assert r.nmea() == want


@pytest.mark.parametrize("rate,dsk,k,fmt", [(250000, False, 0, "cf32"), (240000, False, 0, "cu8"), (500000, True, 1, "cf32"),
                                            (1000000, True, 2, "cf32"), (2000000, True, 3, "cu8")])
def test_rates_resampled_into_a_decimate_by_3_bucket(rate, dsk, k, fmt):
    """Upsample in front of DownsampleKFilter (Model.cpp:213-219 etc.): 250k / 240k -> 288k; with `-go DSK on` 500k -> 576k,
    1 MSPS -> 1152k, 2 MSPS -> 2304k (the reference then prefers these to the next 2^k bucket)."""
    block = (24576 << k) * 2
    x = synth.receiver_stream(block * 5, sample_rate=rate, receiver_id=55, gap_slots=(1, 2))
    if fmt == "cu8":
        x = synth.to_cu8(x)
    _run_multi_sub(x, rate, block, 5, fmt=fmt, dsk=dsk)


@pytest.mark.parametrize("hook,rate,dsk,k,fmt", [(4, 250000, False, 0, "cf32"), (8, 240000, False, 0, "cu8"), (2, 250000, False, 0, "cf32"), (4, 1000000, True, 2, "cf32")])
def test_resampled_decimate_by_3_wave_front_end(hook, rate, dsk, k, fmt, monkeypatch):
    """Round 6, last: with Upsample in front of DownsampleKFilter the one-wave front end forms its tiles itself (k1k_wave<true, ., true>): the
    input span of a tile by global_load_lds out of the ring of three input blocks, the interpolation in the lanes from the host's (index,
    alpha) tables.  Span lengths by the hook k1u_spw (4 / 8; 2: k1k_dsk_frontend, the workgroup form), flushes that begin in the previous
    input block, CU8 through the converted ring; taps path (48 kHz channels) and default path (spectral analysis at the end of the waves)."""
    monkeypatch.setenv("AISGPU_K1U_SPW", str(hook))
    block = (24576 << k) * 2
    x = synth.receiver_stream(block * 5, sample_rate=rate, receiver_id=57, gap_slots=(1, 2))
    if fmt == "cu8":
        x = synth.to_cu8(x)
    _run_multi_sub(x, rate, block, 5, fmt=fmt, dsk=dsk)
    _run_outputs_vs_oracle([x], rate, fmt, block, 5, dsk=dsk)


def test_resampled_decimate_by_3_ladder_message_order():
    """250 kSPS end to end through the host model: Rotate still works on the filter's 8192-sample blocks, so the channels
    alternate every 4096 samples at 48 kHz inside a downstream block."""
    from ais_catcher_amd import host
    rate, block, nblocks = 250000, 49152, 16
    x = synth.receiver_stream(block * nblocks, sample_rate=rate, receiver_id=48, gap_slots=(0, 1), type5_every=4)
    chk = checkers.Ref(model=2, rate=rate) if checkers.have_ref() else checkers.Oracle(model=2, rate=rate)
    chk.feed_blocks(x, block)
    host.reset_sequence()
    m = host.ModelDefaultGPU(sample_rate=rate, block_len=block)
    for b in range(nblocks):
        m.receive(x[b * block:(b + 1) * block])
    assert m.nmea() == chk.nmea() and len(chk.nmea()) >= 6
    assert len(set(l.split(",")[4] for l in chk.nmea())) == 2


def test_phase_search_boxcar():
    """`-go PS_EMA off` (a9'): Demod::PhaseSearch with its 12-symbol boxcar instead of PhaseSearchEMA; three block sizes so
    that the ring slot and the decision history cross block boundaries at different phases."""
    for block, nb, rid in ((131072, 6, 41), (16384, 20, 42), (786432, 2, 43), (393216, 5, 44)): # (the last two: chunk-parallel, ring slot 0 / 7 and 0 / 9 / 6 / 3 / 0 at the block starts)
        x = synth.receiver_stream(block * nb, receiver_id=rid, gap_slots=(1, 2))
        _run_gpu_vs_oracle([x, x[::-1].copy()], 1536000, "cf32", block, nb, ps_ema=False)


def test_model_base_nmea_end_to_end():
    """AIS::ModelBase (-m 1, a14): GPU front end + FM discriminator + 37-tap filter, host SimplePLL/decoder loop; the
    discriminator signs and the NMEA text against the checker."""
    from ais_catcher_amd import host
    block, nblocks = 131072, 12
    x = synth.receiver_stream(block * nblocks, receiver_id=61, gap_slots=(1, 2), type5_every=4)
    chk = checkers.Ref(model=1, taps=True) if checkers.have_ref() else checkers.Oracle(model=1, taps=True)
    chk.feed_blocks(x, block)
    # (1) the device output: sign of the filtered discriminator for every 48 kHz sample
    g = gpu.AisGpu(block_len=block, model=gpu.MODEL_BASE)
    L = block // 32
    for b in range(nblocks):
        g.submit(0, x[b * block:(b + 1) * block])
        g.run()
        g.sync_outputs()
        for ch in range(2):
            o = g.fetch(0, ch)
            assert o["n_groups"] == 0
            ref = chk.bits(ch, 0, 1)[0][b * L:(b + 1) * L] > 0
            assert np.array_equal(o["fm_bits"].astype(bool), ref), "block %d ch %d" % (b, ch)
    g.close()
    # (2) end to end through the C++ host model
    host.reset_sequence()
    m = host.ModelBaseGPU(block_len=block)
    for b in range(nblocks):
        m.receive(x[b * block:(b + 1) * block])
    assert m.nmea() == chk.nmea() and len(chk.nmea()) >= 3
    a, c = m.msg_meta(), chk.msg_meta()
    assert np.array_equal(a[0], c[0]) and np.array_equal(a[1], c[1])


@pytest.mark.parametrize("model", [2, 4, 0])
def test_decimate_by_3_ladder_message_order(model):
    """288 kSPS: DownsampleKFilter hands Rotate 8192 samples at a time, so the reference alternates between the channels every
    4096 samples at 48 kHz inside one input block; the host replay must follow, or the messages come out in another order."""
    from ais_catcher_amd import host
    rate, block, nblocks = 288000, 49152, 14
    x = synth.receiver_stream(block * nblocks, sample_rate=rate, receiver_id=47, gap_slots=(0, 1), type5_every=4)
    chk = checkers.Ref(model=model, rate=rate) if checkers.have_ref() else checkers.Oracle(model=model, rate=rate)
    chk.feed_blocks(x, block)
    host.reset_sequence()
    cls = {2: host.ModelDefaultGPU, 4: host.ModelChallengerGPU, 0: host.ModelStandardGPU}[model]
    m = cls(sample_rate=rate, block_len=block)
    for b in range(nblocks):
        m.receive(x[b * block:(b + 1) * block])
    assert m.nmea() == chk.nmea() and len(chk.nmea()) >= 6
    assert len(set(l.split(",")[4] for l in chk.nmea())) == 2  # both channels carry messages


@pytest.mark.parametrize("rate,fmt,block,nblocks", [(1536000, "cf32", 131072, 24), (1536000, "cu8", 786432, 4), (768000, "cf32", 65536, 24),
                                                    (6000000, "cf32", 786432, 4), (288000, "cf32", 49152, 12)])
def test_model_engine_v2_end_to_end(rate, fmt, block, nblocks):
    """AIS::ModelEngineV2 (-m 11): the device runs the front end and hands over the two 48 kHz channels (bit-exact against the
    checker's taps); V2::Engine -- whose every 512-sample block depends on the state of its own decoders -- runs on the host."""
    from ais_catcher_amd import host
    x = synth.receiver_stream(block * nblocks, sample_rate=rate, receiver_id=46, gap_slots=(1, 2), type5_every=4)
    data = synth.to_cu8(x) if fmt == "cu8" else x
    per = 1 if fmt == "cf32" else 2
    chk = checkers.Ref(model=11, rate=rate, fmt=fmt, taps=True) if checkers.have_ref() else checkers.Oracle(model=11, rate=rate, fmt=fmt, taps=True)
    chk.feed_blocks(data, block)
    # (1) the device output
    g = gpu.AisGpu(sample_rate=rate, block_len=block, input_format=_FMT[fmt], model=gpu.MODEL_V2)
    taps = [chk.tap(0), chk.tap(1)]
    done = [0, 0]
    for b in range(nblocks):
        g.submit(0, data[b * block * per:(b + 1) * block * per])
        g.run()
        g.sync_outputs()
        for sub in range(g.out_count()):
            for ch in range(2):
                o = g.fetch(0, ch, sub)
                n = len(o["c48"])
                assert o["n_groups"] == 0 and n == 512 * o["n_windows"] and o["first_sample48"] == done[ch]
                assert _feq(o["c48"], taps[ch][done[ch]:done[ch] + n]), "block %d ch %d" % (b, ch)
                done[ch] += n
    g.close()
    assert done[0] > 0 and done[0] == done[1]
    # (2) end to end through the C++ host model
    host.reset_sequence()
    m = host.ModelEngineV2GPU(sample_rate=rate, block_len=block, input_format=_FMT[fmt])
    for b in range(nblocks):
        m.receive(data[b * block * per:(b + 1) * block * per])
    assert m.nmea() == chk.nmea() and len(chk.nmea()) >= 3
    a, c = m.msg_meta(), chk.msg_meta()
    assert np.array_equal(a[0], c[0]) and np.array_equal(a[1], c[1])


@pytest.mark.parametrize("roles", ["1", "2", "0"])
@pytest.mark.parametrize("rate,fmt,block,nblocks", [(1536000, "cf32", 131072, 24), (1536000, "cu8", 786432, 4), (768000, "cf32", 65536, 24), (6000000, "cf32", 786432, 4),
                                                    (1536000, "cf32", 16384, 192), (1536000, "cf32", 49152, 64)])  # (one and three engine blocks per call)
def test_model_engine_v2_on_the_device(rate, fmt, block, nblocks, roles, monkeypatch):
    """AISGPU_FLAG_GPU_DECODE with ModelEngineV2 (round 4): the engine's coherent branch runs on the device too
    (kv2_engine: tone gate / slot lock from the decoders' states, Derotate, FilterFL17, five PhaseTrackers, six decoders with
    their reset, the slot-phase learner) -- the 48 kHz channels never leave the device, completed frames come back.  Everything
    is the reference's arithmetic in the reference's order, std::polar of the estimated frequency included (glibc's sinf / cosf
    restated, tests/test_sincosf.py): NMEA text, tag.ppm and the per-message level -- a sum of |derotated, filtered sample|^2
    over the frame -- must equal the compiled reference's bit for bit.
    roles = 1 (round 6, the default up to 512 channels): trackers, FM decoder and the next block's front end on three waves of a workgroup,
    speculating that no message completes, with the exact order restored where one does; 2: the same kernel compiled for three waves per
    SIMD (bigger batches); 0: round 5's one-wave form (test hook v2_roles)."""
    from ais_catcher_amd import host
    monkeypatch.setenv("AISGPU_V2_ROLES", roles)
    x = synth.receiver_stream(block * nblocks, sample_rate=rate, receiver_id=46, gap_slots=(1, 2), type5_every=4)
    data = synth.to_cu8(x) if fmt == "cu8" else x
    per = 1 if fmt == "cf32" else 2
    chk = checkers.Ref(model=11, rate=rate, fmt=fmt) if checkers.have_ref() else checkers.Oracle(model=11, rate=rate, fmt=fmt)
    chk.feed_blocks(data, block)
    host.reset_sequence()
    m = host.ModelEngineV2GPU(sample_rate=rate, block_len=block, input_format=_FMT[fmt], gpu_decode=True)
    for b in range(nblocks):
        assert m.receive(data[b * block * per:(b + 1) * block * per]) == 0
    assert len(chk.nmea()) >= 3
    assert m.nmea() == chk.nmea()
    a, c = m.msg_meta(), chk.msg_meta()
    assert np.array_equal(a[1], c[1]), "tag.ppm"
    assert np.array_equal(a[0], c[0]), "tag.level"
    m.close()


@pytest.mark.parametrize("roles", ["1", "0"])
def test_model_engine_v2_on_the_device_batch_of_distinct_receivers(roles, monkeypatch):
    """Seven distinct receivers = fourteen channels: one full wave of the engine kernel (ten channels) and a partial one, one host
    thread per receiver on a shared batch; every receiver's NMEA text, levels and ppm against the compiled reference, in order.
    The streams put their bursts on the SOTDMA slot grid, so the engines learn the slot phase and take the kernel's slow path
    (Estimate() at an arbitrary offset inside the block) as well: counted."""
    import threading
    from ais_catcher_amd import host
    monkeypatch.setenv("AISGPU_V2_ROLES", roles)
    R, block, nblocks = 7, 131072, 30
    xs = [synth.receiver_stream(block * nblocks, receiver_id=400 + r, gap_slots=(0, 1), type5_every=5) for r in range(R)]
    want = []
    for x in xs:
        chk = checkers.Ref(model=11) if checkers.have_ref() else checkers.Oracle(model=11)
        chk.feed_blocks(x, block)
        want.append((chk.nmea(), chk.msg_meta()))
        chk.close()
    host.reset_sequence()
    batch = host.Batch(n_receivers=R, block_len=block, model=gpu.MODEL_V2, gpu_decode=True)
    ms = [host.ModelEngineV2GPU(block_len=block, batch=batch, rx=r, gpu_decode=True) for r in range(R)]

    def work(r):
        for b in range(nblocks):
            ms[r].receive(xs[r][b * block:(b + 1) * block])
    th = [threading.Thread(target=work, args=(r,)) for r in range(R)]
    [t.start() for t in th]
    [t.join() for t in th]
    for r in range(R):
        # (the multi-part sentence id is one counter per process: compare everything but that field and the checksum over it)
        strip = lambda ls: [",".join(f for i, f in enumerate(l.split("*")[0].split(",")) if i != 3) for l in ls]
        assert strip(ms[r].nmea()) == strip(want[r][0]) and len(want[r][0]) >= 3, "rx %d" % r
        a = ms[r].msg_meta()
        assert np.array_equal(a[0], want[r][1][0]) and np.array_equal(a[1], want[r][1][1]), "rx %d: level / ppm" % r
        ms[r].close()
    batch.close()
    # the slow path on its own context: a learned slot phase asks for estimates at arbitrary offsets
    g = gpu.AisGpu(n_receivers=2, block_len=block, model=gpu.MODEL_V2, gpu_decode=True)
    nf = 0
    for b in range(nblocks):
        for r in range(2):
            g.submit(r, xs[r][b * block:(b + 1) * block])
        g.run()
        g.sync_outputs()
        nf += len(g.frames())
    assert nf >= 6 and g.decoder_fallbacks() > 0
    g.close()


def test_model_engine_v2_three_waves_equal_one_wave_on_a_batch(monkeypatch):
    """Round 6: kv2_engine_roles (trackers / FM decoder / the next block's front end for both values of busy on three waves, every
    block speculating that nobody completes a message) against round 5's one-wave kernel -- which the tests above pin to the compiled
    reference -- on 48 distinct noisy receivers of the bench's workload over eighteen blocks of 15 engine blocks each: every frame record
    (decoder, position, level sum, start / end index, bits), 96 channels' worth, must be identical, and so must the number of
    Estimate() calls at a learned slot phase (the preparation of a block is voided and redone wherever a tracker's decoder completes
    a message, and at the learned slot phase only one answer exists)."""
    import torch
    from ais_catcher_amd import workload
    R, block, nblocks = 48, 245760, 3  # (a whole number of SOTDMA slots per block: the engines learn the slot phase over the passes)
    data = workload.resident_batch(torch, R, nblocks, seed=3, block=block)
    out = []
    for roles in ("1", "2", "0"):
        monkeypatch.setenv("AISGPU_V2_ROLES", roles)
        g = gpu.AisGpu(n_receivers=R, block_len=block, model=gpu.MODEL_V2, gpu_decode=True)
        frames = []
        for b in range(6 * nblocks):
            g.submit_device(data[b % nblocks].data_ptr(), block)
            g.run()
            g.sync_outputs()
            frames += g.frames()
        # (the record carries the whole frame buffer; bits at and beyond `position` are whatever earlier frames left there)
        bits = lambda f: (f["data"][:f["position"] // 8], f["data"][f["position"] // 8] & ((1 << (f["position"] % 8)) - 1))
        out.append((sorted((f["rx"], f["ch"], f["end_idx"], f["phase"], f["start_idx"], f["position"], f["level_sum"]) + bits(f) for f in frames), g.decoder_fallbacks()))
        g.close()
    assert len(out[0][0]) >= 200 and out[0][0] == out[2][0] and out[1][0] == out[2][0]
    assert out[0][1] == out[2][1] and out[1][1] == out[2][1] and out[0][1] > 0


@pytest.mark.parametrize("rate,fmt,block", [(1536000, "cf32", 131072), (1536000, "cu8", 131072), (768000, "cf32", 65536)])
def test_model_standard_nmea_end_to_end(rate, fmt, block):
    """AIS::ModelStandard (-m 0): the device path of ModelBase (front end + FM discriminator + 37-tap filter), then on the host
    Deinterleave(5) and five decoders with their Reset mesh; NMEA text, levels and ppm against the checker."""
    from ais_catcher_amd import host
    nblocks = 10
    x = synth.receiver_stream(block * nblocks, sample_rate=rate, receiver_id=31, gap_slots=(1, 2), type5_every=4)
    data = synth.to_cu8(x) if fmt == "cu8" else x
    per = 1 if fmt == "cf32" else 2
    chk = checkers.Ref(model=0, rate=rate, fmt=fmt) if checkers.have_ref() else checkers.Oracle(model=0, rate=rate, fmt=fmt)
    chk.feed_blocks(data, block)
    host.reset_sequence()
    m = host.ModelStandardGPU(sample_rate=rate, block_len=block, input_format=_FMT[fmt])
    for b in range(nblocks):
        m.receive(data[b * block * per:(b + 1) * block * per])
    assert m.nmea() == chk.nmea() and len(chk.nmea()) >= 3
    a, c = m.msg_meta(), chk.msg_meta()
    assert np.array_equal(a[0], c[0]) and np.array_equal(a[1], c[1])


@pytest.mark.parametrize("block,nblocks,fmt", [(786432, 4, "cf32"), (16384, 96, "cf32"), (131072, 16, "cu8")])
def test_gpu_frame_decoder_nmea(block, nblocks, fmt):
    """AISGPU_FLAG_GPU_DECODE: the ten AIS::Decoder state machines (NRZI, flags, de-stuffing, CRC, early aborts, Reset mesh)
    run on the device; the host only formats the frames that come back.  NMEA text, levels and ppm against the checker,
    with block sizes that cut frames at every possible place."""
    from ais_catcher_amd import host
    x = synth.receiver_stream(block * nblocks, receiver_id=71, type5_every=3, gap_slots=(0, 1))
    data = synth.to_cu8(x) if fmt == "cu8" else x
    per = 1 if fmt == "cf32" else 2
    chk = checkers.Ref(fmt=fmt) if checkers.have_ref() else checkers.Oracle(fmt=fmt)
    chk.feed_blocks(data, block)
    host.reset_sequence()
    m = host.ModelDefaultGPU(block_len=block, input_format=_FMT[fmt], gpu_decode=True)
    for b in range(nblocks):
        m.receive(data[b * block * per:(b + 1) * block * per])
    assert len(chk.nmea()) >= 8
    assert m.nmea() == chk.nmea()
    a, c = m.msg_meta(), chk.msg_meta()
    assert np.array_equal(a[0], c[0]) and np.array_equal(a[1], c[1])


@pytest.mark.parametrize("mode", ["event", "seq"])
def test_gpu_frame_decoder_block_length_sweep(mode, monkeypatch):
    """The device decoders over block lengths of 1 .. 47 windows (odd group counts, partial last words of the packed hard bits,
    frames cut at every place): NMEA text and levels == the checker, for the event-driven kernels and the sequential one."""
    from ais_catcher_amd import host
    if mode == "seq":
        monkeypatch.setenv("AISGPU_K7", "seq")
    for windows in (1, 2, 3, 5, 7, 11, 14, 19, 24, 29, 39, 47):
        block = 32 * 512 * windows
        nblocks = max(3, 786432 // block)
        x = synth.receiver_stream(block * nblocks, receiver_id=300 + windows, type5_every=3, gap_slots=(0, 1))
        chk = checkers.Ref() if checkers.have_ref() else checkers.Oracle()
        chk.feed_blocks(x, block)
        host.reset_sequence()
        m = host.ModelDefaultGPU(block_len=block, gpu_decode=True)
        for b in range(nblocks):
            m.receive(x[b * block:(b + 1) * block])
        assert m.nmea() == chk.nmea(), "windows %d" % windows
        a, c = m.msg_meta(), chk.msg_meta()
        assert np.array_equal(a[0], c[0]) and np.array_equal(a[1], c[1]), "windows %d" % windows
        m.close()


@pytest.mark.parametrize("model", [4, 1, 0])
def test_other_models_block_length_sweep(model):
    """ModelChallenger (PhaseSearch + FM branch, 20 decoders), ModelBase (SimplePLL with decoder feedback) and ModelStandard end
    to end over a handful of block lengths."""
    from ais_catcher_amd import host
    cls = {4: host.ModelChallengerGPU, 1: host.ModelBaseGPU, 0: host.ModelStandardGPU}[model]
    for windows in (2, 5, 14, 19, 24, 39):
        block = 32 * 512 * windows
        nblocks = max(3, 524288 // block)
        x = synth.receiver_stream(block * nblocks, receiver_id=400 + windows, gap_slots=(0, 1), type5_every=4)
        chk = checkers.Ref(model=model) if checkers.have_ref() else checkers.Oracle(model=model)
        chk.feed_blocks(x, block)
        host.reset_sequence()
        m = cls(block_len=block)
        for b in range(nblocks):
            m.receive(x[b * block:(b + 1) * block])
        assert m.nmea() == chk.nmea(), "model %d windows %d" % (model, windows)
        m.close()


@pytest.mark.parametrize("mode", ["seq", "alt"])
def test_gpu_frame_decoder_implementations_share_their_state(mode, monkeypatch):
    """AISGPU_K7=seq: the symbol-by-symbol kernel; alt: event-driven and sequential kernels take turns block by block on the same
    DecState (training counts, start flags and frames in flight cross the block boundaries in both directions)."""
    monkeypatch.setenv("AISGPU_K7", mode)
    test_gpu_frame_decoder_nmea(16384, 96, "cf32")
    test_gpu_frame_decoder_nmea(131072, 16, "cu8")
    test_gpu_frame_decoder_nmea(786432, 4, "cf32")


def test_gpu_frame_decoder_matches_host_decoders_on_a_batch():
    """8 receivers, noisy weak signals (many false trainings, aborted and CRC-failing frames): device decoders == host decoders."""
    from ais_catcher_amd import host
    block, nblocks, R = 131072, 12, 8
    streams = [synth.receiver_stream(block * nblocks, receiver_id=80 + r, gap_slots=(0, 2), noise_sigma=0.05 + 0.03 * r) for r in range(R)]
    out = []
    for dec in (False, True):
        host.reset_sequence()
        batch = host.Batch(n_receivers=R, block_len=block, gpu_decode=dec)
        models = [host.ModelDefaultGPU(block_len=block, batch=batch, rx=r) for r in range(R)]
        import threading
        def work(r):
            for b in range(nblocks):
                models[r].receive(streams[r][b * block:(b + 1) * block])
        th = [threading.Thread(target=work, args=(r,)) for r in range(R)]
        [t.start() for t in th]
        [t.join() for t in th]
        out.append([mm.nmea() for mm in models])  # per receiver, in order
    assert out[0] == out[1]
    assert sum(len(o) for o in out[0]) >= 8


# ---------------------------------------------------------------------------------------------------------------
# The default (fused) back end: checkpointed phasor recurrence + derotation/FIR/ScatterPLL in one kernel.  It has no float
# taps (nothing is materialised), so it is compared on everything it delivers: hard bits, levels, ppm, per block.
# ---------------------------------------------------------------------------------------------------------------
def _run_outputs_vs_oracle(streams, rate, fmt, block, nblocks, **kw):
    R = len(streams)
    okw = dict(dsk=kw.get("dsk", False), ps_ema=kw.get("ps_ema", True), fp_ds=kw.get("fp_ds", False), ma=kw.get("ma", False),
               afc_wide=kw.get("afc_wide", True), droop=kw.get("droop", True))
    g = gpu.AisGpu(sample_rate=rate, n_receivers=R, block_len=block,
                   input_format=_FMT[fmt], taps=False, **kw)
    per = 1 if fmt == "cf32" else 2
    oracles = [checkers.Oracle(model=2, rate=rate, fmt=fmt, taps=True, **okw) for _ in range(R)]
    for o, x in zip(oracles, streams):
        o.feed_blocks(x, block)
    oppm = [[o.tap_ppm(2), o.tap_ppm(3)] for o in oracles]
    obits = [[[o.bits(ch, j) for j in range(5)] for ch in range(2)] for o in oracles]
    gd = [[0, 0] for _ in range(R)]
    wd = [[0, 0] for _ in range(R)]
    for b in range(nblocks):
        for r in range(R):
            g.submit(r, streams[r][b * block * per:(b + 1) * block * per])
        g.run()
        g.sync_outputs()
        for s in range(g.out_count()):
            for r in range(R):
                for ch in range(2):
                    out = g.fetch(r, ch, s)
                    n, g0, W = out["n_groups"], gd[r][ch], out["n_windows"]
                    assert out["first_group"] == g0
                    for j in range(5):
                        assert np.array_equal(out["bits"][j], obits[r][ch][j][0][g0:g0 + n]), "bits b%d s%d r%d c%d j%d" % (b, s, r, ch, j)
                    assert _feq(out["lvl"], obits[r][ch][0][1][g0:g0 + n]), "lvl b%d s%d r%d c%d" % (b, s, r, ch)
                    assert _feq(out["ppm"], oppm[r][ch][wd[r][ch]:wd[r][ch] + W]), "ppm b%d s%d r%d c%d" % (b, s, r, ch)
                    gd[r][ch] += n
                    wd[r][ch] += W
    assert gd[0][0] > 0
    g.close()


@pytest.mark.parametrize("block,nblocks", [(786432, 3), (16384, 30), (65536, 9)])
def test_fused_backend_1536k(block, nblocks):
    xs = [synth.receiver_stream(block * nblocks, receiver_id=90 + r, gap_slots=(0, 2)) for r in range(3)]
    _run_outputs_vs_oracle(xs, 1536000, "cf32", block, nblocks)
    _run_outputs_vs_oracle([synth.to_cu8(xs[0])], 1536000, "cu8", block, nblocks)


@pytest.mark.parametrize("env", [{"AISGPU_K46": "1"}, {"AISGPU_K46": "1", "AISGPU_PS_WARM": "16"}, {}, {"AISGPU_PS_WARM": "16"}, {"AISGPU_K46": "1", "AISGPU_SERIAL": "1"}])
@pytest.mark.parametrize("R,block,nblocks", [(1, 786432, 6), (5, 786432, 3), (2, 491520, 4)])
def test_fir_and_phase_search_in_one_workgroup(env, R, block, nblocks, monkeypatch):
    """Round 5's fused back end (option k46 = 1; measured slower than the two kernels, so not the default -- profiles/r05_expA_k46.txt):
    derotation + FIR + ScatterPLL + PhaseSearchEMA in one workgroup (k46_window_search: four waves = the five sampling phases of
    three adjacent channels, the FIR outputs only in LDS).  Channel counts that leave the last
    workgroup with one or two channels (2, 10 and 4 channels), blocks whose first group begins in the previous block (24,576 samples
    = 4,915.2 groups per block: every alignment within five blocks), a block of exactly three chunks (15,360 samples = 3,072 groups);
    with a 16-symbol warm-up every speculative chunk fails its check and the assembling wave materialises the rows of its
    channels and searches sequentially; without the option the two-kernel default runs on the same inputs.
    Hard bits, levels and ppm of every block against the oracle."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    xs = [synth.receiver_stream(block * nblocks, receiver_id=330 + r, gap_slots=(0, 2)) for r in range(R)]
    _run_outputs_vs_oracle(xs, 1536000, "cf32", block, nblocks)


@pytest.mark.parametrize("rate,kw", [(192000, {}), (768000, {}), (3072000, {}), (12288000, {}), (6000000, {}), (2400000, {}),
                                     (288000, {}), (1152000, {"dsk": True})])
def test_fused_backend_other_ladders(rate, kw):
    if rate in (288000, 1152000):
        block = 24576 * (rate // 288000) * 2
    elif rate in (6000000, 2400000):
        block = 786432 if rate == 6000000 else 393216
    else:
        block = 512 * (rate // 48000) * 8
    x = synth.receiver_stream(block * 3, sample_rate=rate, receiver_id=95, gap_slots=(1, 2))
    _run_outputs_vs_oracle([x], rate, "cf32", block, 3, **kw)


def test_fused_backend_edge_inputs_and_boxcar():
    n = 16384 * 5
    rng = np.random.default_rng(3)
    zero = np.zeros(n, np.complex64)
    dc = np.full(n, 0.25 - 0.5j, np.complex64)
    wild = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    wild[:n // 2] *= np.float32(1e-18)
    wild[n // 2:] *= np.float32(3e4)
    _run_outputs_vs_oracle([zero, dc, wild], 1536000, "cf32", 16384, 5)
    x = synth.receiver_stream(131072 * 4, receiver_id=96)
    _run_outputs_vs_oracle([x], 1536000, "cf32", 131072, 4, ps_ema=False)


@pytest.mark.parametrize("tps,env", [(16, None), (32, None), (48, None), (0, "0"), (24, None)])
def test_spectral_analysis_at_the_end_of_the_front_end_waves(tps, env, monkeypatch):
    """The FFT of SquareFreqOffsetCorrection rides in the front-end kernel when a span is a whole number of 512-sample windows
    (16 tiles): one, two or three windows per channel and span; AISGPU_FFT_IN_K1=0 (option fft_in_k1) / spans of 24 tiles use
    the FFT and search kernels.
    ppm (i.e. every window's peak search), hard bits and levels must not change by a bit, on pure and pre-decimated ladders
    and on integer input."""
    if env is not None:
        monkeypatch.setenv("AISGPU_FFT_IN_K1", env)
    kw = {"tiles_per_span": tps} if tps else {}
    xs = [synth.receiver_stream(98304 * 4, receiver_id=120 + r, gap_slots=(0, 2)) for r in range(3)]
    _run_outputs_vs_oracle(xs, 1536000, "cf32", 98304, 4, **kw)       # 96 tiles per block
    _run_outputs_vs_oracle([synth.to_cu8(xs[1])], 1536000, "cu8", 98304, 4, **kw)
    # 3072 kSPS: five stages in the front-end waves (tiles of 2,048 samples, 16 KB by LDS-DMA for CF32), the same tail behind them
    x5 = synth.receiver_stream(196608 * 3, sample_rate=3072000, receiver_id=125, gap_slots=(0, 2))
    _run_outputs_vs_oracle([x5, x5[::-1].copy()], 3072000, "cf32", 196608, 3, **kw)  # 96 tiles per block
    _run_outputs_vs_oracle([synth.to_cs16(x5)], 3072000, "cs16", 196608, 3, **kw)
    if tps in (16, 0):
        x = synth.receiver_stream(786432 * 2, sample_rate=6144000, receiver_id=123, gap_slots=(1, 2))
        _run_outputs_vs_oracle([x], 6144000, "cf32", 786432, 2, **kw)  # pre-decimation pass in front: 192 tiles of the second pass
        x = synth.receiver_stream(24576 * 3, sample_rate=384000, receiver_id=124, gap_slots=(1, 2))
        _run_outputs_vs_oracle([x], 384000, "cf32", 24576 * 2, 1, **kw)  # two stages only: 96 tiles


@pytest.mark.parametrize("windows", [14, 19, 24, 39])
def test_phase_search_chunk_whose_last_word_ends_in_a_partial_batch(windows):
    """Group counts whose last chunk has 25 .. 31 symbols in its last word (409, 410, 921, 922 ...): the chunk kernel's last,
    partial batch of 8 ends on a word boundary.  (Such a word used to be flushed inside the loop and then overwritten by an
    empty one; found with the 300 kSPS ladder, whose blocks of 12288 samples at 48 kHz have 2457 groups.)"""
    block = 32 * 512 * windows
    xs = [synth.receiver_stream(block * 3, receiver_id=130 + r, gap_slots=(0, 1)) for r in range(2)]
    _run_outputs_vs_oracle(xs, 1536000, "cf32", block, 3)
    _run_gpu_vs_oracle(xs[:1], 1536000, "cf32", block, 2)  # materialised back end (taps)


@pytest.mark.parametrize("fmt", ["cf32", "cu8"])
def test_block_length_sweep(fmt):
    """Every block length of 1 .. 48 windows of 512 samples at 48 kHz (the reference accepts any multiple of 16384 input
    samples at 1536 kSPS): chunk tails, segment tails, span counts and ring phases all change with it; hard bits, levels and
    ppm must not."""
    for windows in range(1, 49):
        block = 32 * 512 * windows
        nblocks = 3 if windows < 24 else 2
        x = synth.receiver_stream(block * nblocks, receiver_id=200 + windows, gap_slots=(0, 1))
        if fmt == "cu8":
            x = synth.to_cu8(x)
        _run_outputs_vs_oracle([x], 1536000, fmt, block, nblocks)


def test_materialised_backend_without_taps(monkeypatch):
    """AISGPU_FUSED=0 keeps the phasor / derotated-sample arrays (the path the taps and the FM branch use) with the deferred
    second half; same outputs."""
    monkeypatch.setenv("AISGPU_FUSED", "0")
    xs = [synth.receiver_stream(65536 * 9, receiver_id=97 + r, gap_slots=(0, 2)) for r in range(2)]
    _run_outputs_vs_oracle(xs, 1536000, "cf32", 65536, 9)


@pytest.mark.parametrize("model,rate,fmt,block", [(1, 1536000, "cf32", 131072), (0, 768000, "cu8", 65536), (4, 1536000, "cf32", 131072),
                                                  (4, 1536000, "cf32", 786432), (1, 288000, "cf32", 49152)])
def test_fm_receiver_float_taps(model, rate, fmt, block):
    """a11 / a12 as FLOATS (north star: intermediate samples within 1e-5 rel; here 0 ulp): the device's Demod::FM output (atan2f
    restated, taps 6/7) and its Filter(Receiver) output (37 taps, taps 8/9) of every 48 kHz sample against FM_a/b.out and
    FR_a/b.out of the compiled reference (Model.cpp:431-432, 500-503, 638-639) -- ModelBase, ModelStandard, ModelChallenger."""
    nblocks = 4
    x = synth.receiver_stream(block * nblocks, sample_rate=rate, receiver_id=150 + model, gap_slots=(1, 2))
    data = synth.to_cu8(x) if fmt == "cu8" else x
    per = 1 if fmt == "cf32" else 2
    chk = (checkers.Ref if checkers.have_ref() else checkers.Oracle)(model=model, rate=rate, fmt=fmt, taps=True)
    chk.feed_blocks(data, block)
    want = {w: chk.tapf(w) for w in (6, 7, 8, 9)}
    g = gpu.AisGpu(sample_rate=rate, block_len=block, input_format=_FMT[fmt], model={1: gpu.MODEL_BASE, 0: gpu.MODEL_STANDARD, 4: gpu.MODEL_CHALLENGER}[model], taps=True)
    L = block // (rate // 48000)
    for b in range(nblocks):
        g.submit(0, data[b * block * per:(b + 1) * block * per])
        g.run()
        g.sync_outputs()
        for w in (6, 7, 8, 9):
            got = g.tapf(w)
            assert len(got) == L
            assert np.array_equal(got.view(np.uint32), want[w][b * L:(b + 1) * L].view(np.uint32)), "tap %d block %d" % (w, b)
    assert np.any(want[8] != 0)
    g.close()


def _oracle_outputs(x, block, rate=1536000):
    """Oracle outputs of one CF32 stream fed in blocks: hard bits [2][5][G], levels [2][G], ppm [2][W]."""
    o = checkers.Oracle(model=2, rate=rate, fmt="cf32", taps=True)
    o.feed_blocks(x, block)
    bits = [[o.bits(ch, j)[0] for j in range(5)] for ch in range(2)]
    lvl = [o.bits(ch, 0)[1] for ch in range(2)]
    ppm = [o.tap_ppm(2), o.tap_ppm(3)]
    o.close()
    return bits, lvl, ppm


@pytest.mark.parametrize("R,unique", [(256, 32), (40, 8)])
def test_benchmarked_path_distinct_receivers_vs_oracle(R, unique):
    """The configuration bench.py measures -- BASELINE configs[3]: R batched receivers x 786,432 CF32 samples resident in HBM,
    DEFAULT path (taps off: fused derotation/FIR back end, spectral analysis at the end of the front-end waves, chunk-parallel
    PhaseSearch), R DISTINCT streams (ais-catcher_amd/workload.py, the generator bench.py uses) -- two blocks, EVERY receiver's
    hard bits, levels and ppm bit-exact against the oracle.  40 receivers: a channel count that is not a multiple of 64."""
    import concurrent.futures
    from ais_catcher_amd import workload
    block, nblocks = workload.BLOCK, 2
    data = workload.resident_batch(torch, R, nblocks, seed=7, unique=unique)
    g = gpu.AisGpu(n_receivers=R, block_len=block)  # taps=False: the default (fused) path
    with concurrent.futures.ThreadPoolExecutor(max_workers=16) as ex:
        futs = [ex.submit(_oracle_outputs, workload.host_stream(data, r, list(range(nblocks))), block) for r in range(R)]
        gd = wd = 0
        got = []
        for b in range(nblocks):
            g.submit_device(data[b].data_ptr(), block)
            g.run()
            g.sync_outputs()
            got.append([[g.fetch(r, ch) for ch in range(2)] for r in range(R)])
        want = [f.result() for f in futs]
    distinct = set()
    for b in range(nblocks):
        n, W = got[b][0][0]["n_groups"], got[b][0][0]["n_windows"]
        for r in range(R):
            bits, lvl, ppm = want[r]
            for ch in range(2):
                out = got[b][r][ch]
                assert out["n_groups"] == n and out["first_group"] == gd
                for j in range(5):
                    assert np.array_equal(out["bits"][j], bits[ch][j][gd:gd + n]), "bits blk %d rx %d ch %d phase %d" % (b, r, ch, j)
                assert _feq(out["lvl"], lvl[ch][gd:gd + n]), "lvl blk %d rx %d ch %d" % (b, r, ch)
                assert _feq(out["ppm"], ppm[ch][wd:wd + W]), "ppm blk %d rx %d ch %d" % (b, r, ch)
            distinct.add(got[b][r][0]["lvl"][:64].tobytes())
        gd += n
        wd += W
    assert len(distinct) == R * nblocks  # the receivers really carried different signals
    g.close()


def test_benchmarked_path_with_device_frame_decoders_vs_oracle():
    """40 distinct receivers (80 channels: the second 64-channel row block is partial) on the default path WITH the frame decoders
    on the device: hard bits / levels / ppm against the oracle and every receiver's NMEA text (frames from aisgpu_frames()
    through the host tail) against the oracle's, in order."""
    import threading
    from ais_catcher_amd import host, workload
    R, block, nblocks = 40, 131072, 6
    data = workload.resident_batch(torch, R, nblocks, seed=11, unique=8, block=block)
    streams = [workload.host_stream(data, r, list(range(nblocks))) for r in range(R)]
    want_nmea = []
    for x in streams:
        c = checkers.Oracle()
        c.feed_blocks(x, block)
        want_nmea.append(c.nmea())
    # (1) decisions with the decoders running beside them
    g = gpu.AisGpu(n_receivers=R, block_len=block, gpu_decode=True)
    want = [_oracle_outputs(x, block) for x in streams]
    gd = wd = 0
    nframes = 0
    for b in range(nblocks):
        g.submit_device(data[b].data_ptr(), block)
        g.run()
        g.sync_outputs()
        nframes += len(g.frames())
        for r in range(R):
            for ch in range(2):
                out = g.fetch(r, ch)
                n, W = out["n_groups"], out["n_windows"]
                for j in range(5):
                    assert np.array_equal(out["bits"][j], want[r][0][ch][j][gd:gd + n]), "bits blk %d rx %d ch %d phase %d" % (b, r, ch, j)
                assert _feq(out["lvl"], want[r][1][ch][gd:gd + n]) and _feq(out["ppm"], want[r][2][ch][wd:wd + W])
        gd += n
        wd += W
    g.close()
    assert nframes >= sum(len(w) for w in want_nmea) // 2
    # (2) end to end: frames -> host tail -> NMEA, one thread per receiver like the reference's device threads
    host.reset_sequence()
    batch = host.Batch(n_receivers=R, block_len=block, gpu_decode=True)
    models = [host.ModelDefaultGPU(block_len=block, batch=batch, rx=r) for r in range(R)]

    def work(r):
        for b in range(nblocks):
            models[r].receive(streams[r][b * block:(b + 1) * block])
    th = [threading.Thread(target=work, args=(r,)) for r in range(R)]
    [t.start() for t in th]
    [t.join() for t in th]
    for r in range(R):
        assert models[r].nmea() == want_nmea[r], "rx %d" % r  # in the reference's order
        models[r].close()
    batch.close()
    assert sum(len(w) for w in want_nmea) >= R


def test_batch_survives_receivers_that_end_early_or_stall():
    """GpuBatch (host/gpu_model.*): a receiver whose input ends calls leave() and the others stop waiting for it; one that just
    stops delivering is evicted after the timeout (the reference marks such a device lost, Device/Device.h:60-61) and gets
    AISGPU_ERR_STATE when it comes back; the remaining receivers' NMEA is untouched by either."""
    import threading
    import time
    from ais_catcher_amd import host
    R, block, nblocks = 4, 131072, 6
    xs = [synth.receiver_stream(block * nblocks, receiver_id=170 + r, gap_slots=(1, 2)) for r in range(R)]
    want = []
    for x in xs:
        c = checkers.Oracle()
        c.feed_blocks(x, block)
        want.append(c.nmea())
    batch = host.Batch(n_receivers=R, block_len=block)
    batch.set_timeout(1500)
    models = [host.ModelDefaultGPU(block_len=block, batch=batch, rx=r) for r in range(R)]
    status = [[] for _ in range(R)]

    def run(r):
        for b in range(nblocks):
            if r == 1 and b == 2:      # end of input after two blocks
                models[r].leave()
                return
            if r == 3 and b == 3:      # stalls (longer than the timeout), then comes back
                time.sleep(4.0)
            status[r].append(models[r].receive(xs[r][b * block:(b + 1) * block]))

    th = [threading.Thread(target=run, args=(r,)) for r in range(R)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert status[0] == [0] * nblocks and status[2] == [0] * nblocks
    assert status[1] == [0, 0]
    assert status[3][:3] == [0, 0, 0] and status[3][3] == 4 and batch.active() == 2
    for r in (0, 2):
        assert models[r].nmea() == want[r] and len(want[r]) >= 2
    # what the receivers that dropped out had decoded until then is a prefix of their reference output
    assert models[1].nmea() == want[1][:len(models[1].nmea())]
    for m in models:
        m.close()
    batch.close()


@pytest.mark.skipif(not checkers.have_refgpu(), reason="oracle/_ref/libaisrefgpu.so not built (needs /root/reference at build time)")
@pytest.mark.parametrize("gpu_model,cpu_model,rate,fmt,block,nblocks,kw", [
    (12, 2, 1536000, "cf32", 786432, 4, {}), (12, 2, 1536000, "cu8", 131072, 12, {}), (12, 2, 1536000, "cu8", 131072, 8, {"fp_ds": True}),
    (12, 2, 1536000, "cf32", 131072, 8, {"ps_ema": False}), (12, 2, 288000, "cf32", 49152, 12, {}), (12, 2, 6000000, "cs16", 786432, 4, {}),
    (14, 4, 1536000, "cf32", 131072, 10, {}), (14, 4, 6000000, "cf32", 786432, 5, {}),
    (20, 0, 1536000, "cf32", 131072, 10, {}), (21, 1, 1536000, "cf32", 131072, 10, {}), (21, 1, 768000, "cu8", 65536, 16, {})])
def test_reference_binding_compiled_against_the_real_reference(gpu_model, cpu_model, rate, fmt, block, nblocks, kw):
    """Row 8(b), for real: integration/reference/Source/DSP/GPU/ModelGPU.cpp -- an AIS::Model subclass (DSP/Model.h:76-126) that
    takes the device's RAW blocks through the reference's own Connection<RAW> (Library/Stream.h), calls the C ABI, and feeds the
    reference's own AIS::Decoder objects with their Reset mesh -- is compiled against the reference's headers and linked with its
    unmodified objects (oracle/Makefile: refgpu).  Engine 12 / 14 (GPU) must print what engine 2 / 4 (the reference's CPU chain)
    prints from the same binary, including the settings that arrive through SetKey (Model.cpp:358-402, 579-594)."""
    x = synth.receiver_stream(block * nblocks, sample_rate=rate, receiver_id=180 + gpu_model, gap_slots=(1, 2), type5_every=4)
    data = {"cu8": synth.to_cu8, "cs16": synth.to_cs16, "cf32": lambda v: v}[fmt](x)
    out = []
    for model in (cpu_model, gpu_model):
        m = checkers.RefGpu(model=model, rate=rate, fmt=fmt, **kw)
        m.feed_blocks(data, block)
        out.append((m.nmea(), m.msg_meta()))
        m.close()
    assert out[0][0] == out[1][0] and len(out[0][0]) >= 3
    assert np.array_equal(out[0][1][0], out[1][1][0]) and np.array_equal(out[0][1][1], out[1][1][1])  # tag.level, tag.ppm per message


@pytest.mark.parametrize("rate,fmt,block,nblocks,afc_wide,droop,extra", [
    (1536000, "cf32", 131072, 6, False, True, {}), (1536000, "cf32", 131072, 6, True, False, {}), (1536000, "cu8", 786432, 2, False, False, {}),
    (768000, "cf32", 65536, 8, False, False, {}), (3072000, "cf32", 262144, 5, False, False, {}), (1536000, "cu8", 131072, 6, True, False, {"fp_ds": True}),
    (192000, "cf32", 16384, 12, False, False, {})])
def test_afc_wide_off_and_droop_off_taps(rate, fmt, block, nblocks, afc_wide, droop, extra):
    """`-go AFC_WIDE off` / `-go DROOP off` (aisgpu_cfg.afc_wide / .droop; Model.cpp:536-540, 384-386, 162-327) on the materialised
    path: all float taps (48 kHz front end, CGF, FIR), hard bits, levels and ppm bit-exact against the oracle -- which is pinned to
    the compiled reference with the same two keys in tests/test_oracle_vs_ref.py::test_afc_wide_off_and_droop_off."""
    xs = [synth.receiver_stream(block * nblocks, sample_rate=rate, receiver_id=330 + r, gap_slots=(0, 2)) for r in range(2)]
    if fmt == "cu8":
        xs = [synth.to_cu8(x) for x in xs]
    _run_gpu_vs_oracle(xs, rate, fmt, block, nblocks, afc_wide=afc_wide, droop=droop, **extra)


@pytest.mark.parametrize("rate,fmt,block,nblocks,afc_wide,droop,extra", [
    (1536000, "cf32", 786432, 3, False, True, {}), (1536000, "cf32", 786432, 3, True, False, {}), (1536000, "cf32", 131072, 6, False, False, {}),
    (768000, "cf32", 393216, 3, False, False, {}), (6000000, "cf32", 786432, 5, False, False, {}), (6000000, "cf32", 786432, 5, False, True, {}),
    (6144000, "cf32", 786432, 3, True, False, {}), (2400000, "cf32", 393216, 5, False, False, {}), (288000, "cf32", 49152, 8, False, False, {}),
    (1536000, "cf32", 786432, 2, False, False, {"ps_ema": False})])
def test_afc_wide_off_and_droop_off_default_path(rate, fmt, block, nblocks, afc_wide, droop, extra):
    """The same two keys on the path bench.py measures (spectral analysis inside the front-end waves, fused derotation / FIR,
    chunk-parallel PhaseSearch), on the direct, the resampled (6 MSPS, 2.4 MSPS), the six-stage and the decimate-by-3 ladders."""
    xs = [synth.receiver_stream(block * nblocks, sample_rate=rate, receiver_id=340 + r, gap_slots=(0, 2)) for r in range(3)]
    _run_outputs_vs_oracle(xs, rate, fmt, block, nblocks, afc_wide=afc_wide, droop=droop, **extra)


@pytest.mark.skipif(not checkers.have_refgpu(), reason="oracle/_ref/libaisrefgpu.so not built (needs /root/reference at build time)")
@pytest.mark.parametrize("gpu_model,cpu_model,rate,fmt,block,nblocks,kw", [
    (12, 2, 1536000, "cf32", 786432, 3, {"afc_wide": False}), (12, 2, 1536000, "cu8", 131072, 10, {"droop": False}),
    (12, 2, 6000000, "cf32", 786432, 4, {"afc_wide": False, "droop": False}), (14, 4, 1536000, "cf32", 131072, 10, {"afc_wide": False, "droop": False}),
    (12, 2, 768000, "cf32", 65536, 12, {"afc_wide": False, "droop": False})])
def test_reference_binding_afc_wide_and_droop_keys(gpu_model, cpu_model, rate, fmt, block, nblocks, kw):
    """The two keys arriving the reference's way -- Model::SetKey(KEY_SETTING_AFC_WIDE / KEY_SETTING_DROOP, "OFF") on the
    AIS::Model subclass of the binding (ModelGPU.cpp: SetKey -> aisgpu_cfg.afc_wide / .droop) -- against the reference's own
    engine given the same keys, from the same binary: NMEA text, tag.level and tag.ppm per message."""
    x = synth.receiver_stream(block * nblocks, sample_rate=rate, receiver_id=350 + gpu_model, gap_slots=(1, 2))
    data = synth.to_cu8(x) if fmt == "cu8" else x
    out = []
    for model in (cpu_model, gpu_model):
        m = checkers.RefGpu(model=model, rate=rate, fmt=fmt, **kw)
        m.feed_blocks(data, block)
        out.append((m.nmea(), m.msg_meta()))
        m.close()
    assert out[0][0] == out[1][0] and len(out[0][0]) >= 3
    assert np.array_equal(out[0][1][0], out[1][1][0]) and np.array_equal(out[0][1][1], out[1][1][1])


@pytest.mark.skipif(not checkers.have_refgpu(), reason="oracle/_ref/libaisrefgpu.so not built (needs /root/reference at build time)")
@pytest.mark.parametrize("rate,fmt,block,nblocks,gpu_decode", [(1536000, "cf32", 786432, 4, False), (1536000, "cu8", 131072, 16, False), (6000000, "cf32", 786432, 4, False),
                                                               (1536000, "cf32", 786432, 4, True), (1536000, "cu8", 131072, 16, True), (288000, "cf32", 49152, 12, False)])
def test_reference_binding_engine_v2(rate, fmt, block, nblocks, gpu_decode):
    """Round 6: AIS::ModelEngineV2 in the reference-side binding (ModelEngineV2GPU, engine 31): by default the GPU front end feeds the
    reference's OWN V2::Engine objects the 48 kHz channels (chain.outC48a >> V2_a: DSP/Model.cpp:452-453); with the decoders on the
    device the whole engine runs there (kv2_engine) and the frames go to the tail of the reference's decoder objects.  Either way
    engine 31 must print what the reference's engine 11 prints from the same binary: NMEA text, tag.level, tag.ppm per message."""
    x = synth.receiver_stream(block * nblocks, sample_rate=rate, receiver_id=390, gap_slots=(1, 2), type5_every=4)
    data = synth.to_cu8(x) if fmt == "cu8" else x
    out = []
    for model, kw in ((11, {}), (31, {"gpu_decode": gpu_decode})):
        m = checkers.RefGpu(model=model, rate=rate, fmt=fmt, **kw)
        m.feed_blocks(data, block)
        out.append((m.nmea(), m.msg_meta()))
        m.close()
    assert out[0][0] == out[1][0] and len(out[0][0]) >= 3
    assert np.array_equal(out[0][1][0], out[1][1][0]) and np.array_equal(out[0][1][1], out[1][1][1])


@pytest.mark.skipif(not checkers.have_refgpu(), reason="oracle/_ref/libaisrefgpu.so not built (needs /root/reference at build time)")
@pytest.mark.parametrize("rate,fmt,block,nblocks,gpu_decode", [(96000, "cf32", 1024 * 48, 8, False), (192000, "cu8", 2048 * 24, 8, False), (150000, "cs16", 2048 * 30, 8, False),
                                                               (48000, "cf32", 512 * 40, 8, False), (96000, "cf32", 1024 * 48, 8, True)])
def test_reference_binding_channel_mode_x(rate, fmt, block, nblocks, gpu_decode):
    """Round 6: `-c X` through the reference-side binding -- AIS::Model::setMode(Mode::X) + buildModel('X', 'X', ...) as Receiver.cpp:87-98,
    220 does -- engine 12 against the reference's engine 2 in the same mode from the same binary (one channel, NMEA with channel letter X)."""
    x = synth.receiver_stream(block * nblocks, sample_rate=rate, receiver_id=395, gap_slots=(1, 2), single_channel=True)
    data = {"cu8": synth.to_cu8, "cs16": synth.to_cs16, "cf32": lambda v: v}[fmt](x)
    out = []
    for model, kw in ((2, {}), (12, {"gpu_decode": gpu_decode})):
        m = checkers.RefGpu(model=model, rate=rate, fmt=fmt, mode_x=True, **kw)
        m.feed_blocks(data, block)
        out.append((m.nmea(), m.msg_meta()))
        m.close()
    assert out[0][0] == out[1][0] and len(out[0][0]) >= 5
    assert np.array_equal(out[0][1][0], out[1][1][0]) and np.array_equal(out[0][1][1], out[1][1][1])


def _by_channel(lines):
    """NMEA lines per channel letter, in order (the A / B interleave of the reference depends on how many FIFO blocks a call carried)."""
    out = {}
    for ln in lines:
        out.setdefault(ln.split(",")[4], []).append(ln)
    return out


_FIFO_BLOCK = 24 * 16 * 16384  # Device/FileRAW.h:43


@pytest.mark.skipif(not checkers.have_refgpu(), reason="oracle/_ref/libaisrefgpu.so not built (needs /root/reference at build time)")
@pytest.mark.parametrize("gpu_model,cpu_model,fmt,runs", [(12, 2, "cu8", 20), (12, 2, "cf32", 6), (14, 4, "cu8", 4), (20, 0, "cu8", 3), (21, 1, "cf32", 3), (31, 11, "cu8", 3)])
def test_binding_behind_the_reference_file_reader(tmp_path, gpu_model, cpu_model, fmt, runs):
    """BASELINE configs[0] / configs[1]'s input path, for real: `-r <file> -s 1536000` = the reference's own Device::RAWFile
    (Device/FileRAW.cpp linked unmodified: reader thread -> FIFO -> run thread), whose run thread hands over ONE OR TWO FIFO blocks
    per Receive() depending on how far the reader got (FileRAW.cpp:120-136, Library/FIFO.h:99-109).  The GPU engine re-blocks
    inside GpuChain::Receive; it must print, per channel and in order, what the reference's CPU engine prints from the same file
    through the same reader -- every time, with the reader thread racing (the file ends in a partial block the reader pads)."""
    per = {"cu8": 2, "cf32": 8}[fmt]
    block = _FIFO_BLOCK // per
    x = synth.receiver_stream(block * 4 + block // 3, receiver_id=300 + gpu_model, gap_slots=(1, 2))  # (no two-sentence messages: their sequence digit follows the process-wide A / B order, Message.cpp:28-39)
    fn = str(tmp_path / ("in." + fmt))
    (synth.to_cu8(x) if fmt == "cu8" else x).tofile(fn)
    m = checkers.RefGpu(model=cpu_model, fmt=fmt, filename=fn)
    rc, calls, _ = m.play_file()
    want = _by_channel(m.nmea())
    m.close()
    assert rc == 0 and sum(len(v) for v in want.values()) >= 20
    seen = set()
    for i in range(runs):
        g = checkers.RefGpu(model=gpu_model, fmt=fmt, filename=fn)
        rc, calls, maxb = g.play_file()
        got = _by_channel(g.nmea())
        g.close()
        seen.add((calls, maxb))
        assert rc == 0, "run %d: the chain stopped (calls %d, largest hand-off %d blocks)" % (i, calls, maxb)
        assert got == want, "run %d (calls %d, largest hand-off %d blocks)" % (i, calls, maxb)
    print("hand-off patterns seen (calls, max blocks per call):", sorted(seen))


@pytest.mark.skipif(not checkers.have_refgpu(), reason="oracle/_ref/libaisrefgpu.so not built (needs /root/reference at build time)")
def test_binding_reblocks_whatever_the_device_sends():
    """Deterministic side of the same contract, on the stub device: two FIFO blocks in one call, then one, then two (what
    FIFO::Front(-1) produces), must equal single-block calls; and with an explicit GPU block (GpuPool::setBlockBytes) calls of
    arbitrary sizes -- smaller than a block, not a multiple of it, larger than two -- are cut into that block, the remainder carried."""
    block = _FIFO_BLOCK // 8
    x = synth.receiver_stream(block * 5, receiver_id=311, gap_slots=(1, 2))
    c = checkers.RefGpu(model=2, fmt="cf32")
    c.feed_blocks(x, block)
    want = c.nmea()
    c.close()
    g = checkers.RefGpu(model=12, fmt="cf32")
    for a, b in ((0, 2), (2, 3), (3, 5)):
        assert g.feed(x[a * block:b * block]) == 0
    assert g.nmea() == want and len(want) >= 20
    g.close()

    small = 131072
    y = synth.receiver_stream(small * 12, receiver_id=312, gap_slots=(1, 2))
    c = checkers.RefGpu(model=2, fmt="cf32")
    c.feed_blocks(y, small)
    want = c.nmea()
    c.close()
    checkers.refgpu_block_bytes(small * 8)
    try:
        g = checkers.RefGpu(model=12, fmt="cf32")
        pos = 0
        for n in (1000, small - 1000, 3 * small + 17, 5, small // 2, 2 * small, 10 ** 9):
            n = min(n, len(y) - pos)
            assert g.feed(y[pos:pos + n]) == 0
            pos += n
        assert pos == len(y)
        assert g.nmea() == want and len(want) >= 8
        g.close()
    finally:
        checkers.refgpu_block_bytes(0)


@pytest.mark.skipif(not checkers.have_refgpu(), reason="oracle/_ref/libaisrefgpu.so not built (needs /root/reference at build time)")
def test_three_file_readers_share_one_gpu_context(tmp_path):
    """Three receivers of one process, each behind its own RAWFile (six reference threads racing), one GPU context: the receivers
    meet once per GPU block although their devices' calls carry different numbers of blocks."""
    import threading
    block = _FIFO_BLOCK // 2
    fns, want = [], []
    for r in range(3):
        x = synth.receiver_stream(block * 3 + 12345, receiver_id=320 + r, gap_slots=(1, 2))
        fns.append(str(tmp_path / ("rx%d.cu8" % r)))
        synth.to_cu8(x).tofile(fns[-1])
        c = checkers.RefGpu(model=2, fmt="cu8")
        cu8 = np.fromfile(fns[-1], np.uint8)
        c.feed_blocks(np.concatenate([cu8, np.zeros(4 * block * 2 - len(cu8), np.uint8)]), block)  # (the reader pads the tail with zero BYTES, FileRAW.cpp:103-104)
        want.append(_by_channel(c.nmea()))
        c.close()
    for rep in range(3):
        gs = [checkers.RefGpu(model=12, fmt="cu8", filename=fn) for fn in fns]  # same configuration: one group, one context
        res = [None] * 3
        ths = [threading.Thread(target=lambda i=i: res.__setitem__(i, gs[i].play_file())) for i in range(3)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        for i in range(3):
            assert res[i][0] == 0, res
            assert _by_channel(gs[i].nmea()) == want[i], "receiver %d, hand-offs %r" % (i, res)
        for g in gs:
            g.close()


@pytest.mark.parametrize("env", [{}, {"AISGPU_PS_WARM": "16"}, {"AISGPU_PS_WARM": "64"}, {"AISGPU_SERIAL": "1"}, {"AISGPU_PS_SEQUENTIAL": "1"}])
def test_fused_back_end_and_its_exact_fallback(env, monkeypatch):
    """The default back end (derotation / FIR kernel, then the chunk-parallel PhaseSearch) on 5 receivers = 10 channels (two full
    quads of the PhaseSearch workgroups and one half-empty one).  With a 16- or 64-symbol warm-up the speculative EMA start of every
    chunk is wrong, the verification flags every workgroup, and the sequential kernel must recompute them exactly; plus the
    single-stream schedule and the sequential search."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    xs = [synth.receiver_stream(786432 * 2, receiver_id=210 + r, gap_slots=(0, 2)) for r in range(5)]
    _run_outputs_vs_oracle(xs, 1536000, "cf32", 786432, 2)
    _run_outputs_vs_oracle(xs[:2], 1536000, "cf32", 98304, 5)   # 614 / 615 groups per block: one chunk, group phase rotating


def test_extreme_receiver_only_costs_its_own_quad():
    """A level step of many decades in ONE receiver defeats the speculative warm-up of its chains (the EMA needs more than 256
    symbols to forget): only that receiver's channel quad goes through the sequential fallback; everybody's outputs stay exact."""
    block, nblocks, R = 786432, 2, 6
    xs = [synth.receiver_stream(block * nblocks, receiver_id=220 + r, gap_slots=(0, 2)) for r in range(R)]
    rng = np.random.default_rng(5)
    wild = (rng.standard_normal(block * nblocks) + 1j * rng.standard_normal(block * nblocks)).astype(np.complex64)
    wild[:block // 2] *= np.float32(3e4)
    wild[block // 2:block] *= np.float32(1e-18)
    wild[block:block + block // 3] *= np.float32(1e3)
    wild[block + block // 3:] *= np.float32(1e-12)
    xs[3] = wild
    _run_outputs_vs_oracle(xs, 1536000, "cf32", block, nblocks)


@pytest.mark.parametrize("model,rate,fmt,block,nblocks", [(4, 1536000, "cf32", 131072, 12), (4, 1536000, "cu8", 786432, 3), (4, 6000000, "cf32", 786432, 5),
                                                         (0, 1536000, "cf32", 131072, 12), (0, 768000, "cu8", 65536, 20), (0, 288000, "cf32", 49152, 12), (2, 288000, "cf32", 49152, 12), (2, 1536000, "cf32", 131072, 8),
                                                         (1, 1536000, "cf32", 131072, 12), (1, 1536000, "cu8", 16384, 60), (1, 2400000, "cf32", 393216, 6),
                                                         (4, 3072000, "cf32", 262144, 9), (0, 3072000, "cu8", 262144, 9), (1, 3072000, "cf32", 786432, 3), (2, 3072000, "cf32", 262144, 9)]) # (3072 kSPS: five stages in the front-end waves)
def test_device_decoders_of_the_other_engines(model, rate, fmt, block, nblocks):
    """AISGPU_FLAG_GPU_DECODE beyond ModelDefault: the decoder state machines of ModelChallenger (ten per channel: FM0..FM3,
    the five coherent ones, FM4 per group, any of them resetting the other nine -- Model.cpp:630-674), of ModelStandard (five on
    the deinterleaved discriminator, Model.cpp:505-514) and of ModelBase (DSP::SimplePLL in front of one decoder whose
    StartTraining / StopTraining signals switch the sampler's loop, DSP.cpp:28-57) run on the device; the host only finishes the
    frames (validate, NMEA).  NMEA text and the per-message tag.level / tag.ppm against the compiled reference, in order."""
    from ais_catcher_amd import host
    x = synth.receiver_stream(block * nblocks, sample_rate=rate, receiver_id=230 + model, gap_slots=(0, 2), type5_every=4)
    data = synth.to_cu8(x) if fmt == "cu8" else x
    per = 1 if fmt == "cf32" else 2
    chk = (checkers.Ref if checkers.have_ref() else checkers.Oracle)(model=model, rate=rate, fmt=fmt)
    chk.feed_blocks(data, block)
    host.reset_sequence()
    cls = {4: host.ModelChallengerGPU, 0: host.ModelStandardGPU, 1: host.ModelBaseGPU, 2: host.ModelDefaultGPU}[model]
    m = cls(sample_rate=rate, block_len=block, input_format=_FMT[fmt], gpu_decode=True)
    for b in range(nblocks):
        assert m.receive(data[b * block * per:(b + 1) * block * per]) == 0
    assert len(chk.nmea()) >= 3
    assert m.nmea() == chk.nmea()
    a, c = m.msg_meta(), chk.msg_meta()
    assert np.array_equal(a[0], c[0]) and np.array_equal(a[1], c[1])
    m.close()


def test_frame_decoder_candidate_lists_overflow_falls_back(monkeypatch):
    """A carrier that repeats preamble + start flag every 20 symbols gives every decoder ~245 frame starts per block -- more than
    the event-driven kernels' lists hold (128).  Such a block goes through the sequential kernel on the device (conditional
    launch, same state in, same state out): the frames are those of AISGPU_K7=seq, nothing fails, and the blocks around it run
    event-driven again."""
    block, nb, sps = 786432, 4, 160
    n = block * nb
    pattern = np.array([0, 1] * 4 + [0, 1, 1, 1, 1, 1, 1, 0] + [1, 0, 1, 1], np.uint8)
    lvl = synth.nrzi(np.tile(pattern, n // sps // len(pattern) + 2))
    rect = lvl[np.minimum(np.arange(n) // sps, len(lvl) - 1)]
    phase = np.cumsum(synth._fftconv_same(rect, synth._gauss(sps))) * (np.pi / 2.0) / sps
    carrier = 0.3 * np.exp(1j * (phase - 2.0 * np.pi * 25000.0 / 1536000 * np.arange(n)))
    carrier[:block] = 0.0                      # block 0: ordinary traffic only (event-driven), blocks 1 .. 2: the carrier on channel A
    carrier[3 * block:] = 0.0                  # block 3: ordinary again
    x = (carrier + synth.receiver_stream(n, receiver_id=78, gap_slots=(0, 1)).astype(np.complex128)).astype(np.complex64)

    def run():
        g = gpu.AisGpu(sample_rate=1536000, n_receivers=1, block_len=block, gpu_decode=True)
        out = []
        for b in range(nb):
            g.submit(0, x[b * block:(b + 1) * block])
            g.run()
            g.sync_outputs()
            out.append(sorted((f["ch"], f["group"], f["phase"], f["position"], f["level_sum"], f["start_idx"], f["end_idx"], f["data"]) for f in g.frames()))
        nfb = g.decoder_fallbacks()
        g.close()
        return out, nfb

    got, nfb = run()
    monkeypatch.setenv("AISGPU_K7", "seq")
    want, nseq = run()
    assert nfb == 2 and nseq == 0               # exactly the two carrier blocks
    assert got == want
    assert len(want[0]) >= 1 and len(want[3]) >= 1  # ordinary messages before and behind


@pytest.mark.parametrize("model", [0, 4])
def test_sequential_decoder_kernels_of_the_other_engines(model, monkeypatch):
    """ModelStandard's and ModelChallenger's decoders run event-driven by default (ModelDefault's kernels on the FM rows / the mesh
    of ten in the reference's order within a group); AISGPU_K7=seq keeps the sequential mesh kernels reachable, and alternating
    blocks between the two implementations (AISGPU_K7=alt) shares their state."""
    for mode in ("seq", "alt"):
        monkeypatch.setenv("AISGPU_K7", mode)
        test_device_decoders_of_the_other_engines(model, 1536000, "cf32", 131072, 12)


def test_device_decoders_of_the_other_engines_on_a_noisy_batch():
    """Six receivers with weak, noisy signals (false trainings, aborted frames, CRC failures) through ModelChallenger's
    twenty-decoder mesh and ModelBase's sampler loop on the device against the host decoders fed with the same decisions."""
    import threading
    from ais_catcher_amd import host
    block, nblocks, R = 131072, 10, 6
    streams = [synth.receiver_stream(block * nblocks, receiver_id=240 + r, gap_slots=(0, 2), noise_sigma=0.04 + 0.03 * r) for r in range(R)]
    for model, cls in ((gpu.MODEL_CHALLENGER, host.ModelChallengerGPU), (gpu.MODEL_BASE, host.ModelBaseGPU), (gpu.MODEL_STANDARD, host.ModelStandardGPU)):
        out = []
        for dec in (False, True):
            host.reset_sequence()
            batch = host.Batch(n_receivers=R, block_len=block, model=model, gpu_decode=dec)
            models = [cls(block_len=block, batch=batch, rx=r) for r in range(R)]

            def work(r):
                for b in range(nblocks):
                    models[r].receive(streams[r][b * block:(b + 1) * block])
            th = [threading.Thread(target=work, args=(r,)) for r in range(R)]
            [t.start() for t in th]
            [t.join() for t in th]
            out.append([mm.nmea() for mm in models])
            for mm in models:
                mm.close()
            batch.close()
        assert out[0] == out[1], "model %d" % model
        assert sum(len(o) for o in out[0]) >= R


@pytest.mark.skipif(not checkers.have_ref(), reason="needs the compiled reference (V2::FreqOffset / FMDemod / FilterFL37 called directly)")
@pytest.mark.parametrize("rate,fmt,block", [(1536000, "cf32", 131072), (1536000, "cu8", 786432), (288000, "cf32", 49152)])
def test_engine_v2_device_stages_against_the_reference_structs(rate, fmt, block):
    """f2: what V2::Engine computes from the 48 kHz channel alone runs on the device for every engine block of a batch --
    FreqOffset::Estimate (f and prominence) of the offset-0 / offset-256 windows, midWins' half-block energies, FMDemod with
    atan2_fast + FilterFL37.  Compared as FLOATS, bit for bit, with the reference's own structs (V2Engine.h:32-101) run on the
    device's 48 kHz samples (themselves bit-exact with the reference, test_model_engine_v2_end_to_end)."""
    import ctypes
    lib = ctypes.CDLL(checkers.os.path.join(checkers.ORACLE_DIR, "_ref", "libaisref_strict.so"))
    lib.ref_v2_estimate.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]
    lib.ref_v2_estimate.restype = None
    lib.ref_v2_fm.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    lib.ref_v2_fm.restype = None
    nblocks = 4
    x = synth.receiver_stream(block * nblocks, sample_rate=rate, receiver_id=250, gap_slots=(0, 2))
    data = synth.to_cu8(x) if fmt == "cu8" else x
    per = 1 if fmt == "cf32" else 2
    g = gpu.AisGpu(sample_rate=rate, block_len=block, input_format=_FMT[fmt], model=gpu.MODEL_V2, taps=True)
    L = block // (rate // 48000)
    W = L // 512
    stream = [np.zeros(512, np.complex64), np.zeros(512, np.complex64)]   # the engine's look-back in front of the stream
    filt_all = [[], []]
    for b in range(nblocks):
        g.submit(0, data[b * block * per:(b + 1) * block * per])
        g.run()
        g.sync_outputs()
        for ch in range(2):
            o = g.fetch(0, ch)
            prev_tail = stream[ch][-512:]
            cur = o["c48"]
            ext = np.concatenate([prev_tail, cur])            # sample -512 .. L-1 of this block
            v2 = o["v2"]
            for w in range(2 * W):
                win = np.ascontiguousarray(ext[256 * w:256 * w + 512])
                f, prom = ctypes.c_float(), ctypes.c_float()
                lib.ref_v2_estimate(win.ctypes.data, ctypes.byref(f), ctypes.byref(prom))
                assert np.float32(f.value).view(np.uint32) == v2["f"][w:w + 1].view(np.uint32)[0], "f blk %d ch %d window %d" % (b, ch, w)
                assert np.float32(prom.value).view(np.uint32) == v2["prom"][w:w + 1].view(np.uint32)[0], "prominence blk %d ch %d window %d" % (b, ch, w)
            for i in range(W + 1):
                e = np.float32(0.0)
                seg = ext[512 * i:512 * i + 256]
                for z in seg:  # midWins sums in order
                    e = np.float32(e + np.float32(np.float32(z.real * z.real) + np.float32(z.imag * z.imag)))
                assert e.view(np.uint32) == v2["energy"][i:i + 1].view(np.uint32)[0], "energy blk %d ch %d %d" % (b, ch, i)
            stream[ch] = np.concatenate([stream[ch], cur])
            filt_all[ch].append((g.tapf(6 + ch), g.tapf(8 + ch), o["fm_bits"]))
    g.close()
    for ch in range(2):
        xs = np.ascontiguousarray(stream[ch][512:])
        n = len(xs)
        disc, filt = np.zeros(n, np.float32), np.zeros(n, np.float32)
        lib.ref_v2_fm(xs.ctypes.data, n, disc.ctypes.data, filt.ctypes.data)
        got_d = np.concatenate([t[0] for t in filt_all[ch]])
        got_f = np.concatenate([t[1] for t in filt_all[ch]])
        got_b = np.concatenate([t[2] for t in filt_all[ch]])
        assert np.array_equal(got_d.view(np.uint32), disc.view(np.uint32)), "FMDemod ch %d" % ch
        assert np.array_equal(got_f.view(np.uint32), filt.view(np.uint32)), "FilterFL37 ch %d" % ch
        assert np.array_equal(got_b.astype(bool), filt > 0)
        assert np.any(filt != 0)


@pytest.mark.parametrize("model", [2, 11])
def test_pipelined_batch_hand_off(model):
    """GpuBatch::setPipelined: receive() of block f returns once f has been started on the device, with block f-1 decoded; the
    receivers copy block f+1 in (double-buffered pinned staging, H2D on a copy stream) and decode f-1 while the device runs f.
    Same NMEA as the checker once the last block has been flushed -- also with the frame decoders on the device.  ModelEngineV2
    (whose outputs are copied to the host inside aisgpu_run(), into two sets of slots by input block) goes the same way."""
    import threading
    from ais_catcher_amd import host
    R, block, nblocks = 5, 131072, 7
    xs = [synth.receiver_stream(block * nblocks, receiver_id=260 + r, gap_slots=(0, 2)) for r in range(R)]
    want = []
    for x in xs:
        c = checkers.Oracle(model=model)
        c.feed_blocks(x, block)
        want.append(c.nmea())
    cls = host.ModelEngineV2GPU if model == 11 else host.ModelDefaultGPU
    for dec in ((False,) if model == 11 else (False, True)):
        host.reset_sequence()
        batch = host.Batch(n_receivers=R, block_len=block, gpu_decode=dec, model=model)
        batch.set_pipelined(True)
        models = [cls(block_len=block, batch=batch, rx=r) for r in range(R)]
        seen_after_first = [None] * R

        def run(r):
            for b in range(nblocks):
                assert models[r].receive(xs[r][b * block:(b + 1) * block]) == 0
                if b == 0:
                    seen_after_first[r] = len(models[r].nmea())
            models[r].flush()

        th = [threading.Thread(target=run, args=(r,)) for r in range(R)]
        [t.start() for t in th]
        [t.join() for t in th]
        assert seen_after_first == [0] * R   # one block deep: nothing is decoded in the first call
        for r in range(R):
            assert models[r].nmea() == want[r] and len(want[r]) >= 2, "rx %d dec %s" % (r, dec)
            models[r].close()
        batch.close()


@pytest.mark.parametrize("mode", [None, "seq", "alt"])
@pytest.mark.parametrize("case", ["busy", "quiet", "short_blocks", "huge_blocks", "long_frames", "batch"])
def test_model_base_chunk_parallel_sampler_and_decoder(case, mode, monkeypatch):
    """ModelBase with AISGPU_FLAG_GPU_DECODE: DSP::SimplePLL (whose loop gain follows the decoder's StartTraining / StopTraining,
    DSP.cpp:28-57, Model.cpp:428-435) and its decoder as the chunk-parallel kernels k7b_spec / k7b_task / k7b_walk / k7b_emit: speculative
    chunks from a fresh state (the block's first one warmed up on the previous block's tail), the exact loop wherever a chunk
    boundary's speculative state is not the previous chunk's end state (frames in flight: back-to-back bursts keep a task running
    across several chunks; silence: nothing to converge on; `long_frames`: 1,000-bit binary messages that lose their closing flag
    keep the decoder in DATAFCS -- and the task in its word-parallel frame mode -- for a thousand symbols, across block
    boundaries), merged where the states meet again.  NMEA text and per-message level / ppm against the compiled reference; `seq` = k7_base alone,
    `alt` = the two implementations take turns block by block on the same DecState."""
    from ais_catcher_amd import host
    if mode:
        monkeypatch.setenv("AISGPU_K7", mode)
    R = 1
    if case == "busy":      # bursts in consecutive slots: hardly any boundary sees a decoder in TRAINING
        block, nblocks, kw = 786432, 3, dict(gap_slots=(0, 0), type5_every=3)
    elif case == "quiet":   # long gaps, and a stretch of exact silence (no sign changes: the speculative phase cannot converge)
        block, nblocks, kw = 786432, 3, dict(gap_slots=(6, 9))
    elif case == "short_blocks":  # 512 / 4096 samples at 48 kHz per block: a single (partial) chunk, then two chunks
        block, nblocks, kw = 16384, 48, dict(gap_slots=(0, 2))
    elif case == "huge_blocks":   # 65,536 samples at 48 kHz per block: more chunks than the walk's notes hold -- the library falls back to k7_base alone
        block, nblocks, kw = 2097152, 2, dict(gap_slots=(1, 3))
    elif case == "long_frames":
        block, nblocks, kw = 786432, 3, dict(gap_slots=(2, 5))
    else:
        R, block, nblocks, kw = 5, 393216, 4, dict(gap_slots=(0, 3))  # (no multi-sentence messages: their sequence digit is process-global)
    xs = [synth.receiver_stream(block * nblocks, receiver_id=300 + r, **kw) for r in range(R)]
    if case == "quiet":
        xs[0][block // 3: block // 3 + 200000] = 0
    if case == "long_frames":  # type 8 (never abandoned for its type, Marine/AIS.cpp:111-142), 168 characters, the last fifth of the burst missing
        long_msg = "8" + ("0123456789:;<=>?@ABCDEFGHIJKLMNOPQRSTUVW`abcdefghijklmnopqrstuvw" * 3)[:167]
        burst = synth.gmsk_burst(long_msg, 1536000)
        rng = np.random.default_rng(77)
        for k in range(9):
            at = int(rng.integers(0, len(xs[0]) - len(burst)))
            if k == 0:
                at = block - len(burst) // 2   # one of them across the first block boundary
            cut = burst[: int(len(burst) * rng.uniform(0.55, 0.95))]
            t = np.arange(len(cut))
            fc = (25000.0 if k & 1 else -25000.0) + rng.uniform(-200.0, 200.0)
            xs[0][at:at + len(cut)] += (0.6 * cut * np.exp(1j * (2.0 * np.pi * fc / 1536000.0 * t + rng.uniform(0, 6.28)))).astype(np.complex64)
    if case == "short_blocks":
        xs.append(synth.receiver_stream(131072 * 6, receiver_id=311, gap_slots=(0, 1)))
    want, got = [], []
    for i, x in enumerate(xs):
        blk = 131072 if (case == "short_blocks" and i == len(xs) - 1) else block
        chk = (checkers.Ref if checkers.have_ref() else checkers.Oracle)(model=1, rate=1536000, fmt="cf32")
        chk.feed_blocks(x, blk)
        want.append((chk.nmea(), chk.msg_meta()))
    if R > 1:
        import threading
        host.reset_sequence()
        batch = host.Batch(n_receivers=R, block_len=block, gpu_decode=True, model=gpu.MODEL_BASE)
        models = [host.ModelBaseGPU(block_len=block, batch=batch, rx=r, gpu_decode=True) for r in range(R)]

        def work(r):
            for b in range(nblocks):
                assert models[r].receive(xs[r][b * block:(b + 1) * block]) == 0
        th = [threading.Thread(target=work, args=(r,)) for r in range(R)]
        [t.start() for t in th]
        [t.join() for t in th]
        for r in range(R):
            assert models[r].nmea() == want[r][0] and len(want[r][0]) >= 2, "rx %d" % r  # in the reference's order
            models[r].close()
        batch.close()
        return
    for i, x in enumerate(xs):
        blk = 131072 if (case == "short_blocks" and i == len(xs) - 1) else block
        host.reset_sequence()
        m = host.ModelBaseGPU(block_len=blk, gpu_decode=True)
        for b in range(len(x) // blk):
            assert m.receive(x[b * blk:(b + 1) * blk]) == 0
        assert m.nmea() == want[i][0] and len(want[i][0]) >= 2, "stream %d" % i
        a, c = m.msg_meta(), want[i][1]
        assert np.array_equal(a[0], c[0]) and np.array_equal(a[1], c[1])
        m.close()


@pytest.mark.parametrize("model", ["default", "standard", "challenger"])
def test_event_driven_decoders_on_many_distinct_receivers_equal_sequential(model, monkeypatch):
    """The event-driven decoder kernels (candidate scan, one run per possible frame, the walk; the walk copies a wave's completed
    messages out one lane per message) on a batch whose decoders all do something different: 24 receivers with their own burst
    schedules and noise, three blocks.  Every frame -- decoder, group, length, bits, level sum, start / end index -- must be the
    one the sequential kernels (k7 = seq) produce."""
    block, nblocks, R = 393216, 3, 24
    mdl = {"default": gpu.MODEL_DEFAULT, "standard": gpu.MODEL_STANDARD, "challenger": gpu.MODEL_CHALLENGER}[model]
    rng = np.random.default_rng(77)
    xs = [synth.receiver_stream(block * nblocks, receiver_id=700 + r, gap_slots=(int(rng.integers(0, 3)), int(rng.integers(3, 6))),
                                type5_every=int(rng.integers(0, 5))) for r in range(R)]

    def used_bits(f):
        n = f["position"]
        d = bytearray(f["data"][:(n + 7) // 8])
        if n % 8:
            d[-1] &= (1 << (n % 8)) - 1
        return bytes(d)

    def run():
        gpu.apply_env_options()
        g = gpu.AisGpu(sample_rate=1536000, n_receivers=R, block_len=block, model=mdl, gpu_decode=True)
        out = []
        for b in range(nblocks):
            for r in range(R):
                g.submit(r, xs[r][b * block:(b + 1) * block])
            g.run()
            g.sync_outputs()
            out.append(sorted((f["rx"], f["ch"], f["phase"], f["group"], f["position"], f["level_sum"], f["start_idx"], f["end_idx"], used_bits(f)) for f in g.frames()))
        nfb = g.decoder_fallbacks()
        g.close()
        return out, nfb

    got, nfb = run()
    monkeypatch.setenv("AISGPU_K7", "seq")
    want, _ = run()
    monkeypatch.delenv("AISGPU_K7")
    for b in range(nblocks):
        assert got[b] == want[b], "block %d: %d / %d frames" % (b, len(got[b]), len(want[b]))
    assert sum(len(b) for b in want) >= 5 * R and nfb == 0


def test_model_base_many_distinct_receivers_chunk_parallel_equals_sequential(monkeypatch):
    """The chunk-parallel ModelBase kernels on a batch whose lanes all do something different: 40 receivers with their own burst
    schedules, gaps, truncated bursts and noise (80 channels: one full wave of the speculative pass and a partial one, ten task
    waves per boundary), four blocks.  Every frame of every receiver -- channel, sample, length, bits -- must be the one the
    sequential kernel (k7 = seq: one lane per channel, symbol by symbol) produces from the same state sequence, and the NMEA of four
    of the receivers the reference's."""
    from ais_catcher_amd import host
    block, nblocks, R = 393216, 4, 40
    rng = np.random.default_rng(2024)
    xs = []
    for r in range(R):
        x = synth.receiver_stream(block * nblocks, receiver_id=900 + r, gap_slots=(int(rng.integers(0, 3)), int(rng.integers(3, 7))), type5_every=int(rng.integers(0, 5)))
        for _ in range(int(rng.integers(0, 4))):   # a few bursts lose their tail (no closing flag: long stays in DATAFCS)
            at = int(rng.integers(0, len(x) - 60000))
            x[at:at + int(rng.integers(2000, 40000))] *= np.float32(0.02)
        xs.append(x)

    def run():
        gpu.apply_env_options()
        g = gpu.AisGpu(sample_rate=1536000, n_receivers=R, block_len=block, model=gpu.MODEL_BASE, gpu_decode=True)
        out = []
        for b in range(nblocks):
            for r in range(R):
                g.submit(r, xs[r][b * block:(b + 1) * block])
            g.run()
            g.sync_outputs()
            out.append(sorted((f["rx"], f["ch"], f["group"], f["position"], used_bits(f)) for f in g.frames()))
        nfb = g.decoder_fallbacks()
        g.close()
        return out, nfb

    def used_bits(f):  # the frame's first `position` bits (what the buffer holds behind them is never read, Marine/AIS.h:150-163)
        n = f["position"]
        d = bytearray(f["data"][:(n + 7) // 8])
        if n % 8:
            d[-1] &= (1 << (n % 8)) - 1
        return bytes(d)

    got, nfb = run()
    monkeypatch.setenv("AISGPU_K7", "seq")
    want, _ = run()
    monkeypatch.delenv("AISGPU_K7")
    for b in range(nblocks):
        assert got[b] == want[b], "block %d: %d / %d frames, first difference %s" % (
            b, len(got[b]), len(want[b]), next(((x[:4], y[:4]) for x, y in zip(got[b], want[b]) if x != y), None))
    assert sum(len(b) for b in want) >= 8 * R and nfb == 0
    for r in (0, 13, 27, 39):
        chk = (checkers.Ref if checkers.have_ref() else checkers.Oracle)(model=1, rate=1536000, fmt="cf32")
        chk.feed_blocks(xs[r], block)
        host.reset_sequence()
        m = host.ModelBaseGPU(block_len=block, gpu_decode=True)
        for b in range(nblocks):
            assert m.receive(xs[r][b * block:(b + 1) * block]) == 0
        assert m.nmea() == chk.nmea(), "rx %d" % r
        m.close()


def test_model_base_frame_list_overflow_falls_back(monkeypatch):
    """A chunk or a boundary task of the chunk-parallel ModelBase kernels that completes more frames than its list takes flags the
    channel, and k7_base decodes that channel's block from the untouched carried state.  With the list capacity set to ONE frame
    (test hook k7b_fcap) back-to-back bursts overflow the lists of boundary tasks in many blocks: the NMEA stays the reference's, the
    fallback counter says that the path ran, and the channels that did not overflow went through the chunk-parallel kernels."""
    from ais_catcher_amd import host
    block, nblocks = 786432, 3
    x = synth.receiver_stream(block * nblocks, receiver_id=431, gap_slots=(0, 0), type5_every=4)
    chk = (checkers.Ref if checkers.have_ref() else checkers.Oracle)(model=1, rate=1536000, fmt="cf32")
    chk.feed_blocks(x, block)
    want = chk.nmea()
    monkeypatch.setenv("AISGPU_K7B_FCAP", "1")
    host.reset_sequence()
    m = host.ModelBaseGPU(block_len=block, gpu_decode=True)
    for b in range(nblocks):
        assert m.receive(x[b * block:(b + 1) * block]) == 0
    assert m.nmea() == want and len(want) >= 10
    m.close()
    gpu.apply_env_options()
    g = gpu.AisGpu(sample_rate=1536000, n_receivers=1, block_len=block, model=gpu.MODEL_BASE, gpu_decode=True)
    for b in range(nblocks):
        g.submit(0, x[b * block:(b + 1) * block])
        g.run()
        g.sync_outputs()
    n_fb = g.decoder_fallbacks()
    g.close()
    assert 1 <= n_fb <= 2 * nblocks, n_fb   # (channel, block) pairs that took the fallback: some, and at most all


@pytest.mark.skipif(not checkers.have_refgpu(), reason="oracle/_ref/libaisrefgpu.so not built (needs /root/reference at build time)")
@pytest.mark.parametrize("gpu_model,cpu_model,opts", [(12, 2, {}), (12, 2, {"gpu_decode": True}), (12, 2, {"pipelined": True}),
                                                      (12, 2, {"gpu_decode": True, "pipelined": True}), (14, 4, {}), (14, 4, {"gpu_decode": True}),
                                                      (20, 0, {}), (20, 0, {"gpu_decode": True}), (21, 1, {}), (21, 1, {"gpu_decode": True})])
def test_eight_reference_receivers_share_one_gpu_context(gpu_model, cpu_model, opts):
    """The batched hand-off compiled against the REAL reference (integration/reference/Source/DSP/GPU/ModelGPU.cpp + GpuBatch):
    eight AIS::Model instances of one libaisrefgpu.so process -- the reference's own Device, TAG, AIS::Decoder, SimplePLL,
    Deinterleave, Message objects -- register with the process-wide GpuPool, their eight device threads meet once per block in ONE
    aisgpu context (n_receivers = 8), and every receiver prints, in order, exactly what the reference's CPU engine prints for its
    stream.  Also with the decoder state machines on the device (frames finished by GpuEmitFrame = the tail of AIS::Decoder::Run)
    and with the pipelined hand-off (messages one block late, Flush() after the last block)."""
    import threading
    R, block, nblocks = 8, 131072, 8
    xs = [synth.receiver_stream(block * nblocks, receiver_id=500 + r, gap_slots=(0, 2)) for r in range(R)]
    want = []
    for x in xs:
        c = checkers.RefGpu(model=cpu_model)
        c.feed_blocks(x, block)
        want.append((c.nmea(), c.msg_meta()))
        c.close()
    models = [checkers.RefGpu(model=gpu_model, **opts) for _ in range(R)]   # all built before the first block: one group, one context

    def run(r):
        for b in range(nblocks):
            assert models[r].feed(xs[r][b * block:(b + 1) * block]) == 0
        if opts.get("pipelined"):
            models[r].flush()

    th = [threading.Thread(target=run, args=(r,)) for r in range(R)]
    [t.start() for t in th]
    [t.join() for t in th]
    for r in range(R):
        assert models[r].nmea() == want[r][0] and len(want[r][0]) >= 2, "rx %d" % r
        a, c = models[r].msg_meta(), want[r][1]
        assert np.array_equal(a[0], c[0]) and np.array_equal(a[1], c[1]), "rx %d: tag.level / tag.ppm per message" % r
        models[r].close()
