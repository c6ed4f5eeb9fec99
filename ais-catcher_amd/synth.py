"""Synthetic dual-channel AIS (GMSK) IQ generator -- SURVEY.md Appendix C / section 8(d).

Builds ITU-R M.1371 frames from NMEA payload strings (ramp, 24-bit training sequence, 0x7E flag,
LSB-first payload + CRC-16/X.25, bit stuffing, 0x7E, NRZI), GMSK-modulates them (BT 0.4, h 0.5,
9600 Bd) and places the bursts slot-aligned at -25 kHz (channel A = the reference's ROT.up branch,
Source/DSP/Model.cpp:341) / +25 kHz (channel B) in a complex baseband stream with AWGN.

The frame layout is what the reference's AIS::Decoder accepts (Source/Marine/AIS.h:82-181:
>4 alternating training bits, exact 01111110 flag, CRC residue check Source/Marine/AIS.cpp:55-64).
Used by tests/ (inputs for parity checks) and bench.py (resident synthetic input); numpy only.
"""
import numpy as np

# pinned payloads: reference python/tests/test_decode.py:12-17 (their decoded fields are pinned there)
PAYLOADS = (
    "15MgK45P3@G?fl0E`JbR0OwT0@MS",
    "177KQJ5000G?tO`K>RA1wUbN0TKH",
    "146i`8001H0k72>O?tWcUa=60`EP",
)
PAYLOAD_TYPE5 = ("55O0W7`00001L@gCWGA2uItLth@DqtL5@F22220j1h742t0Ht0000000" "000000000000000", 2)

BAUD = 9600
SLOT_BITS = 256


def dearmour(payload, fill=0):
    """6-bit de-armouring, MSB first (inverse of Source/Marine/Message.cpp:633-660)."""
    bits = []
    for c in payload:
        v = ord(c) - 48
        if v > 40:
            v -= 8
        bits.extend((v >> (5 - k)) & 1 for k in range(6))
    if fill:
        bits = bits[:-fill]
    return np.array(bits, dtype=np.uint8)


def crc16_x25(bits):
    crc = 0xFFFF
    for b in bits:
        crc = (crc >> 1) ^ 0x8408 if ((int(b) ^ crc) & 1) else crc >> 1
    return crc ^ 0xFFFF


def frame_bits(payload, fill=0):
    """Payload string -> on-air bit sequence before NRZI (Appendix C)."""
    m = dearmour(payload, fill)
    assert len(m) % 8 == 0, "payload must be a whole number of bytes"
    tx = m.reshape(-1, 8)[:, ::-1].reshape(-1)          # each byte LSB first (Message.h:264-273)
    crc = crc16_x25(tx)
    tx = np.concatenate([tx, np.array([(crc >> k) & 1 for k in range(16)], dtype=np.uint8)])
    stuffed = []
    run = 0
    for b in tx:
        stuffed.append(int(b))
        run = run + 1 if b else 0
        if run == 5:
            stuffed.append(0)
            run = 0
    flag = [0, 1, 1, 1, 1, 1, 1, 0]
    bits = [0] * 8 + [0, 1] * 12 + flag + stuffed + flag + [0] * 8
    return np.array(bits, dtype=np.uint8)


def nrzi(bits):
    """Level toggles on a 0 bit, holds on a 1 bit (decoder: Bit = !(d ^ prev), AIS.h:94-96)."""
    lvl = np.empty(len(bits), dtype=np.float64)
    cur = 1.0
    for i, b in enumerate(bits):
        if b == 0:
            cur = -cur
        lvl[i] = cur
    return lvl


_gauss_cache = {}


def _gauss(sps, bt=0.4, span=2):
    key = (sps, bt, span)
    if key not in _gauss_cache:
        t = np.arange(-span * sps, span * sps + 1, dtype=np.float64) / sps
        sigma = np.sqrt(np.log(2.0)) / (2.0 * np.pi * bt)
        g = np.exp(-0.5 * (t / sigma) ** 2)
        _gauss_cache[key] = g / g.sum()
    return _gauss_cache[key]


_burst_cache = {}


def gmsk_burst(payload, sample_rate, fill=0):
    """Unit-amplitude complex baseband GMSK burst (float64 phase, complex128 samples)."""
    key = (payload, sample_rate, fill)
    if key in _burst_cache:
        return _burst_cache[key]
    sps_f = sample_rate / BAUD
    lvl = nrzi(frame_bits(payload, fill))
    n = int(round(len(lvl) * sps_f))
    # rectangular NRZ at a (possibly fractional) samples-per-symbol grid
    idx = np.minimum((np.arange(n) / sps_f).astype(np.int64), len(lvl) - 1)
    rect = lvl[idx]
    g = _gauss(int(round(sps_f)))
    f = np.convolve(rect, g, mode="same") if len(g) < 64 else _fftconv_same(rect, g)
    phase = np.cumsum(f) * (np.pi / 2.0) / sps_f
    out = np.exp(1j * phase)
    _burst_cache[key] = out
    return out


def _fftconv_same(x, g):
    n = len(x) + len(g) - 1
    nf = 1 << (n - 1).bit_length()
    y = np.fft.irfft(np.fft.rfft(x, nf) * np.fft.rfft(g, nf), nf)[:n]
    s = (len(g) - 1) // 2
    return y[s:s + len(x)]


def receiver_stream(n_samples, sample_rate=1536000, receiver_id=0, payloads=PAYLOADS,
                    noise_sigma=0.01, gap_slots=(2, 4), type5_every=0, seed_base=12345,
                    return_schedule=False, single_channel=False):
    """One receiver's complex64 stream of n_samples with slot-aligned bursts on both channels.

    RNG: numpy.default_rng(seed_base + receiver_id). Returns complex64 [n_samples]
    (and the burst schedule [(start_sample, channel, payload)] when return_schedule).
    """
    rng = np.random.default_rng(seed_base + receiver_id)
    x = np.zeros(n_samples, dtype=np.complex128)
    slot = SLOT_BITS * sample_rate / BAUD
    sched = []
    s = int(rng.integers(0, 3))
    k = 0
    while True:
        start = int(round(s * slot))
        if type5_every and (k % type5_every) == type5_every - 1:
            pl, fill = PAYLOAD_TYPE5
        else:
            pl, fill = payloads[k % len(payloads)], 0
        b = gmsk_burst(pl, sample_rate, fill)
        if start + len(b) > n_samples:
            break
        ch = int(rng.integers(0, 2))              # 0 = A (-25 kHz), 1 = B (+25 kHz)
        fc = (0.0 if single_channel else (-25000.0 if ch == 0 else 25000.0)) + rng.uniform(-300.0, 300.0)  # channel mode X: one channel at 0 Hz
        amp = rng.uniform(0.15, 0.5)
        ph0 = rng.uniform(0.0, 2.0 * np.pi)
        t = np.arange(len(b), dtype=np.float64)
        x[start:start + len(b)] += amp * b * np.exp(1j * (2.0 * np.pi * fc / sample_rate * t + ph0))
        sched.append((start, "AB"[ch], pl))
        s += int(np.ceil(len(b) / slot)) + int(rng.integers(gap_slots[0], gap_slots[1] + 1))
        k += 1
    if noise_sigma > 0:
        x += noise_sigma * (rng.standard_normal(n_samples) + 1j * rng.standard_normal(n_samples))
    out = x.astype(np.complex64)
    return (out, sched) if return_schedule else out


def to_cu8(x):
    """complex64 -> interleaved uint8 pairs, round(x*128+128) clipped."""
    v = np.empty((len(x), 2), dtype=np.float32)
    v[:, 0] = x.real
    v[:, 1] = x.imag
    return np.clip(np.round(v * 128.0 + 128.0), 0, 255).astype(np.uint8).reshape(-1)


def to_cs8(x):
    """complex64 -> interleaved int8 pairs, round(x*128) clipped (the reference divides by 128: Utilities/Convert.cpp:266-275)."""
    v = np.empty((len(x), 2), dtype=np.float32)
    v[:, 0] = x.real
    v[:, 1] = x.imag
    return np.clip(np.round(v * 128.0), -128, 127).astype(np.int8).reshape(-1)


def to_cs16(x):
    """complex64 -> interleaved int16 pairs, round(x*32768) clipped (Utilities/Convert.cpp:277-286)."""
    v = np.empty((len(x), 2), dtype=np.float32)
    v[:, 0] = x.real
    v[:, 1] = x.imag
    return np.clip(np.round(v * 32768.0), -32768, 32767).astype(np.int16).reshape(-1)


def expected_nmea(sched):
    """The single-sentence NMEA lines a perfect receiver would print for a schedule."""
    out = []
    for _, ch, pl in sched:
        if len(pl) <= 56:
            body = "AIVDM,1,1,,%s,%s,0" % (ch, pl)
            c = 0
            for ch_ in body:
                c ^= ord(ch_)
            out.append("!%s*%02X" % (body, c))
    return out
