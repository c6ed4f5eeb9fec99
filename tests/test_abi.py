"""CPU: the C-ABI library loads and exports every symbol include/aisgpu.h declares; no compute calls."""
import ctypes
import os
import re

from ais_catcher_amd import gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "aisgpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(aisgpu_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert _declared() == sorted(gpu.EXPORTS)


def test_library_exports_every_declared_symbol():
    lib = gpu.load()
    for name in _declared():
        assert hasattr(lib, name), name


def test_structs_match_header_layout():
    assert ctypes.sizeof(gpu.Cfg) == 10 * 4
    cfg = gpu.Cfg()
    gpu.load().aisgpu_default_cfg(ctypes.byref(cfg))
    assert (cfg.sample_rate, cfg.block_len, cfg.model, cfg.input_format) == (1536000, 786432, 2, 1)


def test_create_rejects_bad_config_without_gpu_work():
    lib = gpu.load()
    cfg = gpu.Cfg()
    lib.aisgpu_default_cfg(ctypes.byref(cfg))
    h = ctypes.c_void_p()
    for bad_rate in (50000, 95999, 13000000):  # below 96k (Model.cpp:109-110); above 12288k
        cfg.sample_rate = bad_rate
        assert lib.aisgpu_create(ctypes.byref(cfg), ctypes.byref(h)) == 1 and not h
    cfg.sample_rate = 1536000
    cfg.block_len = 1000
    assert lib.aisgpu_create(ctypes.byref(cfg), ctypes.byref(h)) == 1 and not h
    assert lib.aisgpu_strerror(2).decode().startswith("no usable")


def test_create_ladder_selection_for_decimate_by_3_rates():
    """Argument validation runs before any device is touched: unsupported -> 1 (ARG); supported -> 0 or 2 (no device here)."""
    lib = gpu.load()
    cfg = gpu.Cfg()
    h = ctypes.c_void_p()

    def rc(rate, block, flags=0):
        lib.aisgpu_default_cfg(ctypes.byref(cfg))
        cfg.sample_rate, cfg.block_len, cfg.flags = rate, block, flags
        r = lib.aisgpu_create(ctypes.byref(cfg), ctypes.byref(h))
        if h:
            lib.aisgpu_destroy(h)
            h.value = None
        return r

    assert rc(288000, 24576 * 4) in (0, 2)                   # DownsampleKFilter ladder, no option needed (Model.cpp:129)
    assert rc(288000, 3072 * 5) == 1                          # not a whole number of the filter's 8192-sample output blocks
    assert rc(2304000, 24576 * 8 * 2, gpu.FLAG_DSK) in (0, 2)
    assert rc(2304000, 512 * 64 * 6) in (0, 2)                # without the option 2304k is resampled up to 3072k
    assert rc(500000, 24576 * 2 * 4, gpu.FLAG_DSK) in (0, 2)  # Upsample in front of DownsampleKFilter: 500k -> 576k
    assert rc(500000, 614400, gpu.FLAG_DSK) == 1              # ... whose flushes must be whole 8192-sample output blocks
    assert rc(500000, 786432) in (0, 2)                       # without the option: resampled up to 768k
    assert rc(250000, 24576 * 3) in (0, 2)                    # 250k -> 288k (no option needed)
    assert rc(96000, 1024 * 16) in (0, 2) and rc(150000, 2048 * 16) in (0, 2) and rc(10000000, 131072 * 6) in (0, 2)
    # channel mode X: 12k .. 192k, one channel (Model.cpp:37-38)
    assert rc(48000, 512 * 8, gpu.FLAG_MODE_X) in (0, 2) and rc(40000, 512 * 8, gpu.FLAG_MODE_X) in (0, 2) and rc(192000, 2048 * 8, gpu.FLAG_MODE_X) in (0, 2)
    assert rc(12000, 512 * 8, gpu.FLAG_MODE_X) in (0, 2) and rc(11999, 512 * 8, gpu.FLAG_MODE_X) == 1 and rc(200000, 4096 * 8, gpu.FLAG_MODE_X) == 1 and rc(48000, 512 * 8) == 1
    # `-go MA on` (Model.cpp:122-126): multiples of 96 kHz, input blocks that are whole 8192-sample output blocks of the downsampler
    assert rc(1536000, 131072, gpu.FLAG_MA_DS) in (0, 2) and rc(2400000, 204800, gpu.FLAG_MA_DS) in (0, 2) and rc(192000, 16384 * 3, gpu.FLAG_MA_DS) in (0, 2)
    assert rc(1536000, 65536, gpu.FLAG_MA_DS) == 1 and rc(1000000, 131072, gpu.FLAG_MA_DS) == 1 and rc(96000, 8192, gpu.FLAG_MA_DS) == 1
    assert rc(48000, 8192, gpu.FLAG_MA_DS | gpu.FLAG_MODE_X) == 1


def test_reference_binding_links_and_fails_loudly_without_a_gpu():
    """oracle/_ref/libaisrefgpu.so = the reference's unmodified objects + integration/reference/Source/DSP/GPU/ModelGPU.cpp (an
    AIS::Model subclass written against the reference's real headers) + libaisgpu.so.  Without a GPU the engine builds (wiring
    only) and its first block reports AISGPU_ERR_NODEV as a std::runtime_error -- there is no CPU fallback behind it."""
    import numpy as np
    import pytest
    import checkers
    if not checkers.have_refgpu():
        pytest.skip("libaisrefgpu.so not built")
    import ctypes
    from ais_catcher_amd import gpu
    if gpu.load().aisgpu_device_count() > 0:
        pytest.skip("a GPU is present: covered by the -m gpu tests")
    m = checkers.RefGpu(model=12)
    assert m.feed(np.zeros(16384, np.complex64)) == -1
    assert m.nmea() == []
    cpu = checkers.RefGpu(model=2)   # the reference's own engine from the same binary still runs
    assert cpu.feed(np.zeros(16384, np.complex64)) == 0
