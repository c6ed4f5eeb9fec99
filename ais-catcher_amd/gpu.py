"""ctypes binding of libaisgpu.so (include/aisgpu.h) -- the C-ABI boundary of the MI355X chain.

No torch types cross this boundary: device buffers are passed as raw pointers (e.g. tensor.data_ptr()).
The library is built in-tree by `make -C ais-catcher_amd/csrc` (see __graft_entry__.build()).
There is NO CPU fallback: if the library or a GPU is missing, construction raises.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AISGPU_LIB") or os.path.join(_HERE, "libaisgpu.so")  # AISGPU_LIB: A/B a kernel build

FMT_CU8, FMT_CF32, FMT_CS8, FMT_CS16 = 0, 1, 2, 3
_ELEMS = {FMT_CU8: 2, FMT_CS8: 2, FMT_CS16: 2, FMT_CF32: 1}  # numpy elements per IQ sample (uint8 / int8 / int16 pairs, complex64)
MODEL_STANDARD = 0
MODEL_BASE = 1
MODEL_V2 = 11
MODEL_DEFAULT = 2
MODEL_CHALLENGER = 4
FLAG_TAPS = 1
FLAG_SERIAL = 2
FLAG_DSK = 4
FLAG_FP_DS = 32
FLAG_PS_BOXCAR = 8
FLAG_GPU_DECODE = 16
FLAG_MODE_X = 64
FLAG_MA_DS = 128

EXPORTS = (
    "aisgpu_default_cfg", "aisgpu_create", "aisgpu_destroy", "aisgpu_submit", "aisgpu_submit_device",
    "aisgpu_run", "aisgpu_sync_outputs", "aisgpu_sync", "aisgpu_fetch", "aisgpu_tap", "aisgpu_stream",
    "aisgpu_frontend_ms", "aisgpu_timing", "aisgpu_strerror", "aisgpu_last_error", "aisgpu_device_count",
    "aisgpu_out_count", "aisgpu_fetch_sub", "aisgpu_selftest", "aisgpu_frames", "aisgpu_ps_fallbacks", "aisgpu_decoder_fallbacks",
    "aisgpu_set_option",
)


class Cfg(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in (
        "sample_rate", "n_receivers", "block_len", "model", "input_format", "afc_wide", "droop",
        "device_id", "flags", "tiles_per_span")]


class Frame(ctypes.Structure):
    _fields_ = [("rx", ctypes.c_int), ("ch", ctypes.c_int), ("phase", ctypes.c_int), ("sub", ctypes.c_int),
                ("group", ctypes.c_int), ("position", ctypes.c_int), ("level_sum", ctypes.c_float),
                ("start_idx", ctypes.c_longlong), ("end_idx", ctypes.c_longlong), ("data", ctypes.c_ubyte * 144)]


class Out(ctypes.Structure):
    _fields_ = [
        ("n_groups", ctypes.c_int),
        ("first_group", ctypes.c_longlong),
        ("bits", ctypes.POINTER(ctypes.c_uint32) * 5),
        ("lvl", ctypes.POINTER(ctypes.c_float)),
        ("n_windows", ctypes.c_int),
        ("ppm", ctypes.POINTER(ctypes.c_float)),
        ("group_window", ctypes.POINTER(ctypes.c_int)),
        ("first_sample48", ctypes.c_longlong),
        ("fm_bits", ctypes.POINTER(ctypes.c_uint32)),
        ("c48", ctypes.POINTER(ctypes.c_float)),
        ("v2_f", ctypes.POINTER(ctypes.c_float)),
        ("v2_prom", ctypes.POINTER(ctypes.c_float)),
        ("v2_energy", ctypes.POINTER(ctypes.c_float)),
    ]


_lib = None


def load():
    """Load libaisgpu.so and declare the prototypes of every symbol include/aisgpu.h exports."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libaisgpu.so is not built (%s); run __graft_entry__.build()" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    vp, ci, cll = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong
    lib.aisgpu_default_cfg.argtypes = [ctypes.POINTER(Cfg)]
    lib.aisgpu_default_cfg.restype = None
    lib.aisgpu_create.argtypes = [ctypes.POINTER(Cfg), ctypes.POINTER(vp)]
    lib.aisgpu_destroy.argtypes = [vp]
    lib.aisgpu_destroy.restype = None
    lib.aisgpu_submit.argtypes = [vp, ci, vp, ci]
    lib.aisgpu_submit_device.argtypes = [vp, vp, cll]
    lib.aisgpu_run.argtypes = [vp]
    lib.aisgpu_sync_outputs.argtypes = [vp]
    lib.aisgpu_sync.argtypes = [vp]
    lib.aisgpu_fetch.argtypes = [vp, ci, ci, ctypes.POINTER(Out)]
    lib.aisgpu_out_count.argtypes = [vp]
    lib.aisgpu_fetch_sub.argtypes = [vp, ci, ci, ci, ctypes.POINTER(Out)]
    lib.aisgpu_ps_fallbacks.argtypes = [vp, ctypes.POINTER(ctypes.c_longlong)]
    lib.aisgpu_decoder_fallbacks.argtypes = [vp, ctypes.POINTER(ctypes.c_longlong)]
    if hasattr(lib, "aisgpu_frames"):
        lib.aisgpu_frames.argtypes = [vp, ctypes.POINTER(ctypes.POINTER(Frame)), ctypes.POINTER(ci)]
    lib.aisgpu_tap.argtypes = [vp, ci, ci, vp, cll]
    lib.aisgpu_tap.restype = cll
    lib.aisgpu_stream.argtypes = [vp]
    lib.aisgpu_stream.restype = vp
    lib.aisgpu_frontend_ms.argtypes = [vp, ctypes.POINTER(ci)]
    lib.aisgpu_frontend_ms.restype = ctypes.c_float
    lib.aisgpu_timing.argtypes = [vp, ci]
    lib.aisgpu_timing.restype = None
    lib.aisgpu_strerror.argtypes = [ci]
    lib.aisgpu_strerror.restype = ctypes.c_char_p
    lib.aisgpu_last_error.argtypes = [vp]
    lib.aisgpu_last_error.restype = ctypes.c_char_p
    lib.aisgpu_device_count.restype = ci
    if hasattr(lib, "aisgpu_set_option"):
        lib.aisgpu_set_option.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
    if hasattr(lib, "aisgpu_selftest"):  # absent from older builds used in A/B runs (AISGPU_LIB)
        lib.aisgpu_selftest.argtypes = [ci, ci, vp, cll]
        lib.aisgpu_selftest.restype = cll
    _lib = lib
    return lib


# Test hooks of the library (include/aisgpu.h: aisgpu_set_option).  The shipped .so reads no environment variable; for the
# tests' convenience THIS wrapper forwards AISGPU_<KEY> from the environment to the option of the same name before a context
# is created (monkeypatch.setenv("AISGPU_PS_WARM", "16") in a test selects the exact-fallback path).
OPTION_KEYS = ("serial", "ps_warm", "ps_sequential", "k7", "k7b_fcap", "fused", "fft_in_k1", "k46", "k1u_spw", "us_k1", "v2_roles")


_forwarded = set()


def apply_env_options(lib=None):
    """Forward AISGPU_<KEY> from the environment to aisgpu_set_option(key).  Only keys that are (or, at the previous call, were) in
    the environment are touched, so an option the application set itself through aisgpu_set_option() survives.  The options are
    process-wide and sampled by aisgpu_create().  An AISGPU_* variable that names no option is an error of the caller's (a renamed
    knob must fail loudly, not select the default path silently)."""
    lib = lib or load()
    if not hasattr(lib, "aisgpu_set_option"):
        return
    known = {"AISGPU_" + k.upper() for k in OPTION_KEYS} | {"AISGPU_LIB", "AISGPU_TRACE", "AISGPU_K7E_STATS", "AISGPU_K7B_STATS"}
    unknown = sorted(v for v in os.environ if v.startswith("AISGPU_") and v not in known)
    if unknown:
        import warnings
        warnings.warn("environment variables that name no aisgpu option: %s (options: %s)" % (", ".join(unknown), ", ".join(OPTION_KEYS)))
    for key in OPTION_KEYS:
        v = os.environ.get("AISGPU_" + key.upper())
        if v:
            lib.aisgpu_set_option(key.encode(), v.encode())
            _forwarded.add(key)
        elif key in _forwarded:   # set by this wrapper earlier and gone from the environment since: back to the default
            lib.aisgpu_set_option(key.encode(), None)
            _forwarded.discard(key)


class AisGpuError(RuntimeError):
    pass


class AisGpu:
    """One context = n_receivers batched dual-channel receivers on one GPU."""

    def __init__(self, sample_rate=1536000, n_receivers=1, block_len=786432, input_format=FMT_CF32,
                 afc_wide=True, droop=True, device_id=0, taps=False, tiles_per_span=0, serial=False, model=MODEL_DEFAULT,
                 dsk=False, ps_ema=True, gpu_decode=False, fp_ds=False, mode_x=False, ma=False):
        self.lib = load()
        apply_env_options(self.lib)
        cfg = Cfg()
        self.lib.aisgpu_default_cfg(ctypes.byref(cfg))
        cfg.sample_rate, cfg.n_receivers, cfg.block_len = sample_rate, n_receivers, block_len
        cfg.model = model
        cfg.input_format, cfg.afc_wide, cfg.droop = input_format, int(afc_wide), int(droop)
        cfg.device_id, cfg.tiles_per_span = device_id, tiles_per_span
        cfg.flags = (FLAG_TAPS if taps else 0) | (FLAG_SERIAL if serial else 0) | (FLAG_DSK if dsk else 0) | (0 if ps_ema else FLAG_PS_BOXCAR) | (FLAG_GPU_DECODE if gpu_decode else 0) | (FLAG_FP_DS if fp_ds else 0) | (FLAG_MODE_X if mode_x else 0) | (FLAG_MA_DS if ma else 0)
        self.cfg = cfg
        self.h = ctypes.c_void_p()
        rc = self.lib.aisgpu_create(ctypes.byref(cfg), ctypes.byref(self.h))
        if rc != 0:
            msg = self.lib.aisgpu_strerror(rc).decode()
            if self.h:
                msg += ": " + self.lib.aisgpu_last_error(self.h).decode()
                self.lib.aisgpu_destroy(self.h)
                self.h = ctypes.c_void_p()
            raise AisGpuError("aisgpu_create failed (%d): %s" % (rc, msg))
        self.n_receivers, self.block_len = n_receivers, block_len

    def _chk(self, rc, what):
        if rc != 0:
            raise AisGpuError("%s failed (%d): %s: %s" % (
                what, rc, self.lib.aisgpu_strerror(rc).decode(), self.lib.aisgpu_last_error(self.h).decode()))

    def submit(self, rx, block):
        block = np.ascontiguousarray(block)
        per = _ELEMS[self.cfg.input_format]
        self._chk(self.lib.aisgpu_submit(self.h, rx, block.ctypes.data, block.size // per), "aisgpu_submit")

    def submit_device(self, ptr, rx_stride_samples):
        self._chk(self.lib.aisgpu_submit_device(self.h, ctypes.c_void_p(ptr), rx_stride_samples), "aisgpu_submit_device")

    def run(self):
        self._chk(self.lib.aisgpu_run(self.h), "aisgpu_run")

    def sync(self):
        self._chk(self.lib.aisgpu_sync(self.h), "aisgpu_sync")

    def sync_outputs(self):
        self._chk(self.lib.aisgpu_sync_outputs(self.h), "aisgpu_sync_outputs")

    def out_count(self):
        return self.lib.aisgpu_out_count(self.h)

    def fetch(self, rx, ch, sub=0):
        """-> dict(bits[5][n_groups] of +-1.0f, lvl[n_groups], ppm[n_windows], first_group, first_sample48)."""
        o = Out()
        self._chk(self.lib.aisgpu_fetch_sub(self.h, sub, rx, ch, ctypes.byref(o)), "aisgpu_fetch_sub")
        n = o.n_groups
        words = (n + 31) // 32
        bits = np.zeros((5, n), np.float32)
        for j in range(5):
            w = np.ctypeslib.as_array(o.bits[j], shape=(max(words, 1),))[:words]
            b = ((w[:, None] >> np.arange(32, dtype=np.uint32)[None, :]) & 1).reshape(-1)[:n]
            bits[j] = np.where(b != 0, 1.0, -1.0)
        lvl = np.ctypeslib.as_array(o.lvl, shape=(max(n, 1),))[:n].copy()
        ppm = np.ctypeslib.as_array(o.ppm, shape=(max(o.n_windows, 1),))[:o.n_windows].copy()
        fm = None
        if o.fm_bits:
            L = o.n_windows * 512
            w = np.ctypeslib.as_array(o.fm_bits, shape=(L // 32,))
            fm = ((w[:, None] >> np.arange(32, dtype=np.uint32)[None, :]) & 1).reshape(-1).astype(np.uint8)
        c48 = None
        if o.c48:
            c48 = np.ctypeslib.as_array(o.c48, shape=(o.n_windows * 1024,)).copy().view(np.complex64)
        v2 = None
        if o.v2_f:
            W = o.n_windows
            v2 = dict(f=np.ctypeslib.as_array(o.v2_f, shape=(2 * W,)).copy(), prom=np.ctypeslib.as_array(o.v2_prom, shape=(2 * W,)).copy(),
                      energy=np.ctypeslib.as_array(o.v2_energy, shape=(W + 1,)).copy())
        return dict(bits=bits, lvl=lvl, ppm=ppm, first_group=o.first_group, first_sample48=o.first_sample48,
                    n_groups=n, n_windows=o.n_windows, fm_bits=fm, c48=c48, v2=v2)

    def frames(self):
        """AISGPU_FLAG_GPU_DECODE: list of dicts, the frames completed since the previous sync_outputs()."""
        fp = ctypes.POINTER(Frame)()
        n = ctypes.c_int()
        self._chk(self.lib.aisgpu_frames(self.h, ctypes.byref(fp), ctypes.byref(n)), "aisgpu_frames")
        return [dict(rx=fp[i].rx, ch=fp[i].ch, phase=fp[i].phase, sub=fp[i].sub, group=fp[i].group, position=fp[i].position,
                     level_sum=fp[i].level_sum, start_idx=fp[i].start_idx, end_idx=fp[i].end_idx, data=bytes(fp[i].data))
                for i in range(n.value)]

    def ps_fallbacks(self):
        """Workgroups (four chains each) of the chunk-parallel PhaseSearchEMA that went through the exact sequential kernel so far."""
        n = ctypes.c_longlong()
        self._chk(self.lib.aisgpu_ps_fallbacks(self.h, ctypes.byref(n)), "aisgpu_ps_fallbacks")
        return n.value

    def decoder_fallbacks(self):
        """Blocks whose frame decoders went through the sequential kernel (candidate lists of the event-driven kernels full)."""
        n = ctypes.c_longlong()
        self._chk(self.lib.aisgpu_decoder_fallbacks(self.h, ctypes.byref(n)), "aisgpu_decoder_fallbacks")
        return n.value

    def tap(self, which, rx=0):
        n = self.lib.aisgpu_tap(self.h, which, rx, None, 0)
        if n < 0:
            raise AisGpuError("aisgpu_tap failed (%d)" % n)
        out = np.zeros(n, np.complex64)
        self.lib.aisgpu_tap(self.h, which, rx, out.ctypes.data, n)
        return out

    def tapf(self, which, rx=0):
        """Real-valued taps 6..9 of the FM receivers (Demod::FM output, Filter(Receiver) output), last downstream block."""
        n = self.lib.aisgpu_tap(self.h, which, rx, None, 0)
        if n < 0:
            raise AisGpuError("aisgpu_tap failed (%d)" % n)
        out = np.zeros(n, np.float32)
        self.lib.aisgpu_tap(self.h, which, rx, out.ctypes.data, n)
        return out

    def timing(self, enable=True):
        self.lib.aisgpu_timing(self.h, int(enable))

    def frontend_ms(self):
        n = ctypes.c_int()
        ms = self.lib.aisgpu_frontend_ms(self.h, ctypes.byref(n))
        return float(ms), n.value

    def stream(self):
        return self.lib.aisgpu_stream(self.h)

    def close(self):
        if self.h:
            self.lib.aisgpu_destroy(self.h)
            self.h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
