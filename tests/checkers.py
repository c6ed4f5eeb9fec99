"""ctypes front ends for the two CHECKERS (test infrastructure, never the product):

  Oracle  -- oracle/libaisoracle.so, our plain-C restatement (prefix ao_)
  Ref     -- oracle/_ref/libaisref_{strict,fast}.so, the reference's own sources compiled in
             place with a C shim (prefix ref_); built only where /root/reference exists, the
             prebuilt .so travels to the GPU box.

Both expose the same surface, so every test can run against either.
"""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "libaisoracle.so"])


def build_ref():
    if os.path.isdir("/root/reference/Source"):
        subprocess.check_call(["make", "-s", "-j8", "-C", ORACLE_DIR, "ref"])


def have_ref(kind="strict"):
    return os.path.exists(os.path.join(ORACLE_DIR, "_ref", "libaisref_%s.so" % kind))


class _Chain:
    """One receiver instance behind either checker library."""

    def __init__(self, lib, prefix, model, rate, fmt, taps, dsk=False, ps_ema=True, fp_ds=False, mode_x=False, ma=False, extra_flags=0,
                 afc_wide=True, droop=True, filename=None):
        self.lib, self.p = lib, prefix
        f = lambda name: getattr(lib, prefix + name)
        f("create").restype = ctypes.c_void_p
        f("create").argtypes = [ctypes.c_int] * 4
        f("feed").argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        f("nmea").argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int]
        f("msg_count").argtypes = [ctypes.c_void_p]
        f("msg_meta").argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        f("tap").restype = ctypes.c_longlong
        f("tap").argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_longlong]
        f("tap_ppm").restype = ctypes.c_longlong
        f("tap_ppm").argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_longlong]
        f("tapf").restype = ctypes.c_longlong
        f("tapf").argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_longlong]
        f("bits").restype = ctypes.c_longlong
        f("bits").argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                              ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong]
        f("set_taps").argtypes = [ctypes.c_void_p, ctypes.c_int]
        f("set_taps").restype = None
        f("destroy").argtypes = [ctypes.c_void_p]
        self._f = f
        self.fmt = fmt
        flags = ((1 if taps else 0) | (2 if dsk else 0) | (0 if ps_ema else 4) | (8 if fp_ds else 0) | (16 if mode_x else 0) | (32 if ma else 0)
                 | (0 if afc_wide else 256) | (0 if droop else 512) | extra_flags)  # bits 8 / 9: `-go AFC_WIDE off` / `-go DROOP off`
        fmt_id = {"cu8": 0, "cf32": 1, "cs8": 2, "cs16": 3}[fmt]
        if filename is not None:  # Ref / RefGpu only: the model on the reference's own Device::RAWFile (play_file() runs it)
            f("create_file").restype = ctypes.c_void_p
            f("create_file").argtypes = [ctypes.c_int] * 4 + [ctypes.c_char_p]
            f("play_file").argtypes = [ctypes.c_void_p, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]
            self.h = f("create_file")(model, rate, fmt_id, flags, os.fsencode(filename))
        else:
            self.h = f("create")(model, rate, fmt_id, flags)
        if not self.h:
            raise RuntimeError("checker create failed")

    def feed(self, block):
        block = np.ascontiguousarray(block)
        return self._f("feed")(self.h, block.ctypes.data, block.nbytes)

    def play_file(self, timeout_s=120.0):
        """Runs the whole file through the model on the reference's own reader / run threads (Device/FileRAW.cpp); returns
        (status, number of Receive() calls the device made, largest number of FIFO blocks in one call)."""
        calls, maxb = ctypes.c_int(0), ctypes.c_int(0)
        rc = self._f("play_file")(self.h, timeout_s, ctypes.byref(calls), ctypes.byref(maxb))
        return rc, calls.value, maxb.value

    def feed_blocks(self, x, block_len):
        """Feed whole blocks of block_len IQ samples (the tail that does not fill a block is dropped)."""
        per = 1 if self.fmt == "cf32" else 2
        n = (len(x) // per) // block_len
        for b in range(n):
            self.feed(x[b * block_len * per:(b + 1) * block_len * per])
        return n

    def nmea(self):
        n = self._f("nmea")(self.h, None, 0)
        buf = ctypes.create_string_buffer(n)
        self._f("nmea")(self.h, buf, n)
        return buf.value.decode().splitlines()

    def msg_meta(self):
        n = self._f("msg_count")(self.h)
        lvl = np.zeros(max(n, 1), np.float32)
        ppm = np.zeros(max(n, 1), np.float32)
        self._f("msg_meta")(self.h, lvl.ctypes.data, ppm.ctypes.data, n)
        return lvl[:n], ppm[:n]

    def tap(self, which):
        n = self._f("tap")(self.h, which, None, 0)
        out = np.zeros(n, np.complex64)
        self._f("tap")(self.h, which, out.ctypes.data, n)
        return out

    def tapf(self, which):
        """Real taps of the FM receivers: 6/7 = Demod::FM output A/B, 8/9 = Filter(Receiver) output A/B."""
        n = self._f("tapf")(self.h, which, None, 0)
        out = np.zeros(n, np.float32)
        self._f("tapf")(self.h, which, out.ctypes.data, n)
        return out

    def set_taps(self, on):
        """Recording of taps / decisions on or off from the next feed() on (chain created with taps=True)."""
        self._f("set_taps")(self.h, 1 if on else 0)

    def tap_ppm(self, which):
        n = self._f("tap_ppm")(self.h, which, None, 0)
        out = np.zeros(n, np.float32)
        self._f("tap_ppm")(self.h, which, out.ctypes.data, n)
        return out

    def bits(self, ch, j, fm=0):
        n = self._f("bits")(self.h, ch, j, fm, None, None, None, 0)
        b = np.zeros(n, np.float32)
        l = np.zeros(n, np.float32)
        i = np.zeros(n, np.int64)
        self._f("bits")(self.h, ch, j, fm, b.ctypes.data, l.ctypes.data, i.ctypes.data, n)
        return b, l, i

    def close(self):
        if self.h:
            self._f("destroy")(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_libs = {}


def _lib(path):
    if path not in _libs:
        _libs[path] = ctypes.CDLL(path)
    return _libs[path]


def Oracle(model=2, rate=1536000, fmt="cf32", taps=False, dsk=False, ps_ema=True, fp_ds=False, mode_x=False, ma=False, afc_wide=True, droop=True):
    path = os.path.join(ORACLE_DIR, "libaisoracle.so")
    if not os.path.exists(path):
        build_oracle()
    lib = _lib(path)
    lib.ao_reset_seq()
    return _Chain(lib, "ao_", model, rate, fmt, taps, dsk, ps_ema, fp_ds, mode_x, ma, afc_wide=afc_wide, droop=droop)


def Ref(model=2, rate=1536000, fmt="cf32", taps=False, kind="strict", dsk=False, ps_ema=True, fp_ds=False, mode_x=False, ma=False, afc_wide=True, droop=True,
        filename=None):
    lib = _lib(os.path.join(ORACLE_DIR, "_ref", "libaisref_%s.so" % kind))
    lib.ref_reset_seq()
    return _Chain(lib, "ref_", model, rate, fmt, taps, dsk, ps_ema, fp_ds, mode_x, ma, afc_wide=afc_wide, droop=droop, filename=filename)


def have_refgpu():
    return os.path.exists(os.path.join(ORACLE_DIR, "_ref", "libaisrefgpu.so"))


def refgpu_block_bytes(n):
    """Bytes per GPU block for the GPU engines built from now on (GpuPool::setBlockBytes); 0 = by the device."""
    lib = _lib(os.path.join(ORACLE_DIR, "_ref", "libaisrefgpu.so"))
    lib.ref_gpu_block_bytes.argtypes = [ctypes.c_int]
    lib.ref_gpu_block_bytes.restype = None
    lib.ref_gpu_block_bytes(n)


def RefGpu(model=12, rate=1536000, fmt="cf32", dsk=False, ps_ema=True, fp_ds=False, ma=False, gpu_decode=False, pipelined=False, afc_wide=True, droop=True,
           filename=None, mode_x=False):
    """oracle/_ref/libaisrefgpu.so: the reference's unmodified sources PLUS the reference-side binding of libaisgpu.so
    (integration/reference/Source/DSP/GPU/ModelGPU.cpp, an AIS::Model subclass compiled against the reference's real headers).
    model 2 / 4 / 0 / 1 = the reference's own ModelDefault / ModelChallenger / ModelStandard / ModelBase, 12 / 14 / 20 / 21 = the same
    engines with the DSP on the GPU.  GPU engines created with the same configuration before their first block share one GPU context
    (feed them from one thread each); gpu_decode: decoder state machines on the device; pipelined: call flush() after the last block."""
    lib = _lib(os.path.join(ORACLE_DIR, "_ref", "libaisrefgpu.so"))
    lib.ref_reset_seq()
    c = _Chain(lib, "ref_", model, rate, fmt, False, dsk, ps_ema, fp_ds, mode_x, ma, extra_flags=(64 if gpu_decode else 0) | (128 if pipelined else 0),
               afc_wide=afc_wide, droop=droop, filename=filename)
    lib.ref_flush.argtypes = [ctypes.c_void_p]
    lib.ref_flush.restype = None
    c.flush = lambda: lib.ref_flush(c.h)
    return c


def oracle_lib():
    path = os.path.join(ORACLE_DIR, "libaisoracle.so")
    if not os.path.exists(path):
        build_oracle()
    return _lib(path)
