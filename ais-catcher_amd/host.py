"""ctypes binding of libaishost.so: the C++ host side (GpuChain / ModelDefaultGPU / frame decoder).

ModelDefaultGPU mirrors the reference's AIS::Model contract (DSP/Model.h:76-126): one instance per
receiver, fed one RAW block per call like the reference's device thread does, NMEA out.
"""
import ctypes
import os

import numpy as np

from . import gpu as _gpu

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libaishost.so")
_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    _gpu.load()
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libaishost.so is not built; run __graft_entry__.build()")
    lib = ctypes.CDLL(LIB_PATH)
    vp, ci, cll = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong
    lib.aishost_batch_create.restype = vp
    lib.aishost_batch_create.argtypes = [ctypes.POINTER(_gpu.Cfg), ctypes.c_char_p, ci]
    lib.aishost_batch_destroy.argtypes = [vp]
    lib.aishost_model_create.restype = vp
    lib.aishost_model_create.argtypes = [vp, ci, ci, ci, ci, ctypes.c_char, ctypes.c_char, ci, ci, ctypes.c_char_p, ci]
    lib.aishost_model_destroy.argtypes = [vp]
    lib.aishost_model_receive.argtypes = [vp, vp, ci]
    lib.aishost_model_leave.argtypes = [vp]
    lib.aishost_model_leave.restype = None
    lib.aishost_batch_set_timeout.argtypes = [vp, ci]
    lib.aishost_batch_set_timeout.restype = None
    lib.aishost_batch_active.argtypes = [vp]
    lib.aishost_batch_set_pipelined.argtypes = [vp, ci]
    lib.aishost_batch_set_pipelined.restype = None
    lib.aishost_model_flush.argtypes = [vp]
    lib.aishost_model_flush.restype = None
    lib.aishost_model_replay.argtypes = [vp, ci, cll, cll, ci, ctypes.POINTER(vp), vp, ci, vp, vp]
    lib.aishost_model_feed48.argtypes = [vp, ci, vp, ci]
    if hasattr(lib, "aishost_model_frame"):
        lib.aishost_model_frame.argtypes = [vp, vp]
    lib.aishost_model_msg_count.argtypes = [vp]
    lib.aishost_model_nmea.argtypes = [vp, ctypes.c_char_p, ci]
    lib.aishost_model_msg_meta.argtypes = [vp, vp, vp, ci]
    _lib = lib
    return lib


class Batch:
    """Shared GPU context for n_receivers ModelDefaultGPU instances (one per receiver thread)."""

    def __init__(self, sample_rate=1536000, n_receivers=1, block_len=786432, input_format=_gpu.FMT_CF32, device_id=0,
                 model=_gpu.MODEL_DEFAULT, gpu_decode=False):
        lib = load()
        cfg = _gpu.Cfg()
        _gpu.load().aisgpu_default_cfg(ctypes.byref(cfg))
        cfg.sample_rate, cfg.n_receivers, cfg.block_len = sample_rate, n_receivers, block_len
        cfg.input_format, cfg.device_id, cfg.model = input_format, device_id, model
        if gpu_decode:
            cfg.flags |= _gpu.FLAG_GPU_DECODE
        err = ctypes.create_string_buffer(512)
        _gpu.apply_env_options()  # (test hooks: AISGPU_<KEY> of the environment -> aisgpu_set_option; the library itself reads none)
        self.h = lib.aishost_batch_create(ctypes.byref(cfg), err, 512)
        if not self.h:
            raise RuntimeError(err.value.decode())
        self.cfg = cfg

    def set_timeout(self, ms):
        """A receiver that has not delivered `ms` after the first one of a block is evicted (<= 0: wait for ever)."""
        load().aishost_batch_set_timeout(self.h, ms)

    def active(self):
        return load().aishost_batch_active(self.h)

    def set_pipelined(self, on=True):
        """Hand-off one block deep: receive() of block f returns once f has been started on the device, having decoded block f-1."""
        load().aishost_batch_set_pipelined(self.h, int(on))

    def close(self):
        if self.h:
            load().aishost_batch_destroy(self.h)
            self.h = None


class ModelDefaultGPU:
    def __init__(self, sample_rate=1536000, block_len=786432, input_format=_gpu.FMT_CF32, ch1="A", ch2="B",
                 batch=None, rx=0, detached=False, model=_gpu.MODEL_DEFAULT, gpu_decode=False, fp_ds=False, mode_x=False, ma=False):
        self.lib = load()
        err = ctypes.create_string_buffer(512)
        self.fmt = input_format
        _gpu.apply_env_options()
        self.h = self.lib.aishost_model_create(batch.h if batch else None, rx, sample_rate, block_len, input_format,
                                               ch1.encode(), ch2.encode(), 1 if detached else 0, model | (0x100 if gpu_decode else 0) | (0x200 if fp_ds else 0) | (0x400 if mode_x else 0) | (0x800 if ma else 0), err, 512)
        if not self.h:
            raise RuntimeError(err.value.decode())

    def receive(self, block):
        """One device block; returns the AISGPU_* status (0 = ok; e.g. 4 once the batch has evicted this receiver)."""
        block = np.ascontiguousarray(block)
        return self.lib.aishost_model_receive(self.h, block.ctypes.data, block.nbytes)

    def flush(self):
        """After the last block of a pipelined batch: collect and decode the last block's outputs."""
        self.lib.aishost_model_flush(self.h)

    def leave(self):
        """End of this receiver's input: the batch it shares stops waiting for it."""
        self.lib.aishost_model_leave(self.h)

    def replay(self, ch, first_group, first_sample48, bits5, lvl, ppm, fm=None):
        """bits5: [5][n_groups] array of +-1 (or 0/1) decisions; packed here like the GPU packs them.
        fm: ModelChallenger only, [512 * len(ppm)] FM-branch decisions (> 0) of the block's 48 kHz samples."""
        n = bits5.shape[1]
        words = (n + 31) // 32
        packed = []
        for j in range(5):
            b = np.zeros(words * 32, np.uint32)
            b[:n] = (np.asarray(bits5[j]) > 0)
            w = (b.reshape(words, 32) << np.arange(32, dtype=np.uint32)[None, :]).sum(axis=1, dtype=np.uint64).astype(np.uint32)
            packed.append(np.ascontiguousarray(w))
        ptrs = (ctypes.c_void_p * 5)(*[p.ctypes.data for p in packed])
        lvl = np.ascontiguousarray(lvl, np.float32)
        ppm = np.ascontiguousarray(ppm, np.float32)
        fmw = None
        if fm is not None:
            fb = (np.asarray(fm) > 0).astype(np.uint32).reshape(-1, 32)
            fmw = np.ascontiguousarray((fb << np.arange(32, dtype=np.uint32)[None, :]).sum(axis=1, dtype=np.uint64).astype(np.uint32))
        self.lib.aishost_model_replay(self.h, ch, first_group, first_sample48, n, ptrs, lvl.ctypes.data, len(ppm), ppm.ctypes.data,
                                      fmw.ctypes.data if fmw is not None else None)

    def frame(self, f):
        """One frame of the device decoders (a dict of gpu.AisGpu.frames()) to the tail of its decoder (validation, NMEA text)."""
        fr = _gpu.Frame()
        for k in ("rx", "ch", "phase", "sub", "group", "position", "level_sum", "start_idx", "end_idx"):
            setattr(fr, k, f[k])
        ctypes.memmove(fr.data, f["data"], min(len(f["data"]), 144))
        self.lib.aishost_model_frame(self.h, ctypes.byref(fr))

    def feed48(self, ch, iq):
        """ModelEngineV2 host logic (detached or not): one block of the 48 kHz channel ch as complex64."""
        iq = np.ascontiguousarray(iq, dtype=np.complex64)
        self.lib.aishost_model_feed48(self.h, ch, iq.ctypes.data, len(iq))

    def nmea(self):
        n = self.lib.aishost_model_nmea(self.h, None, 0)
        buf = ctypes.create_string_buffer(n)
        self.lib.aishost_model_nmea(self.h, buf, n)
        return buf.value.decode().splitlines()

    def msg_meta(self):
        n = self.lib.aishost_model_msg_count(self.h)
        lvl = np.zeros(max(n, 1), np.float32)
        ppm = np.zeros(max(n, 1), np.float32)
        self.lib.aishost_model_msg_meta(self.h, lvl.ctypes.data, ppm.ctypes.data, n)
        return lvl[:n], ppm[:n]

    def close(self):
        if self.h:
            self.lib.aishost_model_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ModelChallengerGPU(ModelDefaultGPU):
    def __init__(self, **kw):
        kw["model"] = _gpu.MODEL_CHALLENGER
        super().__init__(**kw)


class ModelEngineV2GPU(ModelDefaultGPU):
    """AIS::ModelEngineV2 (-m 11): GPU front end; V2::Engine (whose every block depends on its decoders' state) on the host."""

    def __init__(self, **kw):
        kw["model"] = _gpu.MODEL_V2
        super().__init__(**kw)


class ModelStandardGPU(ModelDefaultGPU):
    """AIS::ModelStandard (-m 0): GPU front end + FM discriminator + filter; Deinterleave(5) and five decoders on the host."""

    def __init__(self, **kw):
        kw["model"] = _gpu.MODEL_STANDARD
        super().__init__(**kw)


class ModelBaseGPU(ModelDefaultGPU):
    """AIS::ModelBase (-m 1): GPU front end + FM discriminator + filter; SimplePLL and its decoder feedback on the host."""

    def __init__(self, **kw):
        kw["model"] = _gpu.MODEL_BASE
        super().__init__(**kw)


def reset_sequence():
    load().aishost_reset_sequence()
