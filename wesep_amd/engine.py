"""ctypes binding of the native runtime (include/wesep_engine.h, runtime/libwesep_engine.so): what a Python caller
(tests, `wesep_amd.bin.infer --engine`) uses; C++ callers link the library directly (runtime/separate_main.cc)."""
import ctypes as C
import os

import numpy as np

from ._lib import WesepHipError

ENGINE_ABI_VERSION = 1
DRY_RUN = 1
ENROLL_EMBEDDING, ENROLL_FBANK, ENROLL_WAVE = 0, 1, 2
LIB_PATH = os.environ.get("WESEP_ENGINE_LIB") or os.path.join(
    os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "runtime", "libwesep_engine.so")
SYMBOLS = ("ws_engine_abi_version", "ws_engine_last_error", "ws_engine_create", "ws_engine_destroy", "ws_engine_info",
           "ws_engine_separate", "ws_engine_forward_pcm16")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise WesepHipError(f"{LIB_PATH} is missing: run `python -m wesep_amd.build` (needs hipcc)")
        l = C.CDLL(LIB_PATH)
        l.ws_engine_abi_version.restype = C.c_int
        l.ws_engine_last_error.restype = C.c_char_p
        l.ws_engine_create.restype = C.c_int
        l.ws_engine_create.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        l.ws_engine_destroy.restype = None
        l.ws_engine_destroy.argtypes = [C.c_void_p]
        l.ws_engine_info.restype = C.c_longlong
        l.ws_engine_info.argtypes = [C.c_void_p, C.c_char_p]
        l.ws_engine_separate.restype = C.c_int
        l.ws_engine_separate.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                         C.c_void_p]
        l.ws_engine_forward_pcm16.restype = C.c_int
        l.ws_engine_forward_pcm16.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                              C.c_void_p]
        if l.ws_engine_abi_version() != ENGINE_ABI_VERSION:
            raise WesepHipError("libwesep_engine.so ABI version mismatch; rebuild")
        _lib = l
    return _lib


def _check(rc, what):
    if rc != 0:
        raise WesepHipError(f"{what} failed (rc={rc}): {lib().ws_engine_last_error().decode('utf-8', 'replace')}")


def _quiesce_torch():
    """The engine launches on its own HIP stream.  Kernels of this library running concurrently on ANOTHER stream
    disturb FFT-type kernels (profiles/r02_kernel_race.md: engines sharing a GPU therefore take turns inside
    libwesep_engine.so); a Python process that also drives torch work on the same GPU gets the same guarantee here:
    whatever torch has in flight finishes before the engine starts."""
    import sys
    torch = sys.modules.get("torch")
    if torch is not None and torch.cuda.is_available() and torch.cuda.is_initialized():
        torch.cuda.synchronize()


class Engine:
    """One loaded model on one GPU.  `dry_run=True` needs no GPU: validates the container and every launch's
    argument contract, computes nothing."""

    def __init__(self, weights_path, device=0, dry_run=False):
        self._h = C.c_void_p()
        _check(lib().ws_engine_create(os.fsencode(weights_path), device, DRY_RUN if dry_run else 0, C.byref(self._h)),
               "ws_engine_create")

    def close(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.ws_engine_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:       # interpreter shutdown
            pass

    def info(self, key):
        return int(lib().ws_engine_info(self._h, key.encode()))

    def separate(self, mix, enroll, kind):
        """mix [R, T] float32; enroll: [R, E] (ENROLL_EMBEDDING), [R, Te, F] (ENROLL_FBANK) or [R, Tw]
        (ENROLL_WAVE) -> est [R, T] float32 (numpy, host)."""
        mix = np.ascontiguousarray(mix, dtype=np.float32)
        enroll = np.ascontiguousarray(enroll, dtype=np.float32)
        R, T = mix.shape
        if enroll.shape[0] != R:
            raise ValueError("one enrollment per mixture row")
        est = np.zeros((R, T), dtype=np.float32)
        length = 0 if kind == ENROLL_EMBEDDING else enroll.shape[1]
        _quiesce_torch()
        _check(lib().ws_engine_separate(self._h, mix.ctypes.data, R, T, enroll.ctypes.data, kind, length,
                                        est.ctypes.data), "ws_engine_separate")
        return est

    def forward_pcm16(self, mix, spk1, spk2):
        """int16 [n], int16 [n_enroll] x 2 -> float32 [2, n] in [-1, 1] (SeparateEngine::ForwardFunc)."""
        mix, spk1, spk2 = (np.ascontiguousarray(x, dtype=np.int16) for x in (mix, spk1, spk2))
        n_enroll = min(spk1.shape[0], spk2.shape[0])
        out = np.zeros((2, mix.shape[0]), dtype=np.float32)
        _quiesce_torch()
        _check(lib().ws_engine_forward_pcm16(self._h, mix.ctypes.data, mix.shape[0], spk1.ctypes.data,
                                             spk2.ctypes.data, n_enroll, out.ctypes.data), "ws_engine_forward_pcm16")
        return out
