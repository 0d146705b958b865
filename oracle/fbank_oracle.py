"""TEST INFRASTRUCTURE -- not product code.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module.

CPU restatement of the enrollment filterbank that the SSA second pass puts on the training critical path
(SURVEY.md section 8 row f-2):

  * `compute_fbank`  wesep/utils/funcs.py:91-116  -- per row: waveform * 2^15, then
    `torchaudio.compliance.kaldi.fbank(num_mel_bins, frame_length, frame_shift, dither, sample_frequency,
    window_type="hamming", use_energy=False)` (third-party, absent from this image; the same options are used by the
    datapipe, wesep/dataset/processor.py:480-511);
  * `apply_cmvn`     wesep/utils/funcs.py:119-140 -- per row: minus the mean over frames, optionally divided by
    sqrt(unbiased var + 1e-8).

The kaldi algorithm restated here (snip-edges framing, per-frame dither, DC removal, 0.97 pre-emphasis with the first
sample replicated, symmetric Hamming window, zero padding to the next power of two, power spectrum, triangular
filters that are linear on the 1127*ln(1+f/700) mel scale between 20 Hz and Nyquist, log floored at float epsilon) is
PINNED against the reference's own C++ implementation of the same front-end, `wenet::Fbank`
(runtime/frontend/fbank.h:31-222, fft.cc), compiled from the reference sources by oracle/build_ref.py into
oracle/_ref/libref_fbank.so: tests/test_oracle_golden.py compares both on seeded waveforms, and the committed
fixtures tests/golden/fbank_*.npz hold the outputs of that C++ code (dither 0; the dithered path is stochastic in
the reference too -- funcs.py passes dither=1.0 and torchaudio draws from the global torch RNG -- so it is checked
through its statistics only)."""
import ctypes
import math
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_LIB = os.path.join(HERE, "_ref", "libref_fbank.so")
FLT_EPS = float(np.finfo(np.float32).eps)


def mel_scale(f):
    return 1127.0 * np.log(1.0 + np.asarray(f, dtype=np.float64) / 700.0)


def mel_banks(num_bins, padded, sample_rate, low_freq=20.0, high_freq=0.0):
    """[num_bins, padded/2 + 1] triangular weights (kaldi `get_mel_banks` without VTLN; fbank.h:52-89).  The last
    column (Nyquist) is zero, as in torchaudio's right-padding of the bank matrix."""
    nfft = padded // 2
    nyq = 0.5 * sample_rate
    if high_freq <= 0.0:
        high_freq += nyq
    width = sample_rate / padded
    ml, mh = mel_scale(low_freq), mel_scale(high_freq)
    delta = (mh - ml) / (num_bins + 1)
    b = np.arange(num_bins, dtype=np.float64)[:, None]
    left, center, right = ml + b * delta, ml + (b + 1) * delta, ml + (b + 2) * delta
    mel = mel_scale(width * np.arange(nfft))[None, :]
    up = (mel - left) / (center - left)
    down = (right - mel) / (right - center)
    out = np.zeros((num_bins, nfft + 1))
    out[:, :nfft] = np.maximum(0.0, np.minimum(up, down))
    return out


def frame_count(num_samples, win, shift):
    return 0 if num_samples < win else 1 + (num_samples - win) // shift


def kaldi_fbank(wave, num_mel_bins=80, frame_length=25, frame_shift=10, dither=0.0, sample_rate=16000, rng=None,
                dtype=np.float64):
    """wave [T] (already scaled to the int16 range) -> [frames, num_mel_bins] log-mel energies."""
    win = int(sample_rate * frame_length * 0.001)
    shift = int(sample_rate * frame_shift * 0.001)
    padded = 1 << (win - 1).bit_length()
    wave = np.asarray(wave, dtype=dtype)
    m = frame_count(wave.shape[0], win, shift)
    if m == 0:
        return np.zeros((0, num_mel_bins), dtype=dtype)
    idx = shift * np.arange(m)[:, None] + np.arange(win)[None, :]
    x = wave[idx]
    if dither != 0.0:
        rng = rng or np.random.default_rng(0)
        x = x + dither * rng.standard_normal(x.shape).astype(dtype)
    x = x - x.mean(axis=1, keepdims=True)
    prev = np.concatenate([x[:, :1], x[:, :-1]], axis=1)
    x = x - dtype(0.97) * prev
    n = np.arange(win, dtype=np.float64)
    window = (0.54 - 0.46 * np.cos(2.0 * math.pi * n / (win - 1))).astype(dtype)
    x = x * window
    spec = np.fft.rfft(x.astype(np.float64), n=padded, axis=1)
    power = (spec.real ** 2 + spec.imag ** 2).astype(dtype)
    mel = power @ mel_banks(num_mel_bins, padded, sample_rate).astype(dtype).T
    return np.log(np.maximum(mel, dtype(FLT_EPS)))


def compute_fbank(data, num_mel_bins=80, frame_length=25, frame_shift=10, dither=1.0, sample_rate=16000, rng=None,
                  dtype=np.float64):
    """funcs.py:91-116: data [R, T] in [-1, 1] -> [R, frames, num_mel_bins]."""
    data = np.asarray(data)
    rng = rng or np.random.default_rng(0)
    return np.stack([kaldi_fbank(row.astype(dtype) * dtype(1 << 15), num_mel_bins, frame_length, frame_shift, dither,
                                 sample_rate, rng, dtype) for row in data], 0)


def apply_cmvn(data, norm_mean=True, norm_var=False):
    """funcs.py:119-140: data [R, frames, D]."""
    out = []
    for mat in np.asarray(data):
        if norm_mean:
            mat = mat - mat.mean(axis=0)
        if norm_var:
            mat = mat / np.sqrt(mat.var(axis=0, ddof=1) + 1e-8)
        out.append(mat)
    return np.stack(out, 0)


# ---- the compiled reference (oracle/_ref) ------------------------------------------------------------------
_ref = None


def ref_available():
    return os.path.exists(REF_LIB)


def ref_fbank(wave_i16_scale, num_mel_bins=80, sample_rate=16000, frame_length=25, frame_shift=10, apply_mean=False):
    """`wenet::Fbank::Compute` (+ `SeparateEngine::ApplyMean`) of the reference runtime on one waveform."""
    global _ref
    if _ref is None:
        _ref = ctypes.CDLL(REF_LIB)
        _ref.ref_fbank.restype = ctypes.c_int
        _ref.ref_fbank.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                   ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    w = np.ascontiguousarray(wave_i16_scale, dtype=np.float32)
    win = int(sample_rate * frame_length * 0.001)
    shift = int(sample_rate * frame_shift * 0.001)
    m = frame_count(w.shape[0], win, shift)
    out = np.zeros((max(m, 1), num_mel_bins), dtype=np.float32)
    n = _ref.ref_fbank(w.ctypes.data, w.shape[0], num_mel_bins, sample_rate, win, shift, int(apply_mean),
                       out.ctypes.data, max(m, 1))
    assert n == m, (n, m)
    return out[:m]
