"""Host helpers with the reference's names and signatures (wesep/utils/funcs.py):

  * `clip_gradients`  funcs.py:79-88   (per-tensor L2 clip; the fused multi-tensor path lives in optim.py)
  * `compute_fbank`   funcs.py:91-116  batch kaldi filterbank of device waveforms [R, T] -> [R, frames, num_mel_bins]
  * `apply_cmvn`      funcs.py:119-140 per-row mean (and optionally variance) normalisation over frames

`compute_fbank` / `apply_cmvn` are what the SSA self-enrollment pass (executor.py:89-100) puts on the training
critical path: the reference loops over rows on the host and calls
`torchaudio.compliance.kaldi.fbank` once per row.  Here the whole batch is four launches on the device, by
composition of existing entry points:

  1. every per-frame step of the kaldi front-end before the power spectrum is linear in the frame -- int16 scaling,
     DC removal (I - 11^T/n), pre-emphasis 0.97 with the first sample replicated, the symmetric Hamming window, zero
     padding to 512 and the real DFT -- so they are folded on the host (float64) into ONE [2*256, win] basis, and the
     spectrum of all frames of all rows is one exact-fp32 MFMA GEMM (`ws_gemm_nt`) whose A operand is the waveform
     itself read through an overlapping row view (frame f of row r starts at r*T + f*shift; nothing is unfolded);
     dither, which kaldi adds per frame element, goes through the same basis as a second GEMM on N(0, dither^2)
     noise whose result enters the first one as the residual operand;
  2. `ws_power_spec` -> |X|^2 of bins 0..255 (the Nyquist bin has zero weight in every kaldi mel filter);
  3. the triangular mel bank as a second GEMM, then log(max(x, eps)) as `ws_prelu_fwd` (slope 0, bias -eps) +
     `ws_log_eps`;
  4. CMN: `ws_chan_sums` over the frames of each row + `ws_affine_fwd`.

Deterministic part pinned through oracle/fbank_oracle.py against the reference's own C++ kaldi front-end
(runtime/frontend/fbank.h); the dither uses torch's device generator (the reference draws from torch's global CPU
generator -- stochastic on both sides)."""
import math

import torch

from .. import dev
from .._lib import WesepHipError
from ..dev import Rows, flat
from ..functional import _empty
from ..optim import clip_gradients  # noqa: F401

FLT_EPS = 1.1920928955078125e-07   # torch.finfo(torch.float32).eps: kaldi's log floor
_TABLES = {}


def _mel(f):
    return 1127.0 * math.log(1.0 + f / 700.0)


def _fbank_tables(num_mel_bins, win, padded, sample_rate, device):
    """(basis [2*(padded/2), win], mel^T [num_mel_bins, padded/2]) on `device`, cached."""
    key = (num_mel_bins, win, padded, sample_rate, device.type, device.index)
    if key in _TABLES:
        return _TABLES[key]
    nf = padded // 2
    n = torch.arange(win, dtype=torch.float64)
    window = 0.54 - 0.46 * torch.cos(2.0 * math.pi * n / (win - 1))
    ang = 2.0 * math.pi * torch.arange(nf, dtype=torch.float64).unsqueeze(1) * n.unsqueeze(0) / padded
    dft = torch.empty(2 * nf, win, dtype=torch.float64)
    dft[0::2] = torch.cos(ang)
    dft[1::2] = -torch.sin(ang)
    pre = torch.eye(win, dtype=torch.float64)                 # y[i] = x[i] - 0.97 x[max(i-1, 0)]
    pre[0, 0] -= 0.97
    idx = torch.arange(1, win)
    pre[idx, idx - 1] = -0.97
    dc = torch.eye(win, dtype=torch.float64) - 1.0 / win      # x - mean(x)
    basis = (dft * window.unsqueeze(0)) @ pre @ dc * float(1 << 15)
    # triangular filters, linear on the mel scale between 20 Hz and Nyquist (kaldi get_mel_banks, no VTLN)
    lo, hi = _mel(20.0), _mel(0.5 * sample_rate)
    delta = (hi - lo) / (num_mel_bins + 1)
    b = torch.arange(num_mel_bins, dtype=torch.float64).unsqueeze(1)
    left, center, right = lo + b * delta, lo + (b + 1) * delta, lo + (b + 2) * delta
    mel = 1127.0 * torch.log(1.0 + (sample_rate / padded) * torch.arange(nf, dtype=torch.float64) / 700.0)
    mel = mel.unsqueeze(0)
    bank = torch.clamp(torch.minimum((mel - left) / (center - left), (right - mel) / (right - center)), min=0.0)
    _TABLES[key] = (basis.float().to(device).contiguous(), bank.float().to(device).contiguous())
    return _TABLES[key]


@torch.no_grad()
def compute_fbank(data, num_mel_bins=80, frame_length=25, frame_shift=10, dither=1.0, sample_rate=16000):
    """Extract fbank: data [R, T] in [-1, 1] on the device -> [R, 1 + (T - win)//shift, num_mel_bins]."""
    if not data.is_cuda:
        raise WesepHipError("compute_fbank: wesep_amd has no CPU path")
    if data.dim() != 2:
        raise ValueError(f"compute_fbank: expected [rows, samples], got {tuple(data.shape)}")
    if num_mel_bins % 4:
        raise NotImplementedError("compute_fbank: num_mel_bins must be a multiple of 4")
    data = data.detach().float().contiguous()
    R, T = data.shape
    win = int(sample_rate * frame_length * 0.001)
    shift = int(sample_rate * frame_shift * 0.001)
    if T < win:
        raise ValueError(f"compute_fbank: {T} samples are shorter than one {win}-sample frame")
    padded = 1 << (win - 1).bit_length()
    nf = padded // 2
    Tf = 1 + (T - win) // shift
    M = R * Tf
    d = data.device
    basis, bank = _fbank_tables(num_mel_bins, win, padded, sample_rate, d)
    noise_spec = None
    if dither != 0.0:
        # basis carries the 2^15 waveform scaling; the dither is added after that scaling
        noise = torch.randn(M, win, device=d, dtype=torch.float32) * (float(dither) / float(1 << 15))
        noise_spec = _empty(d, M, 2 * nf)
        dev.gemm_nt(A=noise, a_rows=flat(win), M=M, N=2 * nf, K=win, W=basis, ldw=win, C_out=noise_spec,
                    c_rows=flat(2 * nf), vec=3 if win % 4 == 0 else 0, mode="f32")
    spec = _empty(d, M, 2 * nf)
    aligned = T % 4 == 0 and shift % 4 == 0 and win % 4 == 0
    dev.gemm_nt(A=data, a_rows=Rows(Tf, T, shift), M=M, N=2 * nf, K=win, W=basis, ldw=win, C_out=spec,
                c_rows=flat(2 * nf), R=noise_spec, vec=(3 if aligned else (2 if win % 4 == 0 else 0)), mode="f32")
    power = _empty(d, M, nf)
    dev.power_spec(spec, M, nf, 2 * nf, nf, power)
    feats = _empty(d, M, num_mel_bins)
    dev.gemm_nt(A=power, a_rows=flat(nf), M=M, N=num_mel_bins, K=nf, W=bank, ldw=nf, C_out=feats,
                c_rows=flat(num_mel_bins), vec=3, mode="f32")
    # log(max(x, eps)) = log(relu(x - eps) + eps)
    floor = torch.full((1, num_mel_bins), -FLT_EPS, device=d, dtype=torch.float32)
    slope = torch.zeros(1, device=d, dtype=torch.float32)
    out = _empty(d, M, num_mel_bins)
    dev.prelu_fwd(feats, floor, slope, M, num_mel_bins, M, out)
    dev.log_eps(out, FLT_EPS)
    return out.view(R, Tf, num_mel_bins)


@torch.no_grad()
def apply_cmvn(data, norm_mean=True, norm_var=False):
    """Apply CMVN: data [R, frames, D] on the device; each row minus its mean over frames."""
    if not data.is_cuda:
        raise WesepHipError("apply_cmvn: wesep_amd has no CPU path")
    if norm_var:
        raise NotImplementedError("apply_cmvn(norm_var=True): no call site in the reference enables it")
    if not norm_mean:
        return data
    data = data.detach().float().contiguous()
    R, Tf, D = data.shape
    neg_mean = dev.chan_sums(data, None, None, 1, Tf, R, D)[:, 0, :].contiguous()
    dev.affine_fwd(neg_mean, None, None, -1.0 / Tf, R, 1, D, neg_mean)
    out = _empty(data.device, R, Tf, D)
    dev.affine_fwd(data, None, neg_mean, 1.0, R * Tf, Tf, D, out)
    return out
