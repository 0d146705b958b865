"""In-model enrollment front-end of the joint-training path when the datapipe delivers raw enrollment audio
(`spk_feat=False`, `feat_type="consistent"`; wesep/models/bsrnn.py:231-242,343-350):

    PreEmphasis(0.97) -> MelSpectrogram(sr, n_fft = win_length = 512, hop 128, f_min 20, hamming, n_mels) -> +1e-8 ->
    log -> minus the mean over time -> [R, Tf, n_mels]          (all under no_grad in the reference)

`PreEmphasis` is reference code (wesep/modules/common/speaker.py:10-23); `MelSpectrogram` is
`torchaudio.transforms.MelSpectrogram` -- third-party, absent here, restated from its documented defaults
(centre = True, reflect padding, power 2, HTK mel scale, no filterbank normalisation, f_max = sr/2): parity for that
half is UNPINNED.  Buffers carry the upstream names (`flipped_filter`; `spectrogram.window`, `mel_scale.fb`) so
reference state_dicts load.  On the device: one framing kernel, the windowed DFT and the mel projection as two
exact-fp32 MFMA GEMMs on row views, a power kernel, log and the time mean (csrc/conv2d.hip, tasnet.hip)."""
import math

import torch
import torch.nn as nn

from ... import dev
from ...dev import Rows, flat
from ...functional import _empty


def melscale_fbanks(n_freqs, f_min, f_max, n_mels, sample_rate):
    """torchaudio.functional.melscale_fbanks(norm=None, mel_scale='htk') -> [n_freqs, n_mels]."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * math.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * math.log10(1.0 + f_max / 700.0)
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.max(torch.zeros(1), torch.min(down, up))


class PreEmphasis(nn.Module):
    def __init__(self, coef: float = 0.97):
        super().__init__()
        self.coef = coef
        self.register_buffer("flipped_filter", torch.FloatTensor([-coef, 1.0]).unsqueeze(0).unsqueeze(0))


class _Spectrogram(nn.Module):
    def __init__(self, n_fft):
        super().__init__()
        self.register_buffer("window", torch.hamming_window(n_fft))


class _MelScale(nn.Module):
    def __init__(self, n_mels, sample_rate, f_min, n_stft):
        super().__init__()
        self.register_buffer("fb", melscale_fbanks(n_stft, f_min, float(sample_rate // 2), n_mels, sample_rate))


class MelSpectrogram(nn.Module):
    """Buffer container with torchaudio's layout (spectrogram.window [n_fft], mel_scale.fb [n_fft/2+1, n_mels])."""

    def __init__(self, sample_rate=16000, n_fft=512, win_length=512, hop_length=128, f_min=20.0, n_mels=80):
        super().__init__()
        if win_length != n_fft:
            raise NotImplementedError("MelSpectrogram front-end: win_length == n_fft only")
        self.n_fft, self.hop, self.n_mels = n_fft, hop_length, n_mels
        self.spectrogram = _Spectrogram(n_fft)
        self.mel_scale = _MelScale(n_mels, sample_rate, f_min, n_fft // 2 + 1)
        self._tables = {}

    def tables(self, device):
        """(windowed DFT basis [2*nf (padded to %4), n_fft], fb^T [n_mels, nf padded to %4]) on `device`."""
        key = (device.type, device.index)
        if key not in self._tables:
            n, nf = self.n_fft, self.n_fft // 2 + 1
            w = self.spectrogram.window.detach().double().cpu()
            k = torch.arange(n, dtype=torch.float64)
            ang = 2.0 * math.pi * torch.arange(nf, dtype=torch.float64).unsqueeze(1) * k.unsqueeze(0) / n
            basis = torch.zeros(-(-2 * nf // 4) * 4, n, dtype=torch.float64)
            basis[0:2 * nf:2] = torch.cos(ang) * w
            basis[1:2 * nf:2] = -torch.sin(ang) * w
            fbt = torch.zeros(self.n_mels, -(-nf // 4) * 4)
            fbt[:, :nf] = self.mel_scale.fb.detach().cpu().t()
            self._tables[key] = (basis.float().to(device).contiguous(), fbt.to(device).contiguous())
        return self._tables[key]


@torch.no_grad()
def fbank_frontend(wav, pre: PreEmphasis, mel: MelSpectrogram):
    """wav [R, Tw] -> log-mel features [R, Tf, n_mels], mean-normalised over time (bsrnn.py:343-350)."""
    if not wav.is_cuda:
        from ..._lib import WesepHipError
        raise WesepHipError("fbank front-end: wesep_amd has no CPU path")
    wav = wav.float().contiguous()
    R, T = wav.shape
    n, hop, nm = mel.n_fft, mel.hop, mel.n_mels
    nf, pad = n // 2 + 1, n // 2
    d = wav.device
    basis, fbt = mel.tables(d)
    ldo = -(-(T + 2 * pad) // 4) * 4
    xp = torch.zeros(R, ldo, device=d, dtype=torch.float32)
    dev.preemph_pad(wav, R, T, pad, ldo, pre.coef, xp)
    Tf = 1 + T // hop
    M = R * Tf
    lds = basis.shape[0]
    spec = _empty(d, M, lds)
    dev.gemm_nt(A=xp, a_rows=Rows(Tf, ldo, hop), M=M, N=lds, K=n, W=basis, ldw=n, C_out=spec, c_rows=flat(lds),
                vec=3, mode="f32")
    ldp = fbt.shape[1]
    power = _empty(d, M, ldp)
    dev.power_spec(spec, M, nf, lds, ldp, power)
    feats = _empty(d, M, nm)
    dev.gemm_nt(A=power, a_rows=flat(ldp), M=M, N=nm, K=ldp, W=fbt, ldw=ldp, C_out=feats, c_rows=flat(nm), vec=3,
                mode="f32")
    dev.log_eps(feats, 1e-8)
    neg_mean = dev.chan_sums(feats, None, None, 1, Tf, R, nm)[:, 0, :].contiguous()
    dev.affine_fwd(neg_mean, None, None, -1.0 / Tf, R, 1, nm, neg_mean)
    dev.affine_fwd(feats, None, neg_mean, 1.0, M, Tf, nm, feats)
    return feats.view(R, Tf, nm)
