"""TEST INFRASTRUCTURE.  Coarse CPU emulation of the pBSRNN autograd functions (`wesep_amd/functional.py`) for host-
logic tests of models that COMPOSE them (BSRNN_Multi's two passes over one band split, the SSA step).

`tests/emu_dev.py` emulates individual `ws_*` entry points; the pBSRNN functions drive grouped GEMMs through
device-side descriptor tables, which is not worth emulating call by call.  Instead each autograd Function the
model code touches is replaced by an object with the same `.apply(...)` signature, Z layout ([R, K, Tf, N]) and
semantics, written as differentiable torch statements that follow the oracle (oracle/bsrnn_oracle.py).  The HIP
implementations of these functions are covered by tests/test_bsrnn_gpu.py; what this harness checks is everything
above them.  The product has no CPU path; nothing outside tests/ imports this."""
import torch
import torch.nn.functional as F

from oracle import bsrnn_oracle as O

WIN, HOP = 512, 128


class BandSplitFn:
    @staticmethod
    def apply(wav, plan, *params):
        R, T = wav.shape
        window = torch.hann_window(WIN, dtype=wav.dtype)
        spec = torch.stft(wav, n_fft=WIN, hop_length=HOP, window=window, return_complex=True)    # [R, F, Tf]
        ri = torch.stack([spec.real, spec.imag], 1)
        feats, f0 = [], 0
        for g, bw in enumerate(plan.bw):
            gw, gb, cw, cb = params[4 * g:4 * g + 4]
            sb = O._group_norm1(ri[:, :, f0:f0 + bw].reshape(R, 2 * bw, -1), gw, gb)
            feats.append(F.conv1d(sb, cw, cb))
            f0 += bw
        z = torch.stack(feats, 1)                                   # [R, K, N, Tf]
        return z.permute(0, 1, 3, 2).contiguous(), (spec, list(plan.bw))


class MaskDecodeFn:
    @staticmethod
    def apply(z, xbs, plan, T, *params):
        spec, bws = xbs
        R = z.shape[0]
        zz = z.permute(0, 1, 3, 2)                                  # [R, K, N, Tf]
        bands, f0 = [], 0
        for i, bw in enumerate(bws):
            gw, gb, w1, b1, w2, b2, w3, b3 = params[8 * i:8 * i + 8]
            h = O._group_norm1(zz[:, i], gw, gb)
            h = torch.tanh(F.conv1d(h, w1, b1))
            h = torch.tanh(F.conv1d(h, w2, b2))
            o = F.conv1d(h, w3, b3).view(R, 2, 2, bw, -1)
            m = o[:, 0] * torch.sigmoid(o[:, 1])
            xb = spec[:, f0:f0 + bw]
            bands.append(torch.complex(xb.real * m[:, 0] - xb.imag * m[:, 1], xb.real * m[:, 1] + xb.imag * m[:, 0]))
            f0 += bw
        window = torch.hann_window(WIN, dtype=z.dtype)
        return torch.istft(torch.cat(bands, 1), n_fft=WIN, hop_length=HOP, window=window, length=T)


def resrnn(z, view, norm_w, norm_b, *params, carrier=None, cache=None):
    R, K, Tf, N = z.shape
    names = ("rnn.weight_ih_l0", "rnn.weight_hh_l0", "rnn.bias_ih_l0", "rnn.bias_hh_l0", "rnn.weight_ih_l0_reverse",
             "rnn.weight_hh_l0_reverse", "rnn.bias_ih_l0_reverse", "rnn.bias_hh_l0_reverse", "proj.weight", "proj.bias")
    p = dict(zip(names, params))
    p["norm.weight"], p["norm.bias"] = norm_w, norm_b
    if view == "time":
        x = z.permute(0, 1, 3, 2).reshape(R * K, N, Tf)
        return O.res_rnn(p, "", x).view(R, K, N, Tf).permute(0, 1, 3, 2).contiguous()
    x = z.permute(0, 2, 3, 1).reshape(R * Tf, N, K)
    return O.res_rnn(p, "", x).view(R, Tf, N, K).permute(0, 3, 1, 2).contiguous()


class AffineFn:
    @staticmethod
    def apply(z, a, b, a0):
        scale = a0 + (a[:, None, None, :] if a is not None else 0.0)
        return z * scale + (b[:, None, None, :] if b is not None else 0.0)


class ConcatFuseFn:
    @staticmethod
    def apply(z, e, W, b):
        R, K, Tf, N = z.shape
        ee = e.view(R, 1, 1, -1).expand(R, K, Tf, e.shape[1])
        return F.linear(torch.cat([z, ee], 3), W, b)


class SISDRFn:
    @staticmethod
    def apply(est, tgt, eps):
        return O.sisdr_loss(est, tgt, eps)


def install(monkeypatch, real_resrnn=False):
    """On top of emu_dev.install (entry-point emulation for the speaker encoder / front-end / linear layers).
    `real_resrnn=True` keeps the product's blocked-layout ResRNN (functional.ResRNNBlkFn) in place -- it then needs
    tests/emu_blk.py installed as well -- and only stands in for the band split, mask decode, fusion and loss."""
    import wesep_amd.functional as f0
    for name, obj in (("BandSplitFn", BandSplitFn), ("MaskDecodeFn", MaskDecodeFn), ("resrnn", resrnn),
                      ("AffineFn", AffineFn), ("ConcatFuseFn", ConcatFuseFn), ("SISDRFn", SISDRFn)):
        if name == "resrnn" and real_resrnn:
            continue
        monkeypatch.setattr(f0, name, obj)
    monkeypatch.setattr(f0, "make_wgrad_carrier", lambda params, blocked=None: None)
    monkeypatch.setattr(f0, "reset_deferred_wgrads", lambda device: None)
