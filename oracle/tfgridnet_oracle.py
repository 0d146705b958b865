"""TEST INFRASTRUCTURE ONLY -- CPU fp32 restatement of the wesep TF-GridNet path,
fixed-embedding mode (`joint_training=False`), single microphone, one source.  GROUNDWORK: the HIP path for this row
is NOT built (wesep_amd.models.get_model("TFGridNet") raises); the oracle is pinned now so that the next round starts
from a checked reference.

Functional re-statement (parameters as a dict keyed by the reference's `state_dict` names) of
  * `wesep/models/tfgridnet.py:197-302`               TFGridNet.forward (RMS normalisation, STFT n_fft 128 / hop 64,
                                                     3x3 conv + GroupNorm(1), n_layers x (speaker fusion, GridNetBlock),
                                                     3x3 transposed conv, iSTFT, de-normalisation)
  * `wesep/modules/tfgridnet/gridnet_block.py:118-227` GridNetBlock.forward (intra-frame and inter-frame BLSTM paths with
                                                     unfold / ConvTranspose1d, full-band multi-head self-attention)
  * `gridnet_block.py:230-284`                        LayerNormalization4DCF, AllHeadPReLULayerNormalization4DCF
  * `wesep/modules/common/speaker.py:102-121`         SpeakerFuseLayer (multiply / additive) on the [B, C, F, T] view
Pinned by `tests/golden/tfgridnet_*.npz` (generated from the real reference by `oracle/make_golden.py`).
Only tests/ may import this module."""
import math
from dataclasses import dataclass
from typing import Dict

import torch
import torch.nn.functional as F

from oracle.bsrnn_oracle import spk_transform


@dataclass
class TFGridNetConfig:
    n_fft: int = 128
    stride: int = 64
    n_layers: int = 6
    lstm_hidden_units: int = 192
    attn_n_head: int = 4
    attn_approx_qk_dim: int = 512
    emb_dim: int = 48
    emb_ks: int = 4
    emb_hs: int = 1
    eps: float = 1.0e-5
    spk_emb_dim: int = 256
    use_spk_transform: bool = False
    spk_fuse_type: str = "multiply"
    n_srcs: int = 1       # output sources: deconv to 2 * n_srcs channels, est [B, n_srcs, N] (tfgridnet.py:192-194,280-300)
    n_imics: int = 1      # microphones: input [B, N, M], conv from 2 * n_imics channels (tfgridnet.py:173,216-244)

    @property
    def n_freqs(self):
        return self.n_fft // 2 + 1


def param_shapes(cfg: TFGridNetConfig) -> Dict[str, tuple]:
    s: Dict[str, tuple] = {}
    E, Fq, C, H, ks = cfg.spk_emb_dim, cfg.n_freqs, cfg.emb_dim, cfg.lstm_hidden_units, cfg.emb_ks
    if cfg.use_spk_transform:
        s["spk_transform.transforms.0.weight"] = (128, E, 1)
        s["spk_transform.transforms.0.bias"] = (128,)
        s["spk_transform.transforms.1.weight"] = (128, 128, 1)
        s["spk_transform.transforms.1.bias"] = (128,)
        s["spk_transform.transforms.3.weight"] = (E, 128, 1)
        s["spk_transform.transforms.3.bias"] = (E,)
    if cfg.spk_fuse_type == "FiLM":            # norm.py:84-137
        s["spk_fuse.fc.gamma_fcs.0.weight"], s["spk_fuse.fc.gamma_fcs.0.bias"] = (Fq, E), (Fq,)
        s["spk_fuse.fc.beta_fcs.0.weight"], s["spk_fuse.fc.beta_fcs.0.bias"] = (Fq, E), (Fq,)
    elif cfg.spk_fuse_type in ("multiply", "additive", "concat"):
        s["spk_fuse.fc.linear.weight"] = (Fq, E + Fq if cfg.spk_fuse_type == "concat" else E)
        s["spk_fuse.fc.linear.bias"] = (Fq,)
    else:
        raise NotImplementedError(cfg.spk_fuse_type)
    s["conv.0.weight"] = (C, 2 * cfg.n_imics, 3, 3)
    s["conv.0.bias"] = (C,)
    s["conv.1.weight"] = (C,)
    s["conv.1.bias"] = (C,)
    Eq = math.ceil(cfg.attn_approx_qk_dim / Fq)
    nh = cfg.attn_n_head
    for i in range(cfg.n_layers):
        q = f"blocks.{i}."
        for path in ("intra", "inter"):
            s[q + f"{path}_norm.weight"] = (C,)
            s[q + f"{path}_norm.bias"] = (C,)
            for sfx in ("", "_reverse"):
                s[q + f"{path}_rnn.weight_ih_l0{sfx}"] = (4 * H, C * ks)
                s[q + f"{path}_rnn.weight_hh_l0{sfx}"] = (4 * H, H)
                s[q + f"{path}_rnn.bias_ih_l0{sfx}"] = (4 * H,)
                s[q + f"{path}_rnn.bias_hh_l0{sfx}"] = (4 * H,)
            if cfg.emb_ks == cfg.emb_hs:
                s[q + f"{path}_linear.weight"] = (C * ks, 2 * H)
                s[q + f"{path}_linear.bias"] = (C * ks,)
            else:
                s[q + f"{path}_linear.weight"] = (2 * H, C, ks)
                s[q + f"{path}_linear.bias"] = (C,)
        for name, ch in (("Q", Eq), ("K", Eq), ("V", C // nh)):
            s[q + f"attn_conv_{name}.weight"] = (nh * ch, C, 1, 1)
            s[q + f"attn_conv_{name}.bias"] = (nh * ch,)
            s[q + f"attn_norm_{name}.gamma"] = (1, nh, ch, 1, Fq)
            s[q + f"attn_norm_{name}.beta"] = (1, nh, ch, 1, Fq)
            s[q + f"attn_norm_{name}.act.weight"] = (nh,)
        s[q + "attn_concat_proj.0.weight"] = (C, C, 1, 1)
        s[q + "attn_concat_proj.0.bias"] = (C,)
        s[q + "attn_concat_proj.1.weight"] = (1,)
        s[q + "attn_concat_proj.2.gamma"] = (1, C, 1, Fq)
        s[q + "attn_concat_proj.2.beta"] = (1, C, 1, Fq)
    s["deconv.weight"] = (C, 2 * cfg.n_srcs, 3, 3)
    s["deconv.bias"] = (2 * cfg.n_srcs,)
    return s


def synth_params(cfg: TFGridNetConfig, seed: int) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, shp in param_shapes(cfg).items():
        if k.endswith(("gamma", "norm.weight", "conv.1.weight")):
            v = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif k.endswith(("beta", "norm.bias", "conv.1.bias")):
            v = 0.1 * torch.randn(shp, generator=g)
        elif k.endswith("act.weight") or k.endswith("attn_concat_proj.1.weight"):
            v = 0.25 + 0.05 * torch.randn(shp, generator=g)
        elif "bias" in k:
            v = 0.05 * torch.randn(shp, generator=g)
        else:
            fan = 1
            for d in shp[1:]:
                fan *= d
            if k == "deconv.weight" or k.endswith("_linear.weight") and len(shp) == 3:
                fan = shp[0]
            v = torch.randn(shp, generator=g) / fan ** 0.5
        out[k] = v
    return out


def _blstm(p, q, x, in_dim, hid):
    """nn.LSTM(in_dim, hid, 1, batch_first, bidirectional) with the parameters of prefix q (gridnet_block.py:40-46)."""
    rnn = torch.nn.LSTM(in_dim, hid, 1, batch_first=True, bidirectional=True)
    names = [n for n, _ in rnn.named_parameters()]
    return torch.func.functional_call(rnn, {n: p[q + n] for n in names}, (x,))[0]


def _ln4dcf(x, gamma, beta, eps):
    """LayerNormalization4DCF (gridnet_block.py:230-255): statistics over (C, F) per (b, t)."""
    mu = x.mean(dim=(1, 3), keepdim=True)
    std = torch.sqrt(x.var(dim=(1, 3), unbiased=False, keepdim=True) + eps)
    return (x - mu) / std * gamma + beta


def _allhead(p, q, x, nh, ch, Fq, eps):
    """AllHeadPReLULayerNormalization4DCF (gridnet_block.py:258-284): PReLU per head, LN over (E, F) per (b, h, t)."""
    B, _, T, _ = x.shape
    x = x.view(B, nh, ch, T, Fq)
    x = F.prelu(x, p[q + "act.weight"])
    mu = x.mean(dim=(2, 4), keepdim=True)
    std = torch.sqrt(x.var(dim=(2, 4), unbiased=False, keepdim=True) + eps)
    return (x - mu) / std * p[q + "gamma"] + p[q + "beta"]


def _rnn_path(p, cfg, q, path, x):
    """One BLSTM path of GridNetBlock on x [B', L, C] (sequences along L): LayerNorm, unfold(ks, hs), BLSTM, ConvTranspose1d /
    Linear, residual (gridnet_block.py:139-160)."""
    C, H, ks, hs = cfg.emb_dim, cfg.lstm_hidden_units, cfg.emb_ks, cfg.emb_hs
    y = F.layer_norm(x, (C,), p[q + f"{path}_norm.weight"], p[q + f"{path}_norm.bias"], cfg.eps)
    Bp, Lr, _ = y.shape
    if ks == hs:
        y = y.reshape(Bp, Lr // ks, ks * C)
        y = _blstm(p, q + f"{path}_rnn.", y, ks * C, H)
        y = F.linear(y, p[q + f"{path}_linear.weight"], p[q + f"{path}_linear.bias"]).reshape(Bp, Lr, C)
    else:
        y = F.unfold(y.transpose(1, 2)[..., None], (ks, 1), stride=(hs, 1)).transpose(1, 2)   # [B', n, C*ks]
        y = _blstm(p, q + f"{path}_rnn.", y, ks * C, H).transpose(1, 2)                      # [B', 2H, n]
        y = F.conv_transpose1d(y, p[q + f"{path}_linear.weight"], p[q + f"{path}_linear.bias"], stride=hs)
        y = y.transpose(1, 2)                                                               # [B', L, C]
    return y + x


def gridnet_block(p, cfg: TFGridNetConfig, q, x):
    """GridNetBlock.forward (gridnet_block.py:118-227): x [B, C, T, Q] -> same shape."""
    B, C, oT, oQ = x.shape
    ks, hs, nh, Fq = cfg.emb_ks, cfg.emb_hs, cfg.attn_n_head, cfg.n_freqs
    olp = ks - hs
    T = math.ceil((oT + 2 * olp - ks) / hs) * hs + ks
    Q = math.ceil((oQ + 2 * olp - ks) / hs) * hs + ks
    h = F.pad(x.permute(0, 2, 3, 1), (0, 0, olp, Q - oQ - olp, olp, T - oT - olp))          # [B, T, Q, C]
    h = _rnn_path(p, cfg, q, "intra", h.reshape(B * T, Q, C)).reshape(B, T, Q, C)
    h = h.transpose(1, 2)                                                                    # [B, Q, T, C]
    h = _rnn_path(p, cfg, q, "inter", h.reshape(B * Q, T, C)).reshape(B, Q, T, C)
    inter = h.permute(0, 3, 2, 1)[..., olp:olp + oT, olp:olp + oQ]                           # [B, C, T, Q]
    Eq = math.ceil(cfg.attn_approx_qk_dim / Fq)
    Qh = _allhead(p, q + "attn_norm_Q.", F.conv2d(inter, p[q + "attn_conv_Q.weight"], p[q + "attn_conv_Q.bias"]), nh, Eq, Fq, cfg.eps)
    Kh = _allhead(p, q + "attn_norm_K.", F.conv2d(inter, p[q + "attn_conv_K.weight"], p[q + "attn_conv_K.bias"]), nh, Eq, Fq, cfg.eps)
    Vh = _allhead(p, q + "attn_norm_V.", F.conv2d(inter, p[q + "attn_conv_V.weight"], p[q + "attn_conv_V.bias"]), nh, C // nh, Fq, cfg.eps)
    Qm = Qh.reshape(B * nh, Eq, oT, oQ).transpose(1, 2).flatten(2)                           # [B', T, E*Q]
    Km = Kh.reshape(B * nh, Eq, oT, oQ).transpose(1, 2).flatten(2).transpose(1, 2)           # [B', E*Q, T]
    Vm = Vh.reshape(B * nh, C // nh, oT, oQ).transpose(1, 2)                                 # [B', T, c, Q]
    vshape = Vm.shape
    att = F.softmax(torch.matmul(Qm, Km) / (Qm.shape[-1] ** 0.5), dim=2)
    V = torch.matmul(att, Vm.flatten(2)).reshape(vshape).transpose(1, 2)                     # [B', c, T, Q]
    V = V.contiguous().view(B, C, oT, oQ)
    V = F.conv2d(V, p[q + "attn_concat_proj.0.weight"], p[q + "attn_concat_proj.0.bias"])
    V = F.prelu(V, p[q + "attn_concat_proj.1.weight"])
    V = _ln4dcf(V, p[q + "attn_concat_proj.2.gamma"], p[q + "attn_concat_proj.2.beta"], cfg.eps)
    return V + inter


def tfgridnet_forward(p: Dict[str, torch.Tensor], cfg: TFGridNetConfig, wav: torch.Tensor, emb: torch.Tensor):
    """`tfgridnet.py:197-302`, joint_training=False: wav [B, N] (or [B, N, M] with n_imics = M > 1), emb [B, E] ->
    est [B, N] (or [B, n_srcs, N] with n_srcs > 1)."""
    B, n = wav.shape[0], wav.shape[1]
    w3 = wav.unsqueeze(-1) if wav.dim() == 2 else wav                                        # [B, N, M]
    M = w3.shape[2]
    std = torch.std(w3, dim=(1, 2), keepdim=True)                                            # [B, 1, 1]
    x = (w3 / std).transpose(1, 2).reshape(B * M, n)
    win = torch.hann_window(cfg.n_fft)
    spec = torch.stft(x, cfg.n_fft, cfg.stride, cfg.n_fft, window=win, return_complex=True, onesided=True)
    spec = spec.transpose(1, 2)                                                              # [B*M, T, F]
    spec = spec.view(B, M, spec.shape[1], spec.shape[2])
    h = torch.cat((spec.real, spec.imag), 1)                                                 # [B, 2M, T, F]
    _, _, nT, nF = h.shape
    h = F.group_norm(F.conv2d(h, p["conv.0.weight"], p["conv.0.bias"], padding=(1, 1)), 1, p["conv.1.weight"],
                     p["conv.1.bias"], cfg.eps)
    if cfg.use_spk_transform:
        emb = spk_transform(p, emb)
    if cfg.spk_fuse_type == "FiLM":
        gm = F.linear(emb, p["spk_fuse.fc.gamma_fcs.0.weight"], p["spk_fuse.fc.gamma_fcs.0.bias"]).view(B, 1, 1, nF)
        bt = F.linear(emb, p["spk_fuse.fc.beta_fcs.0.weight"], p["spk_fuse.fc.beta_fcs.0.bias"]).view(B, 1, 1, nF)
    elif cfg.spk_fuse_type != "concat":
        t = F.linear(emb, p["spk_fuse.fc.linear.weight"], p["spk_fuse.fc.linear.bias"]).view(B, 1, 1, nF)
    for i in range(cfg.n_layers):
        if cfg.spk_fuse_type == "FiLM":
            h = (1 + gm) * h + bt
        elif cfg.spk_fuse_type == "concat":     # speaker.py:95-101 on the [B, C, F, T] view: Linear over cat[x, e] along F
            ee = emb.view(B, 1, 1, -1).expand(-1, h.shape[1], h.shape[2], -1)                      # [B, C, T, E]
            h = F.linear(torch.cat([h, ee], 3), p["spk_fuse.fc.linear.weight"], p["spk_fuse.fc.linear.bias"])
        else:
            h = h * t if cfg.spk_fuse_type == "multiply" else h + t
        h = gridnet_block(p, cfg, f"blocks.{i}.", h)
    h = F.conv_transpose2d(h, p["deconv.weight"], p["deconv.bias"], padding=(1, 1))          # [B, 2S, T, F]
    S = cfg.n_srcs
    h = h.view(B, S, 2, nT, nF)
    est = torch.complex(h[:, :, 0], h[:, :, 1]).reshape(B * S, nT, nF).transpose(1, 2)       # [B*S, F, T]
    y = torch.istft(est, cfg.n_fft, cfg.stride, cfg.n_fft, window=win, onesided=True, length=n)
    return (y.view(B, S, n) * std.view(B, 1, 1)).squeeze(1)
