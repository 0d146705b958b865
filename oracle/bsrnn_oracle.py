"""TEST INFRASTRUCTURE ONLY -- CPU fp32 restatement of the wesep pBSRNN hot path.

This file is the parity oracle for the HIP path in `wesep_amd/`.  It is a
functional re-statement (plain torch CPU ops, parameters passed as a dict
keyed by the reference's `state_dict` names) of:

  * `wesep/models/bsrnn.py:300-394`   BSRNN.forward
  * `wesep/models/bsrnn.py:38-46`     ResRNN.forward
  * `wesep/models/bsrnn.py:69-83`     BSNet.forward
  * `wesep/models/bsrnn.py:125-148`   FuseSeparation.forward
  * `wesep/modules/common/speaker.py:26-49,63-125`  SpeakerTransform / SpeakerFuseLayer
  * `wesep/modules/common/norm.py:84-139`           FiLM
  * auraloss.time.SISDRLoss (third-party, unpinned in requirements.txt:25;
    used via `wesep/utils/losses.py:24-25`) -- restated from its published source
  * `wesep/utils/funcs.py:79-88`      clip_gradients (per-tensor L2 clip)
  * torch.optim.Adam with coupled L2 weight decay (`wesep/bin/train.py:237-238`)
  * `wesep/utils/schedulers.py:99-222` ExponentialDecrease

Pinning: the reference ships no tests or golden vectors, so
the oracle is pinned against outputs of the reference itself, imported in the
authoring container with third-party stubs by `oracle/make_golden.py`; the
resulting fixtures live in `tests/golden/` and `tests/test_oracle_golden.py`
checks this file against them.  SISDRLoss (auraloss) is cross-checked against
the reference's independent numpy `cal_SISNR` (`wesep/utils/score.py:7-21`)
inside the golden script, but auraloss itself is absent: that part is
"parity unpinned" against the third-party source.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this module.  The product path (`wesep_amd`) never does.
"""
import math
from dataclasses import dataclass, field
from typing import Dict, List

import numpy as np
import torch
import torch.nn.functional as F

GN_EPS = float(torch.finfo(torch.float32).eps)  # bsrnn.py:23,184


@dataclass
class BSRNNConfig:
    """Constructor arguments of the reference BSRNN that matter on the path
    (`bsrnn.py:154-175`), fixed-embedding mode (`joint_training=False`)."""
    spk_emb_dim: int = 256
    sr: int = 16000
    win: int = 512
    stride: int = 128
    feature_dim: int = 128
    num_repeat: int = 6
    use_spk_transform: bool = False
    spk_fuse_type: str = "multiply"
    multi_fuse: bool = False
    band_width: List[int] = field(default_factory=list)

    def __post_init__(self):
        if not self.band_width:
            self.band_width = band_widths(self.sr, self.win)

    @property
    def nband(self):
        return len(self.band_width)

    @property
    def enc_dim(self):
        return self.win // 2 + 1


def band_widths(sr: int, win: int) -> List[int]:
    """Band table, `bsrnn.py:190-209`."""
    enc = win // 2 + 1
    nyq = sr / 2.0
    bw100 = int(np.floor(100 / nyq * enc))
    bw200 = int(np.floor(200 / nyq * enc))
    bw500 = int(np.floor(500 / nyq * enc))
    bw2k = int(np.floor(2000 / nyq * enc))
    bws = [bw100] * 15 + [bw200] * 10 + [bw500] * 5 + [bw2k]
    bws.append(enc - int(np.sum(bws)))
    return bws


def fuse_layer_indices(cfg: BSRNNConfig):
    """Index of every fuse layer / BSNet inside `separator.separation`
    (`bsrnn.py:106-123`)."""
    fuse, nets = [], []
    if cfg.multi_fuse:
        for i in range(cfg.num_repeat):
            fuse.append(2 * i)
            nets.append(2 * i + 1)
    else:
        fuse.append(0)
        nets = list(range(1, cfg.num_repeat + 1))
    return fuse, nets


def param_shapes(cfg: BSRNNConfig) -> Dict[str, tuple]:
    """Name -> shape of every parameter, in the reference's registration order."""
    N, H, E = cfg.feature_dim, 2 * cfg.feature_dim, cfg.spk_emb_dim
    shapes: Dict[str, tuple] = {}
    if cfg.use_spk_transform:  # speaker.py:26-43 (embed 256, hid 128, 3 layers)
        shapes["spk_transform.transforms.0.weight"] = (128, E, 1)
        shapes["spk_transform.transforms.0.bias"] = (128,)
        shapes["spk_transform.transforms.1.weight"] = (128, 128, 1)
        shapes["spk_transform.transforms.1.bias"] = (128,)
        shapes["spk_transform.transforms.3.weight"] = (E, 128, 1)
        shapes["spk_transform.transforms.3.bias"] = (E,)
    for i, bw in enumerate(cfg.band_width):
        shapes[f"BN.{i}.0.weight"] = (2 * bw,)
        shapes[f"BN.{i}.0.bias"] = (2 * bw,)
        shapes[f"BN.{i}.1.weight"] = (N, 2 * bw, 1)
        shapes[f"BN.{i}.1.bias"] = (N,)
    fuse, nets = fuse_layer_indices(cfg)

    def add_fuse(idx):
        p = f"separator.separation.{idx}.fc."
        if cfg.spk_fuse_type == "concat":
            shapes[p + "linear.weight"] = (N, E + N)
            shapes[p + "linear.bias"] = (N,)
        elif cfg.spk_fuse_type in ("additive", "multiply"):
            shapes[p + "linear.weight"] = (N, E)
            shapes[p + "linear.bias"] = (N,)
        elif cfg.spk_fuse_type == "FiLM":
            shapes[p + "gamma_fcs.0.weight"] = (N, E)
            shapes[p + "gamma_fcs.0.bias"] = (N,)
            shapes[p + "beta_fcs.0.weight"] = (N, E)
            shapes[p + "beta_fcs.0.bias"] = (N,)
        else:
            raise ValueError("Fuse type not defined.")

    def add_net(idx):
        for rnn in ("band_rnn", "band_comm"):
            p = f"separator.separation.{idx}.{rnn}."
            shapes[p + "norm.weight"] = (N,)
            shapes[p + "norm.bias"] = (N,)
            for sfx in ("", "_reverse"):
                shapes[p + f"rnn.weight_ih_l0{sfx}"] = (4 * H, N)
                shapes[p + f"rnn.weight_hh_l0{sfx}"] = (4 * H, H)
                shapes[p + f"rnn.bias_ih_l0{sfx}"] = (4 * H,)
                shapes[p + f"rnn.bias_hh_l0{sfx}"] = (4 * H,)
            shapes[p + "proj.weight"] = (N, 2 * H)
            shapes[p + "proj.bias"] = (N,)

    order = sorted([(i, "f") for i in fuse] + [(i, "n") for i in nets])
    for idx, kind in order:
        (add_fuse if kind == "f" else add_net)(idx)
    for i, bw in enumerate(cfg.band_width):
        shapes[f"mask.{i}.0.weight"] = (N,)
        shapes[f"mask.{i}.0.bias"] = (N,)
        shapes[f"mask.{i}.1.weight"] = (4 * N, N, 1)
        shapes[f"mask.{i}.1.bias"] = (4 * N,)
        shapes[f"mask.{i}.3.weight"] = (4 * N, 4 * N, 1)
        shapes[f"mask.{i}.3.bias"] = (4 * N,)
        shapes[f"mask.{i}.5.weight"] = (4 * bw, 4 * N, 1)
        shapes[f"mask.{i}.5.bias"] = (4 * bw,)
    return shapes


def synth_params(cfg: BSRNNConfig, seed: int) -> Dict[str, torch.Tensor]:
    """Deterministic CPU parameter set (same values on every machine with this
    torch build): weights U(-a, a) with a = 1/sqrt(fan_in), norm weights
    1 + 0.1*N(0,1), biases 0.1*U(-a, a)-like.  Used instead of shipping ~50 MB
    of reference-initialised weights in the fixtures."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name, shape in param_shapes(cfg).items():
        if len(shape) >= 2:
            fan_in = int(np.prod(shape[1:]))
            a = 1.0 / math.sqrt(fan_in)
            t = (torch.rand(shape, generator=g) * 2 - 1) * a
        elif name.endswith("norm.weight") or (name.endswith(".0.weight") and name.startswith(("BN.", "mask."))):
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif "bias_ih" in name or "bias_hh" in name:
            t = (torch.rand(shape, generator=g) * 2 - 1) / math.sqrt(2 * cfg.feature_dim)
        else:
            t = 0.1 * torch.randn(shape, generator=g)
        out[name] = t.float().contiguous()
    return out


# --------------------------------------------------------------------------
# forward pieces
# --------------------------------------------------------------------------
def _group_norm1(x, w, b):
    """GroupNorm(1, C, eps=finfo.eps) on [B, C, L] (`bsrnn.py:26,256,275`)."""
    return F.group_norm(x, 1, w, b, GN_EPS)


def res_rnn(p: Dict[str, torch.Tensor], prefix: str, x: torch.Tensor) -> torch.Tensor:
    """`bsrnn.py:38-46`: x [B, N, L] -> x + proj(BLSTM(norm(x)))."""
    B, N, L = x.shape
    y = _group_norm1(x, p[prefix + "norm.weight"], p[prefix + "norm.bias"])
    y = y.transpose(1, 2).contiguous()
    H = p[prefix + "rnn.weight_hh_l0"].shape[1]
    flat = [p[prefix + "rnn." + k] for k in (
        "weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0",
        "weight_ih_l0_reverse", "weight_hh_l0_reverse", "bias_ih_l0_reverse",
        "bias_hh_l0_reverse")]
    h0 = x.new_zeros(2, B, H)
    out, _, _ = torch._VF.lstm(y, (h0, h0.clone()), flat, True, 1, 0.0, False, True, True)
    out = F.linear(out.reshape(-1, 2 * H), p[prefix + "proj.weight"], p[prefix + "proj.bias"])
    return x + out.view(B, L, N).transpose(1, 2)


def bs_net(p, prefix, x, nband):
    """`bsrnn.py:69-83`: x [B, nband*N, T]."""
    B, KN, T = x.shape
    N = KN // nband
    y = res_rnn(p, prefix + "band_rnn.", x.reshape(B * nband, N, T)).view(B, nband, N, T)
    y = y.permute(0, 3, 2, 1).reshape(B * T, N, nband)
    y = res_rnn(p, prefix + "band_comm.", y).view(B, T, N, nband).permute(0, 3, 2, 1)
    return y.reshape(B, KN, T)


def spk_transform(p, e):
    """`speaker.py:26-49` (Conv1d k=1 chain on [B, E])."""
    pre = "spk_transform.transforms."
    h = F.linear(e, p[pre + "0.weight"].squeeze(-1), p[pre + "0.bias"])
    h = torch.tanh(F.linear(h, p[pre + "1.weight"].squeeze(-1), p[pre + "1.bias"]))
    return F.linear(h, p[pre + "3.weight"].squeeze(-1), p[pre + "3.bias"])


def speaker_fuse(p, prefix, kind, x, e):
    """`speaker.py:81-125` on the 4-D path: x [B, K, N, T], e [B, E]."""
    if kind == "concat":
        B, K, N, T = x.shape
        W, b = p[prefix + "fc.linear.weight"], p[prefix + "fc.linear.bias"]
        ee = e.view(B, 1, 1, -1).expand(B, K, T, e.shape[1])
        y = torch.cat([x.transpose(2, 3), ee], dim=3)
        return F.linear(y, W, b).transpose(2, 3).contiguous()
    if kind == "additive":
        a = F.linear(e, p[prefix + "fc.linear.weight"], p[prefix + "fc.linear.bias"])
        return x + a[:, None, :, None]
    if kind == "multiply":
        a = F.linear(e, p[prefix + "fc.linear.weight"], p[prefix + "fc.linear.bias"])
        return x * a[:, None, :, None]
    if kind == "FiLM":  # norm.py:118-139
        g = F.linear(e, p[prefix + "fc.gamma_fcs.0.weight"], p[prefix + "fc.gamma_fcs.0.bias"])
        bt = F.linear(e, p[prefix + "fc.beta_fcs.0.weight"], p[prefix + "fc.beta_fcs.0.bias"])
        return (1 + g[:, None, :, None]) * x + bt[:, None, :, None]
    raise ValueError("Fuse type not defined.")


def band_split(p, cfg: BSRNNConfig, wav):
    """`bsrnn.py:306-337`: wav [R, T] -> (complex spectrogram [R, F, Tf], band features z [R, K, N, Tf])."""
    R, T = wav.shape
    window = torch.hann_window(cfg.win, dtype=wav.dtype)
    spec = torch.stft(wav, n_fft=cfg.win, hop_length=cfg.stride, window=window,
                      return_complex=True)                      # [R, F, Tf]
    spec_ri = torch.stack([spec.real, spec.imag], 1)           # [R, 2, F, Tf]
    feats, f0 = [], 0
    for i, bw in enumerate(cfg.band_width):
        sb = spec_ri[:, :, f0:f0 + bw].reshape(R, 2 * bw, -1)
        sb = _group_norm1(sb, p[f"BN.{i}.0.weight"], p[f"BN.{i}.0.bias"])
        feats.append(F.conv1d(sb, p[f"BN.{i}.1.weight"], p[f"BN.{i}.1.bias"]))
        f0 += bw
    return spec, torch.stack(feats, 1)                          # [R, K, N, Tf]


def separate(p, cfg: BSRNNConfig, z, emb):
    """`bsrnn.py:359-364`: spk_transform + FuseSeparation on z [R, K, N, Tf] with the embedding [R, E]."""
    R, K, N = z.shape[0], cfg.nband, cfg.feature_dim
    e = spk_transform(p, emb) if cfg.use_spk_transform else emb
    fuse, nets = fuse_layer_indices(cfg)
    order = sorted([(i, "f") for i in fuse] + [(i, "n") for i in nets])
    for idx, kind in order:
        pre = f"separator.separation.{idx}."
        if kind == "f":
            z = speaker_fuse(p, pre, cfg.spk_fuse_type, z, e)
        else:
            z = bs_net(p, pre, z.reshape(R, K * N, -1), K).view(R, K, N, -1)
    return z


def mask_decode(p, cfg: BSRNNConfig, z, spec, T):
    """`bsrnn.py:366-392`: mask MLP, GLU complex mask on the mixture's band spectra, iSTFT -> (est, est_spec)."""
    R = z.shape[0]
    window = torch.hann_window(cfg.win, dtype=z.dtype)
    est_bands, f0 = [], 0
    for i, bw in enumerate(cfg.band_width):
        h = _group_norm1(z[:, i], p[f"mask.{i}.0.weight"], p[f"mask.{i}.0.bias"])
        h = torch.tanh(F.conv1d(h, p[f"mask.{i}.1.weight"], p[f"mask.{i}.1.bias"]))
        h = torch.tanh(F.conv1d(h, p[f"mask.{i}.3.weight"], p[f"mask.{i}.3.bias"]))
        o = F.conv1d(h, p[f"mask.{i}.5.weight"], p[f"mask.{i}.5.bias"]).view(R, 2, 2, bw, -1)
        m = o[:, 0] * torch.sigmoid(o[:, 1])                    # [R, 2(re/im), bw, Tf]
        xb = spec[:, f0:f0 + bw]
        er = xb.real * m[:, 0] - xb.imag * m[:, 1]
        ei = xb.real * m[:, 1] + xb.imag * m[:, 0]
        est_bands.append(torch.complex(er, ei))
        f0 += bw
    est_spec = torch.cat(est_bands, 1)
    return torch.istft(est_spec, n_fft=cfg.win, hop_length=cfg.stride, window=window, length=T), est_spec


def bsrnn_forward(p: Dict[str, torch.Tensor], cfg: BSRNNConfig, wav: torch.Tensor,
                  emb: torch.Tensor, return_intermediates: bool = False):
    """`bsrnn.py:300-394`, `joint_training=False`: wav [R, T], emb [R, E] -> est [R, T]."""
    spec, z = band_split(p, cfg, wav)
    inter = {"spec": spec, "z0": z}
    z = separate(p, cfg, z, emb)
    inter["z_sep"] = z
    est, inter["est_spec"] = mask_decode(p, cfg, z, spec, wav.shape[1])
    if return_intermediates:
        return est, inter
    return est


def bsrnn_multi_forward(p, cfg: BSRNNConfig, wav, enroll, embed_fn):
    """`BSRNN_Multi.forward` in grad mode (bsrnn_multi_optim.py:300-470): two separator passes over one band split;
    the second one is conditioned on the embedding of the first pass's detached estimate.  `embed_fn(waveform
    [R, Tw]) -> [R, E]` is the jointly trained path (in-model fbank front-end + speaker encoder).
    -> (s, self_s, embedding of the enrollment, embedding of s)."""
    spec, z = band_split(p, cfg, wav)
    e1 = embed_fn(enroll)
    s, _ = mask_decode(p, cfg, separate(p, cfg, z, e1), spec, wav.shape[1])
    e2 = embed_fn(s.detach())
    self_s, _ = mask_decode(p, cfg, separate(p, cfg, z, e2), spec, wav.shape[1])
    return s, self_s, e1, e2


# --------------------------------------------------------------------------
# loss / clip / optimizer / scheduler
# --------------------------------------------------------------------------
def sisdr_loss(est: torch.Tensor, target: torch.Tensor, eps: float = 1e-8) -> torch.Tensor:
    """auraloss.time.SISDRLoss(zero_mean=True, eps=1e-8, reduction='mean')."""
    x = est - est.mean(-1, keepdim=True)
    t = target - target.mean(-1, keepdim=True)
    alpha = (x * t).sum(-1) / ((t ** 2).sum(-1) + eps)
    tt = t * alpha.unsqueeze(-1)
    res = x - tt
    val = 10 * torch.log10((tt ** 2).sum(-1) / ((res ** 2).sum(-1) + eps) + eps)
    return -val.mean()


def clip_gradients_(grads: Dict[str, torch.Tensor], clip: float) -> Dict[str, float]:
    """`funcs.py:79-88`: per-tensor L2 clip, in place; returns the norms."""
    norms = {}
    for k, g in grads.items():
        n = g.norm(2)
        norms[k] = float(n)
        coef = clip / (n + 1e-6)
        if coef < 1:
            g.mul_(coef)
    return norms


def adam_l2_step_(param, grad, exp_avg, exp_avg_sq, step: int, lr: float,
                  beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0):
    """torch.optim.Adam single-tensor update (coupled L2, not AdamW), in place."""
    g = grad + weight_decay * param if weight_decay != 0 else grad
    exp_avg.mul_(beta1).add_(g, alpha=1 - beta1)
    exp_avg_sq.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (exp_avg_sq.sqrt() / math.sqrt(bc2)).add_(eps)
    param.addcdiv_(exp_avg, denom, value=-(lr / bc1))


def exponential_decrease_lr(cur_iter, max_iter, initial_lr, final_lr, warm_up_iter=0,
                            scale_ratio=1.0, warm_from_zero=False):
    """`schedulers.py:118-128,217-222`."""
    coeff = 1.0 * scale_ratio
    if cur_iter < warm_up_iter:
        if warm_from_zero:
            coeff = scale_ratio * cur_iter / warm_up_iter
        elif scale_ratio > 1:
            coeff = (scale_ratio - 1) * cur_iter / warm_up_iter + 1.0
    return coeff * initial_lr * math.exp((cur_iter / max_iter) * math.log(final_lr / initial_lr))


def cal_sisnr_np(ref_sig: np.ndarray, out_sig: np.ndarray, eps: float = 1e-8) -> float:
    """Evaluation metric `wesep/utils/score.py:7-21` (numpy, different eps placement)."""
    ref_sig = ref_sig - np.mean(ref_sig)
    out_sig = out_sig - np.mean(out_sig)
    ref_energy = np.sum(ref_sig ** 2) + eps
    proj = np.sum(ref_sig * out_sig) * ref_sig / ref_energy
    noise = out_sig - proj
    ratio = np.sum(proj ** 2) / (np.sum(noise ** 2) + eps)
    return float(10 * np.log(ratio + eps) / np.log(10.0))


def synth_batch(R: int, T: int, seed: int, emb_dim: int = 256):
    """Synthetic 2-speaker rows exactly as BASELINE.md section 3 prescribes:
    s1, s2 ~ 0.1*N(0,1); mix = s1+s2 peak-normalised to <= 1; rows interleaved
    (mix,s1),(mix,s2) (`dataset.py:217-227`); enrollment N(0,1) [R, emb_dim]."""
    assert R % 2 == 0
    g = torch.Generator().manual_seed(seed)
    s = 0.1 * torch.randn(R // 2, 2, T, generator=g)
    mix = s.sum(1)
    peak = mix.abs().amax(-1, keepdim=True).clamp_min(1.0)
    mix = mix / peak
    s = s / peak[:, None]
    wav_mix = mix[:, None, :].expand(R // 2, 2, T).reshape(R, T).contiguous()
    wav_tgt = s.reshape(R, T).contiguous()
    emb = torch.randn(R, emb_dim, generator=g)
    return wav_mix.float(), wav_tgt.float(), emb.float()
