"""TEST INFRASTRUCTURE ONLY -- CPU fp32 restatement of the wesep DPCCN path,
fixed-embedding mode (`joint_training=False`).  Groundwork for the next round: the HIP path for this row is NOT built
yet (wesep_amd.models.get_model("DPCCN") raises); this oracle is already pinned so that work can start from a
checked reference.

Functional re-statement (parameters as a dict keyed by the reference's `state_dict` names) of
  * `wesep/models/dpccn.py:206-290`           DPCCN.forward (STFT -> conv2d -> dense encoder -> speaker fusion ->
                                              strided encoder -> TCN on the flattened (T, F') grid -> U-Net decoder ->
                                              pyramid pooling -> deconv -> iSTFT)
  * `wesep/modules/dpccn/convs.py:28-152`     Conv2dBlock / ConvTrans2dBlock (conv - ELU - InstanceNorm2d),
                                              DenseBlock (5 densely connected Conv2dBlocks), TCNBlock
                                              (InstanceNorm1d - ELU - depthwise dilated conv - IN - ELU - 1x1 conv, residual)
  * `wesep/modules/common/speaker.py:63-125`  SpeakerFuseLayer on the [B, C, F, T] view (multiply / additive / concat)
Pinned by `tests/golden/dpccn_*.npz` (generated from the real reference by `oracle/make_golden.py`).
Only tests/ may import this module."""
from dataclasses import dataclass
from typing import Dict

import torch
import torch.nn.functional as F

from oracle.bsrnn_oracle import spk_transform


@dataclass
class DPCCNConfig:
    win: int = 512
    stride: int = 128
    spk_emb_dim: int = 256
    use_spk_transform: bool = False
    spk_fuse_type: str = "multiply"
    feature_dim: int = 257
    tcn_dims: int = 384
    tcn_blocks: int = 10
    tcn_layers: int = 2
    pool_size: tuple = (4, 8, 16, 32)
    causal: bool = False            # TCN blocks pad dil * (k - 1) on both sides and cut the tail (convs.py:126-127,147-148)


def _dense_shapes(s, p, cin, cout, mode):
    n = 1 if mode == "enc" else 2
    for i in range(1, 6):
        o = cout if i == 5 else cin
        s[f"{p}conv{i}.conv2d.weight"] = (o, cin * (n + i - 1), 3, 3)
        s[f"{p}conv{i}.conv2d.bias"] = (o,)


def param_shapes(cfg: DPCCNConfig) -> Dict[str, tuple]:
    s: Dict[str, tuple] = {}
    s["conv2d.weight"] = (16, 2, 3, 3)
    s["conv2d.bias"] = (16,)
    _dense_shapes(s, "encoder.0.", 16, 16, "enc")
    for i in range(4):
        s[f"encoder.{i + 1}.0.conv2d.weight"] = (32, 16 if i == 0 else 32, 3, 3)
        s[f"encoder.{i + 1}.0.conv2d.bias"] = (32,)
        _dense_shapes(s, f"encoder.{i + 1}.1.", 32, 32, "enc")
    for j, (ci, co) in zip((5, 6, 7), ((32, 64), (64, 128), (128, 384))):
        s[f"encoder.{j}.conv2d.weight"] = (co, ci, 3, 3)
        s[f"encoder.{j}.conv2d.bias"] = (co,)
    E, Fd = cfg.spk_emb_dim, cfg.feature_dim
    if cfg.use_spk_transform:
        s["spk_transform.transforms.0.weight"] = (128, E, 1)
        s["spk_transform.transforms.0.bias"] = (128,)
        s["spk_transform.transforms.1.weight"] = (128, 128, 1)
        s["spk_transform.transforms.1.bias"] = (128,)
        s["spk_transform.transforms.3.weight"] = (E, 128, 1)
        s["spk_transform.transforms.3.bias"] = (E,)
    if cfg.spk_fuse_type == "concat":
        s["spk_fuse.fc.linear.weight"] = (Fd, E + Fd)
        s["spk_fuse.fc.linear.bias"] = (Fd,)
    elif cfg.spk_fuse_type in ("additive", "multiply"):
        s["spk_fuse.fc.linear.weight"] = (Fd, E)
        s["spk_fuse.fc.linear.bias"] = (Fd,)
    elif cfg.spk_fuse_type == "FiLM":          # norm.py:84-137, one layer: gamma and beta Linears embed -> feature
        s["spk_fuse.fc.gamma_fcs.0.weight"], s["spk_fuse.fc.gamma_fcs.0.bias"] = (Fd, E), (Fd,)
        s["spk_fuse.fc.beta_fcs.0.weight"], s["spk_fuse.fc.beta_fcs.0.bias"] = (Fd, E), (Fd,)
    else:
        raise NotImplementedError(cfg.spk_fuse_type)
    D = cfg.tcn_dims
    for l in range(cfg.tcn_layers):
        for b in range(cfg.tcn_blocks):
            q = f"tcn_layers.{l}.{b}."
            s[q + "dconv1.weight"] = (D, 1, 3)
            s[q + "dconv1.bias"] = (D,)
            s[q + "dconv2.weight"] = (D, D, 1)
            s[q + "dconv2.bias"] = (D,)
    for j, (ci, co) in zip((0, 1, 2), ((768, 128), (256, 64), (128, 32))):
        s[f"decoder.{j}.convtrans2d.weight"] = (ci, co, 3, 3)
        s[f"decoder.{j}.convtrans2d.bias"] = (co,)
    for i in range(4):
        _dense_shapes(s, f"decoder.{i + 3}.0.", 32, 64, "dec")
        co = 32 if i != 3 else 16
        s[f"decoder.{i + 3}.1.convtrans2d.weight"] = (64, co, 3, 3)
        s[f"decoder.{i + 3}.1.convtrans2d.bias"] = (co,)
    _dense_shapes(s, "decoder.7.", 16, 32, "dec")
    for i in range(len(cfg.pool_size)):
        s[f"avg_pool.{i}.1.weight"] = (8, 32, 1, 1)
        s[f"avg_pool.{i}.1.bias"] = (8,)
    s["avg_proj.weight"] = (32, 32 + 8 * len(cfg.pool_size), 1, 1)
    s["avg_proj.bias"] = (32,)
    s["deconv2d.weight"] = (32, 2, 3, 3)
    s["deconv2d.bias"] = (2,)
    return s


def synth_params(cfg: DPCCNConfig, seed: int) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, shp in param_shapes(cfg).items():
        if k.endswith("bias"):
            out[k] = 0.05 * torch.randn(shp, generator=g)
        else:
            fan = 1
            for d in shp[1:]:
                fan *= d
            if "convtrans" in k or k.startswith("deconv2d"):
                fan = shp[0] * shp[2] * shp[3]
            out[k] = torch.randn(shp, generator=g) / fan ** 0.5
    return out


def _cblock(p, q, x, stride=(1, 1)):
    """Conv2dBlock (convs.py:28-50): conv - ELU - InstanceNorm2d (no affine, eps 1e-5)."""
    return F.instance_norm(F.elu(F.conv2d(x, p[q + "conv2d.weight"], p[q + "conv2d.bias"], stride, (1, 1))))


def _tblock(p, q, x, stride=(1, 2)):
    """ConvTrans2dBlock (convs.py:53-77), padding (1, 0)... the model passes padding=(1, 1) (dpccn.py:113-117)."""
    return F.instance_norm(F.elu(F.conv_transpose2d(x, p[q + "convtrans2d.weight"], p[q + "convtrans2d.bias"], stride,
                                                    (1, 1), (0, 0))))


def _dense(p, q, x):
    """DenseBlock (convs.py:80-112)."""
    feats = [x]
    for i in range(1, 6):
        y = _cblock(p, f"{q}conv{i}.", torch.cat(feats, 1))
        feats.append(y)
    return feats[-1]


def _tcn(p, q, x, dil, causal=False):
    """TCNBlock (convs.py:115-152)."""
    D = x.shape[1]
    y = F.elu(F.instance_norm(x))
    if causal:
        y = F.conv1d(y, p[q + "dconv1.weight"], p[q + "dconv1.bias"], padding=2 * dil, dilation=dil, groups=D)[:, :, :-2 * dil]
    else:
        y = F.conv1d(y, p[q + "dconv1.weight"], p[q + "dconv1.bias"], padding=dil, dilation=dil, groups=D)
    y = F.elu(F.instance_norm(y))
    return x + F.conv1d(y, p[q + "dconv2.weight"], p[q + "dconv2.bias"])


def _fuse(p, cfg, x, e):
    """SpeakerFuseLayer (speaker.py:81-125) on x [B, C, F, T] with e [B, 1, E, 1]: the Linear acts on dim 2."""
    if cfg.spk_fuse_type == "FiLM":            # x = (1 + gamma(e)) x + beta(e), gamma / beta per frequency bin (norm.py:116-134)
        ev = e.squeeze(-1)                                                                       # [B, 1, E]
        gm = F.linear(ev, p["spk_fuse.fc.gamma_fcs.0.weight"], p["spk_fuse.fc.gamma_fcs.0.bias"]).unsqueeze(-1)
        bt = F.linear(ev, p["spk_fuse.fc.beta_fcs.0.weight"], p["spk_fuse.fc.beta_fcs.0.bias"]).unsqueeze(-1)
        return (1 + gm) * x + bt
    w, b = p["spk_fuse.fc.linear.weight"], p["spk_fuse.fc.linear.bias"]
    if cfg.spk_fuse_type == "concat":
        ee = e.expand(-1, x.shape[1], -1, x.shape[3])
        y = torch.cat([x, ee], 2)
        return F.linear(y.transpose(2, 3), w, b).transpose(2, 3)
    t = F.linear(e.transpose(2, 3), w, b).transpose(2, 3)          # [B, 1, F, 1]
    return x + t if cfg.spk_fuse_type == "additive" else x * t


def dpccn_forward(p: Dict[str, torch.Tensor], cfg: DPCCNConfig, wav: torch.Tensor, emb: torch.Tensor) -> torch.Tensor:
    """`dpccn.py:206-290`, joint_training=False: wav [B, T], emb [B, E] -> est [B, T]."""
    B, nsample = wav.shape
    win = torch.hann_window(cfg.win)
    spec = torch.stft(wav, cfg.win, cfg.stride, window=win, return_complex=True)
    x = torch.stack([spec.real, spec.imag], 1).transpose(2, 3)               # [B, 2, T, F]
    out = F.conv2d(x, p["conv2d.weight"], p["conv2d.bias"], (1, 1), (1, 1))
    out = _dense(p, "encoder.0.", out)
    if cfg.use_spk_transform:
        emb = spk_transform(p, emb)
    e = emb.unsqueeze(1).unsqueeze(3)
    out = _fuse(p, cfg, out.transpose(2, 3), e).transpose(2, 3)
    skips = [out]
    for i in range(4):
        out = _dense(p, f"encoder.{i + 1}.1.", _cblock(p, f"encoder.{i + 1}.0.", out, (1, 2)))
        skips.append(out)
    for j in (5, 6, 7):
        out = _cblock(p, f"encoder.{j}.", out, (1, 2))
        skips.append(out)
    Bn, N, T, Fq = out.shape
    y = out.reshape(Bn, N, T * Fq)
    for l in range(cfg.tcn_layers):
        for b in range(cfg.tcn_blocks):
            y = _tcn(p, f"tcn_layers.{l}.{b}.", y, 2 ** b, cfg.causal)
    out = y.reshape(Bn, N, T, Fq)
    skips = skips[::-1]
    for j in range(3):
        out = _tblock(p, f"decoder.{j}.", torch.cat([skips[j], out], 1))
    for i in range(4):
        out = _tblock(p, f"decoder.{i + 3}.1.", _dense(p, f"decoder.{i + 3}.0.", torch.cat([skips[i + 3], out], 1)))
    out = _dense(p, "decoder.7.", torch.cat([skips[7], out], 1))
    Bn, N, T, Fq = out.shape
    pools = []
    for i, sz in enumerate(cfg.pool_size):
        a = F.conv2d(F.avg_pool2d(out, sz), p[f"avg_pool.{i}.1.weight"], p[f"avg_pool.{i}.1.bias"])
        pools.append(F.interpolate(a, size=(T, Fq), mode="bilinear"))
    out = F.conv2d(torch.cat([out, *pools], 1), p["avg_proj.weight"], p["avg_proj.bias"])
    out = F.conv_transpose2d(out, p["deconv2d.weight"], p["deconv2d.bias"], (1, 1), (1, 1))
    est = out.transpose(2, 3)                                               # [B, 2, F, T]
    est = torch.complex(est[:, 0], est[:, 1])
    return torch.istft(est, cfg.win, cfg.stride, window=win, length=nsample)
