"""TEST INFRASTRUCTURE ONLY -- CPU fp32 restatement of the wespeaker ResNet speaker encoder that wesep trains
jointly with the separator.

**Parity unpinned.**  The model lives in the third-party package `wespeaker` (`wespeaker.models.speaker_model
.get_speaker_model`, imported at `wesep/models/bsrnn.py:9`, instantiated at `:217` with `spk_args` from
`confs/bsrnn.yaml:58-64`: ResNet34, feat_dim 80, embed_dim 256, pooling TSTP, two_emb_layer False); the package is
neither vendored under /root/reference nor pinned (`requirements.txt` does not list it) and is absent from this
image, so there is no reference output to generate fixtures from.  This file restates the published architecture
(wespeaker/models/resnet.py, pooling_layers.py -- from the upstream source as recalled):

  x [B, T, F] -> permute -> [B, 1, F, T]
  conv1 3x3(1 -> m) + BN + ReLU;  layer1..4 of BasicBlock(planes m, 2m, 4m, 8m; first stride 1, 2, 2, 2):
      BasicBlock: conv3x3(stride) - BN - ReLU - conv3x3 - BN, + shortcut (1x1 conv(stride) + BN when the shape
      changes), ReLU;  all convolutions bias-free
  TSTP: mean over T and sqrt(unbiased var over T + 1e-7) of [B, C*F', T'], concatenated
  seg_1 Linear(2 * C * F' -> embed_dim);  two_emb_layer False: returns (tensor(0.), embed_a)

The HIP path is tested against this restatement; parameter names follow the upstream module tree so that wespeaker
checkpoints (`spk_model_init`) load.  Only tests/ may import this module."""
from typing import Dict

import torch
import torch.nn.functional as F

BN_EPS, BN_MOMENTUM = 1e-5, 0.1
NUM_BLOCKS = {"ResNet18": (2, 2, 2, 2), "ResNet34": (3, 4, 6, 3), "ResNet50": (3, 4, 6, 3), "ResNet101": (3, 4, 23, 3),
              "ResNet152": (3, 8, 36, 3)}
BOTTLENECK = ("ResNet50", "ResNet101", "ResNet152")   # Bottleneck blocks (1x1 - 3x3(stride) - 1x1, expansion 4)


def _blocks(num_blocks, m, expansion=1):
    """(prefix, in_planes, planes, stride) of every block in order; a block outputs expansion * planes channels."""
    out, inp = [], m
    for li, (n, planes, stride) in enumerate(zip(num_blocks, (m, 2 * m, 4 * m, 8 * m), (1, 2, 2, 2)), start=1):
        for bi in range(n):
            out.append((f"layer{li}.{bi}.", inp, planes, stride if bi == 0 else 1))
            inp = planes * expansion
    return out


def _bn_shapes(s, name, c):
    s[name + ".weight"] = (c,)
    s[name + ".bias"] = (c,)
    s[name + ".running_mean"] = (c,)
    s[name + ".running_var"] = (c,)
    s[name + ".num_batches_tracked"] = ()


def param_shapes(num_blocks=(3, 4, 6, 3), m=32, feat_dim=80, embed_dim=256, prefix="", bottleneck=False,
                 two_emb_layer=False, pooling="TSTP") -> Dict[str, tuple]:
    s: Dict[str, tuple] = {}
    ex = 4 if bottleneck else 1
    s[prefix + "conv1.weight"] = (m, 1, 3, 3)
    _bn_shapes(s, prefix + "bn1", m)
    for q, inp, planes, stride in _blocks(num_blocks, m, ex):
        q = prefix + q
        if bottleneck:
            s[q + "conv1.weight"] = (planes, inp, 1, 1)
            _bn_shapes(s, q + "bn1", planes)
            s[q + "conv2.weight"] = (planes, planes, 3, 3)
            _bn_shapes(s, q + "bn2", planes)
            s[q + "conv3.weight"] = (ex * planes, planes, 1, 1)
            _bn_shapes(s, q + "bn3", ex * planes)
        else:
            s[q + "conv1.weight"] = (planes, inp, 3, 3)
            _bn_shapes(s, q + "bn1", planes)
            s[q + "conv2.weight"] = (planes, planes, 3, 3)
            _bn_shapes(s, q + "bn2", planes)
        if stride != 1 or inp != ex * planes:
            s[q + "shortcut.0.weight"] = (ex * planes, inp, 1, 1)
            _bn_shapes(s, q + "shortcut.1", ex * planes)
    stats_dim = (feat_dim // 8) * m * 8 * ex
    if pooling == "ASTP":      # pooling_layers.ASTP(in_dim, bottleneck_dim=128): two 1x1 convolutions
        s[prefix + "pool.linear1.weight"], s[prefix + "pool.linear1.bias"] = (128, stats_dim, 1), (128,)
        s[prefix + "pool.linear2.weight"], s[prefix + "pool.linear2.bias"] = (stats_dim, 128, 1), (stats_dim,)
    s[prefix + "seg_1.weight"] = (embed_dim, (1 if pooling in ("TAP", "TSDP") else 2) * stats_dim)
    s[prefix + "seg_1.bias"] = (embed_dim,)
    if two_emb_layer:      # seg_bn_1 = BatchNorm1d(embed_dim, affine=False): buffers only
        s[prefix + "seg_bn_1.running_mean"], s[prefix + "seg_bn_1.running_var"] = (embed_dim,), (embed_dim,)
        s[prefix + "seg_bn_1.num_batches_tracked"] = ()
        s[prefix + "seg_2.weight"], s[prefix + "seg_2.bias"] = (embed_dim, embed_dim), (embed_dim,)
    return s


def is_buffer(name: str) -> bool:
    return name.endswith(("running_mean", "running_var", "num_batches_tracked"))


def synth_params(seed: int, **kw) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, shp in param_shapes(**kw).items():
        if k.endswith("running_mean"):
            v = torch.zeros(shp)
        elif k.endswith("running_var"):
            v = torch.ones(shp)
        elif k.endswith("num_batches_tracked"):
            v = torch.zeros(shp, dtype=torch.long)
        elif ".bn" in k or "bn1." in k or "shortcut.1" in k or "seg_bn_1" in k:
            v = (1.0 + 0.1 * torch.randn(shp, generator=g)) if k.endswith("weight") else 0.1 * torch.randn(shp, generator=g)
        elif k.endswith("bias"):
            v = 0.05 * torch.randn(shp, generator=g)
        else:
            fan_in = 1
            for d in shp[1:]:
                fan_in *= d
            v = torch.randn(shp, generator=g) * (2.0 / fan_in) ** 0.5
        out[k] = v
    return out


def resnet_forward(p, x, num_blocks=(3, 4, 6, 3), m=32, prefix="", training=True, new_buffers=None, relu_masks=None,
                   bottleneck=False, two_emb_layer=False, pooling="TSTP"):
    """x [B, T, F] -> embed_a [B, embed_dim]  (two_emb_layer: (embed_a, embed_b), embed_b = seg_2(BN(relu(embed_a)))).
    relu_masks: optional list of boolean [B, C, F', T'] tensors, one per ReLU in evaluation order (stem, then per block:
    after bn1, after the residual sum): the ReLU then multiplies by the given mask instead of by (z > 0).  Tests use it
    to differentiate the restatement on the SAME linear region as the implementation under test -- a pre-activation
    within rounding distance of zero otherwise flips its whole downstream gradient (a ReLU kink, not an arithmetic
    error)."""
    masks = list(relu_masks) if relu_masks is not None else None

    def relu(z):
        if masks is None:
            return F.relu(z)
        mk = masks.pop(0)
        assert mk.shape == z.shape, (mk.shape, z.shape)
        return z * mk.to(z.dtype)

    def bn(name, y):
        rm, rv = p[name + ".running_mean"].clone(), p[name + ".running_var"].clone()
        out = F.batch_norm(y, rm, rv, p[name + ".weight"], p[name + ".bias"], training, BN_MOMENTUM, BN_EPS)
        if new_buffers is not None and training:
            new_buffers[name + ".running_mean"], new_buffers[name + ".running_var"] = rm, rv
        return out
    y = x.permute(0, 2, 1).unsqueeze(1)
    y = relu(bn(prefix + "bn1", F.conv2d(y, p[prefix + "conv1.weight"], padding=1)))
    for q, inp, planes, stride in _blocks(num_blocks, m, 4 if bottleneck else 1):
        q = prefix + q
        if bottleneck:
            o = relu(bn(q + "bn1", F.conv2d(y, p[q + "conv1.weight"])))
            o = relu(bn(q + "bn2", F.conv2d(o, p[q + "conv2.weight"], stride=stride, padding=1)))
            o = bn(q + "bn3", F.conv2d(o, p[q + "conv3.weight"]))
        else:
            o = relu(bn(q + "bn1", F.conv2d(y, p[q + "conv1.weight"], stride=stride, padding=1)))
            o = bn(q + "bn2", F.conv2d(o, p[q + "conv2.weight"], padding=1))
        sc = y
        if (q + "shortcut.0.weight") in p:
            sc = bn(q + "shortcut.1", F.conv2d(y, p[q + "shortcut.0.weight"], stride=stride))
        y = relu(o + sc)
    # pooling_layers.py: TSTP mean || std, TAP mean, TSDP std (unbiased variance + 1e-7), ASTP attentive statistics on
    # the [B, C * F', T] view (softmax over T of W2 tanh(W1 x); weighted mean || sqrt(clamp(weighted var, 1e-7)))
    if pooling == "ASTP":
        h = y.reshape(y.shape[0], y.shape[1] * y.shape[2], y.shape[3])
        a = torch.tanh(F.conv1d(h, p[prefix + "pool.linear1.weight"], p[prefix + "pool.linear1.bias"]))
        alpha = torch.softmax(F.conv1d(a, p[prefix + "pool.linear2.weight"], p[prefix + "pool.linear2.bias"]), dim=2)
        mean = torch.sum(alpha * h, dim=2)
        var = torch.sum(alpha * h ** 2, dim=2) - mean ** 2
        stats = torch.cat([mean, torch.sqrt(var.clamp(min=1e-7))], 1)
    else:
        mean = y.mean(-1)
        std = torch.sqrt(torch.var(y, dim=-1) + 1e-7)
        stats = {"TSTP": torch.cat((mean.flatten(1), std.flatten(1)), 1), "TAP": mean.flatten(1),
                 "TSDP": std.flatten(1)}[pooling]
    embed_a = F.linear(stats, p[prefix + "seg_1.weight"], p[prefix + "seg_1.bias"])
    if not two_emb_layer:
        return embed_a
    q = prefix + "seg_bn_1"
    rm, rv = p[q + ".running_mean"].clone(), p[q + ".running_var"].clone()
    o = F.batch_norm(F.relu(embed_a), rm, rv, None, None, training, BN_MOMENTUM, BN_EPS)
    if new_buffers is not None and training:
        new_buffers[q + ".running_mean"], new_buffers[q + ".running_var"] = rm, rv
    return embed_a, F.linear(o, p[prefix + "seg_2.weight"], p[prefix + "seg_2.bias"])


# ---- in-model fbank front-end (bsrnn.py:231-242,343-350) ----------------------------
def melscale_fbanks(n_freqs, f_min, f_max, n_mels, sample_rate):
    """torchaudio.functional.melscale_fbanks(norm=None, mel_scale="htk"), restated (torchaudio is absent: unpinned)."""
    import math
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * math.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * math.log10(1.0 + f_max / 700.0)
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    return torch.max(torch.zeros(1), torch.min((-1.0 * slopes[:, :-2]) / f_diff[:-1], slopes[:, 2:] / f_diff[1:]))


def fbank_frontend(wav, sr=16000, n_fft=512, hop=128, n_mels=80, f_min=20.0, coef=0.97):
    """PreEmphasis (speaker.py:10-23, reference code) -> MelSpectrogram (torchaudio defaults: centre, reflect,
    power 2, HTK, no norm) -> log(+1e-8) -> minus the time mean -> [R, Tf, n_mels]  (bsrnn.py:343-350)."""
    x = F.pad(wav.unsqueeze(1), (1, 0), "reflect")
    y = F.conv1d(x, torch.tensor([[[-coef, 1.0]]])).squeeze(1)
    spec = torch.stft(y, n_fft, hop, n_fft, window=torch.hamming_window(n_fft), center=True, pad_mode="reflect",
                      normalized=False, onesided=True, return_complex=True).abs().pow(2.0)
    fb = melscale_fbanks(n_fft // 2 + 1, f_min, float(sr // 2), n_mels, sr)
    mel = torch.matmul(spec.transpose(-1, -2), fb).transpose(-1, -2)       # [R, n_mels, Tf]
    feat = (mel + 1e-8).log()
    feat = feat - feat.mean(dim=-1, keepdim=True)
    return feat.permute(0, 2, 1)
