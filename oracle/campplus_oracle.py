"""TEST INFRASTRUCTURE ONLY -- CPU fp32 restatement of the wespeaker CAM++ speaker encoder (`CAMPPlus`; SURVEY.md
section 8 row a12: the shipped recipe lists it as an alternative encoder, examples/librimix/tse/v2/confs/bsrnn.yaml:66-74;
instantiated through `get_speaker_model`, wesep/models/bsrnn.py:9,217).

**Parity unpinned.**  `wespeaker` is a third-party package that is neither vendored under /root/reference nor pinned
nor installed here, so there is no reference output to generate fixtures from.  This file restates the published
architecture (wespeaker/models/campplus.py + campplus_layers.py as recalled, after 3D-Speaker; Wang et al., "CAM++: A
Fast and Efficient Network for Speaker Verification Using Context-Aware Masking", Interspeech 2023):

  x [B, T, F] -> [B, F, T]
  head     FCM: Conv2d(1 -> 32, 3x3) BN ReLU; two stages of two BasicResBlocks (first of each: stride (2, 1) over the mel
           axis, with a 1x1 conv + BN shortcut); Conv2d(32 -> 32, 3x3, stride (2, 1)) BN ReLU; [B, 32, F/8, T] ->
           [B, 32 * F/8, T]
  xvector  tdnn       Conv1d(320 -> 128, k 5, stride 2, pad 2, no bias) BN ReLU
           block1-3   12 / 24 / 16 CAM-dense-TDNN layers (kernel 3, dilation 1 / 2 / 2, growth 32), each
                      BN ReLU Conv1d(cin -> 128, 1) BN ReLU CAMLayer, its 32 outputs concatenated to its input;
                      CAMLayer: y = Conv1d(128 -> 32, k 3, dilated); context = mean_T(x) + segment average (100
                      frames, ceil mode, last segment shorter); mask = sigmoid(W2 relu(W1 context)); y * mask
           transit1-3 BN ReLU Conv1d(c -> c / 2, 1, no bias)
           out_nonlinear BN ReLU; stats TSTP (mean || sqrt(unbiased var + 1e-7)); dense Conv1d(1024 -> embed, 1, no
           bias) BatchNorm1d(affine = False)

A sanity anchor: the parameter count of this restatement for feat_dim 80 / embed_dim 512 is 7.18 M, the figure
published for CAM++.  The HIP path is tested against this file; parameter names follow the upstream module tree so
that wespeaker checkpoints load.  Only tests/ may import this module."""
from typing import Dict

import torch
import torch.nn.functional as F

BN_EPS, BN_MOMENTUM = 1e-5, 0.1
BLOCKS = ((12, 3, 1), (24, 3, 2), (16, 3, 2))       # (layers, kernel, dilation)
SEG_LEN = 100


def param_shapes(feat_dim=80, embed_dim=512, growth_rate=32, bn_size=4, init_channels=128, m_channels=32,
                 blocks=BLOCKS) -> Dict[str, tuple]:
    s = {}

    def bn(name, c, affine=True):
        if affine:
            s[name + ".weight"], s[name + ".bias"] = (c,), (c,)
        s[name + ".running_mean"], s[name + ".running_var"], s[name + ".num_batches_tracked"] = (c,), (c,), ()

    m = m_channels
    s["head.conv1.weight"] = (m, 1, 3, 3)
    bn("head.bn1", m)
    for li in (1, 2):
        for bi in (0, 1):
            q = f"head.layer{li}.{bi}."
            s[q + "conv1.weight"] = (m, m, 3, 3)
            bn(q + "bn1", m)
            s[q + "conv2.weight"] = (m, m, 3, 3)
            bn(q + "bn2", m)
            if bi == 0:
                s[q + "shortcut.0.weight"] = (m, m, 1, 1)
                bn(q + "shortcut.1", m)
    s["head.conv2.weight"] = (m, m, 3, 3)
    bn("head.bn2", m)
    c = m * (feat_dim // 8)
    s["xvector.tdnn.linear.weight"] = (init_channels, c, 5)
    bn("xvector.tdnn.nonlinear.batchnorm", init_channels)
    c = init_channels
    bnc = bn_size * growth_rate
    for bi, (layers, k, _) in enumerate(blocks):
        for i in range(layers):
            q = f"xvector.block{bi + 1}.tdnnd{i + 1}."
            cin = c + i * growth_rate
            bn(q + "nonlinear1.batchnorm", cin)
            s[q + "linear1.weight"] = (bnc, cin, 1)
            bn(q + "nonlinear2.batchnorm", bnc)
            s[q + "cam_layer.linear_local.weight"] = (growth_rate, bnc, k)
            s[q + "cam_layer.linear1.weight"], s[q + "cam_layer.linear1.bias"] = (bnc // 2, bnc, 1), (bnc // 2,)
            s[q + "cam_layer.linear2.weight"], s[q + "cam_layer.linear2.bias"] = (growth_rate, bnc // 2, 1), (growth_rate,)
        c += layers * growth_rate
        bn(f"xvector.transit{bi + 1}.nonlinear.batchnorm", c)
        s[f"xvector.transit{bi + 1}.linear.weight"] = (c // 2, c, 1)
        c //= 2
    bn("xvector.out_nonlinear.batchnorm", c)
    s["xvector.dense.linear.weight"] = (embed_dim, 2 * c, 1)
    bn("xvector.dense.nonlinear.batchnorm", embed_dim, affine=False)
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
        elif len(shp) == 1 and ("bn" in k.split(".")[-2] or "batchnorm" in k or k.split(".")[-2] == "1"):
            v = (1.0 + 0.1 * torch.randn(shp, generator=g)) if k.endswith("weight") else 0.1 * torch.randn(shp, generator=g)
        elif k.endswith("bias"):
            v = 0.05 * torch.randn(shp, generator=g)
        else:
            fan_in = 1
            for d in shp[1:]:
                fan_in *= d
            v = torch.randn(shp, generator=g) * (2.0 / fan_in) ** 0.5       # ReLU-preserving scale (upstream: kaiming)
        out[k] = v
    return out


def seg_pooling(x, seg_len=SEG_LEN):
    """[B, C, T] -> per-frame average of the frame's `seg_len` segment (the last segment may be shorter)."""
    seg = F.avg_pool1d(x, kernel_size=seg_len, stride=seg_len, ceil_mode=True)
    shape = seg.shape
    seg = seg.unsqueeze(-1).expand(*shape, seg_len).reshape(*shape[:-1], -1)
    return seg[..., :x.shape[-1]]


def campplus_forward(p, x, blocks=BLOCKS, training=True, new_buffers=None, prefix="", relu_masks=None):
    """x [B, T, F] -> embedding [B, embed_dim].
    relu_masks: optional list of boolean tensors, one per BatchNorm-side ReLU in evaluation order (head: [B, C, F', T];
    backbone: [B, C, T']): those ReLUs then multiply by the given mask instead of by (z > 0), so that a test can
    differentiate the restatement on the SAME linear region as the implementation under test (a pre-activation within
    rounding distance of zero otherwise flips its whole downstream gradient: a kink, not an arithmetic error).  The small
    ReLU inside the context-aware mask ([B, 64, segments]) is not masked."""
    masks = list(relu_masks) if relu_masks is not None else None

    def relu(z):
        if masks is None:
            return F.relu(z)
        mk = masks.pop(0)
        assert mk.shape == z.shape, (mk.shape, z.shape)
        return z * mk.to(z.dtype)

    def bn(name, y, affine=True):
        name = prefix + name
        rm, rv = p[name + ".running_mean"].clone(), p[name + ".running_var"].clone()
        out = F.batch_norm(y, rm, rv, p[name + ".weight"] if affine else None, p[name + ".bias"] if affine else None,
                           training, BN_MOMENTUM, BN_EPS)
        if new_buffers is not None and training:
            new_buffers[name + ".running_mean"], new_buffers[name + ".running_var"] = rm, rv
        return out

    def w(name):
        return p[prefix + name]

    y = x.permute(0, 2, 1).unsqueeze(1)                                   # [B, 1, F, T]
    y = relu(bn("head.bn1", F.conv2d(y, w("head.conv1.weight"), padding=1)))
    for li in (1, 2):
        for bi in (0, 1):
            q = f"head.layer{li}.{bi}."
            stride = (2, 1) if bi == 0 else (1, 1)
            o = relu(bn(q + "bn1", F.conv2d(y, w(q + "conv1.weight"), stride=stride, padding=1)))
            o = bn(q + "bn2", F.conv2d(o, w(q + "conv2.weight"), padding=1))
            sc = bn(q + "shortcut.1", F.conv2d(y, w(q + "shortcut.0.weight"), stride=stride)) if bi == 0 else y
            y = relu(o + sc)
    y = relu(bn("head.bn2", F.conv2d(y, w("head.conv2.weight"), stride=(2, 1), padding=1)))
    y = y.reshape(y.shape[0], y.shape[1] * y.shape[2], y.shape[3])        # [B, 32 * F/8, T], channel = c * F/8 + f
    y = relu(bn("xvector.tdnn.nonlinear.batchnorm", F.conv1d(y, w("xvector.tdnn.linear.weight"), stride=2, padding=2)))
    for bi, (layers, k, dil) in enumerate(blocks):
        for i in range(layers):
            q = f"xvector.block{bi + 1}.tdnnd{i + 1}."
            h = F.conv1d(relu(bn(q + "nonlinear1.batchnorm", y)), w(q + "linear1.weight"))
            h = relu(bn(q + "nonlinear2.batchnorm", h))
            local = F.conv1d(h, w(q + "cam_layer.linear_local.weight"), padding=(k - 1) // 2 * dil, dilation=dil)
            ctx = h.mean(-1, keepdim=True) + seg_pooling(h)
            ctx = F.relu(F.conv1d(ctx, w(q + "cam_layer.linear1.weight"), w(q + "cam_layer.linear1.bias")))
            mask = torch.sigmoid(F.conv1d(ctx, w(q + "cam_layer.linear2.weight"), w(q + "cam_layer.linear2.bias")))
            y = torch.cat([y, local * mask], 1)
        q = f"xvector.transit{bi + 1}."
        y = F.conv1d(relu(bn(q + "nonlinear.batchnorm", y)), w(q + "linear.weight"))
    y = relu(bn("xvector.out_nonlinear.batchnorm", y))
    stats = torch.cat([y.mean(-1), torch.sqrt(torch.var(y, dim=-1) + 1e-7)], 1)
    emb = F.conv1d(stats.unsqueeze(-1), w("xvector.dense.linear.weight")).squeeze(-1)
    return bn("xvector.dense.nonlinear.batchnorm", emb, affine=False)
