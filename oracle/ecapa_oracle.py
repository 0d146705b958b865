"""TEST INFRASTRUCTURE ONLY -- CPU fp32 restatement of the wespeaker ECAPA-TDNN speaker encoder (SURVEY.md section 8
rows a12 / f-4: the reference's published checkpoint is `bsrnn_ecapa_vox1`, wesep/cli/hub.py:86-95; the shipped recipe
lists `ECAPA_TDNN_GLOB_c512` with ASTP pooling and a 192-d embedding as the alternative encoder,
examples/librimix/tse/v2/confs/bsrnn.yaml:66-71; instantiated through `get_speaker_model`, wesep/models/bsrnn.py:9,217).

**Parity unpinned.**  `wespeaker` is a third-party package that is neither vendored under /root/reference nor pinned
nor installed here, so there is no reference output to generate fixtures from.  This file restates the published
architecture (wespeaker/models/ecapa_tdnn.py, pooling_layers.py as recalled; Desplanques et al., Interspeech 2020):

  x [B, T, F] -> [B, F, T]
  layer1   Conv1d(F -> C, k 5, pad 2) -> ReLU -> BatchNorm1d                       (every block is Conv -> ReLU -> BN)
  layer2-4 SE_Res2Block(dilation 2 / 3 / 4): x + SE(CRB_1x1(Res2(CRB_1x1(x)))), Res2 with scale 8 (7 dilated k 3
           convolutions on C / 8 channels, group i >= 1 adds the previous group's output first, the last group passes
           through); SE: sigmoid(W2 relu(W1 mean_T(x))) per channel
  conv     Conv1d(3C -> 1536, k 1) on cat(out2, out3, out4) -> ReLU
  pool     ASTP with global context: alpha = softmax_T(W2 tanh(W1 cat(x, mean_T, sqrt(var_T + 1e-7))));
           mean = sum alpha x, std = sqrt(clamp(sum alpha x^2 - mean^2, 1e-7)), concatenated
  bn       BatchNorm1d(3072) -> linear Linear(3072 -> embed_dim) (-> bn2 when emb_bn)

A sanity anchor: the parameter count of this restatement for ECAPA_TDNN_GLOB_c512 (feat 80, embed 192) is 6.19 M, the
figure wespeaker publishes for that model.  The HIP path is tested against this file; parameter names follow the
upstream module tree so that wespeaker checkpoints load.  Only tests/ may import this module."""
from typing import Dict

import torch
import torch.nn.functional as F

BN_EPS, BN_MOMENTUM = 1e-5, 0.1


def param_shapes(channels=512, feat_dim=80, embed_dim=192, global_context_att=True, emb_bn=False, scale=8) -> Dict[str, tuple]:
    s = {}

    def bn(name, c):
        s[name + ".weight"], s[name + ".bias"] = (c,), (c,)
        s[name + ".running_mean"], s[name + ".running_var"], s[name + ".num_batches_tracked"] = (c,), (c,), ()

    def crb(name, cin, cout, k):
        s[name + ".conv.weight"], s[name + ".conv.bias"] = (cout, cin, k), (cout,)
        bn(name + ".bn", cout)

    crb("layer1", feat_dim, channels, 5)
    w = channels // scale
    for li in (2, 3, 4):
        q = f"layer{li}.se_res2block."
        crb(q + "0", channels, channels, 1)
        for i in range(scale - 1):
            s[q + f"1.convs.{i}.weight"], s[q + f"1.convs.{i}.bias"] = (w, w, 3), (w,)
            bn(q + f"1.bns.{i}", w)
        crb(q + "2", channels, channels, 1)
        s[q + "3.linear1.weight"], s[q + "3.linear1.bias"] = (128, channels), (128,)
        s[q + "3.linear2.weight"], s[q + "3.linear2.bias"] = (channels, 128), (channels,)
    s["conv.weight"], s["conv.bias"] = (1536, 3 * channels, 1), (1536,)
    s["pool.linear1.weight"], s["pool.linear1.bias"] = (128, 1536 * (3 if global_context_att else 1), 1), (128,)
    s["pool.linear2.weight"], s["pool.linear2.bias"] = (1536, 128, 1), (1536,)
    bn("bn", 3072)
    s["linear.weight"], s["linear.bias"] = (embed_dim, 3072), (embed_dim,)
    if emb_bn:
        bn("bn2", embed_dim)
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
        elif ".bn." in k or ".bns." in k or k.startswith(("bn.", "bn2.")):
            v = (1.0 + 0.1 * torch.randn(shp, generator=g)) if k.endswith("weight") else 0.1 * torch.randn(shp, generator=g)
        elif k.endswith("bias"):
            v = 0.05 * torch.randn(shp, generator=g)
        else:
            fan_in = 1
            for d in shp[1:]:
                fan_in *= d
            v = torch.randn(shp, generator=g) * (1.0 / fan_in) ** 0.5
        out[k] = v
    return out


def ecapa_forward(p, x, global_context_att=True, emb_bn=False, scale=8, training=True, new_buffers=None, prefix=""):
    """x [B, T, F] -> embedding [B, embed_dim]."""
    def bn(name, y):
        rm, rv = p[name + ".running_mean"].clone(), p[name + ".running_var"].clone()
        out = F.batch_norm(y, rm, rv, p[name + ".weight"], p[name + ".bias"], training, BN_MOMENTUM, BN_EPS)
        if new_buffers is not None and training:
            new_buffers[name + ".running_mean"], new_buffers[name + ".running_var"] = rm, rv
        return out

    def crb(name, y, dil=1):
        w = p[name + ".conv.weight"]
        k = w.shape[2]
        return bn(name + ".bn", F.relu(F.conv1d(y, w, p[name + ".conv.bias"], padding=dil * (k // 2), dilation=dil)))

    y = x.permute(0, 2, 1)
    out1 = crb(prefix + "layer1", y)
    outs, cur = [], out1
    for li, dil in ((2, 2), (3, 3), (4, 4)):
        q = prefix + f"layer{li}.se_res2block."
        h = crb(q + "0", cur)
        width = h.shape[1] // scale
        spx = torch.split(h, width, 1)
        parts, sp = [], None
        for i in range(scale - 1):
            sp = spx[i] if i == 0 else sp + spx[i]
            sp = F.conv1d(sp, p[q + f"1.convs.{i}.weight"], p[q + f"1.convs.{i}.bias"], padding=dil, dilation=dil)
            sp = bn(q + f"1.bns.{i}", F.relu(sp))
            parts.append(sp)
        parts.append(spx[scale - 1])
        h = crb(q + "2", torch.cat(parts, 1))
        g = F.relu(F.linear(h.mean(2), p[q + "3.linear1.weight"], p[q + "3.linear1.bias"]))
        g = torch.sigmoid(F.linear(g, p[q + "3.linear2.weight"], p[q + "3.linear2.bias"]))
        cur = cur + h * g.unsqueeze(2)
        outs.append(cur)
    h = F.relu(F.conv1d(torch.cat(outs, 1), p[prefix + "conv.weight"], p[prefix + "conv.bias"]))
    if global_context_att:
        mean = h.mean(-1, keepdim=True).expand_as(h)
        std = torch.sqrt(torch.var(h, dim=-1, keepdim=True) + 1e-7).expand_as(h)
        a_in = torch.cat((h, mean, std), 1)
    else:
        a_in = h
    a = torch.tanh(F.conv1d(a_in, p[prefix + "pool.linear1.weight"], p[prefix + "pool.linear1.bias"]))
    alpha = torch.softmax(F.conv1d(a, p[prefix + "pool.linear2.weight"], p[prefix + "pool.linear2.bias"]), dim=2)
    mean = torch.sum(alpha * h, dim=2)
    var = torch.sum(alpha * h ** 2, dim=2) - mean ** 2
    stats = torch.cat([mean, torch.sqrt(var.clamp(min=1e-7))], 1)
    emb = F.linear(bn(prefix + "bn", stats), p[prefix + "linear.weight"], p[prefix + "linear.bias"])
    if emb_bn:
        emb = bn(prefix + "bn2", emb)
    return emb
