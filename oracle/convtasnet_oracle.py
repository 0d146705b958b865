"""TEST INFRASTRUCTURE ONLY -- CPU fp32 restatement of the wesep Conv-TasNet / SpEx+ path
(BASELINE.json configs[0]).

Functional re-statement (plain torch CPU ops, parameters passed as a dict keyed by the
reference's `state_dict` names) of, in fixed-embedding mode (`joint_training=False`) and in the SpEx+
joint mode (`joint_training=True, spk_feat=False`, enrollment waveform through the shared encoder):

  * `wesep/models/convtasnet.py:162-219`           ConvTasNet.forward (Multi encoder / decoder)
  * `wesep/modules/tasnet/encoder.py:66-114`        MultiEncoder (three Conv1d + ReLU, cLN, 1x1 proj)
  * `wesep/modules/tasnet/convs.py:107-160`         Conv1DBlock4Fuse (concatConv speaker fusion)
  * `wesep/modules/tasnet/convs.py:41-104`          Conv1DBlock (1x1 - PReLU - norm - dwconv - PReLU - norm - 1x1)
  * `wesep/modules/tasnet/separation.py:7-54,57-186` Separation / FuseSeparation (multi_fuse or not)
  * `wesep/modules/common/norm.py:7-76`             GlobalChannelLayerNorm (gLN), ChannelWiseLayerNorm (cLN)
  * `wesep/modules/tasnet/decoder.py:66-114`        MultiDecoder (3 x (1x1 mask, ReLU, multiply, ConvTranspose1d))
  * `wesep/modules/tasnet/speaker.py:7-64`         ResBlock / ResNet4SpExplus (joint training on the shared
    encoder, BatchNorm1d in training mode with running-statistics update) and the multi-task speaker head
  * the multi-scale SI-SDR objective of `spexplus.yaml:27-30` (loss_posi [0,1,2], weights .8/.1/.1),
    summed by `wesep/utils/executor.py:108-122`

Pinning: fixtures produced by the REAL reference (`oracle/make_golden.py`, cases `convtasnet_*`),
checked by `tests/test_oracle_golden.py`.  Out of this restatement (the product raises
`NotImplementedError` for them): Deep / plain encoders, `skip_con=True`, causal blocks, `norm="BN"`,
fusion types other than concatConv, joint training with a wespeaker model (`spk_feat=True`).

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this module.
"""
from dataclasses import dataclass
from typing import Dict, List

import torch
import torch.nn.functional as F

from oracle.bsrnn_oracle import sisdr_loss, spk_transform  # same third-party restatements

LN_EPS = 1e-5  # norm.py:18 (gLN) and nn.LayerNorm default (cLN)


@dataclass
class ConvTasNetConfig:
    """Constructor arguments of the reference ConvTasNet (`convtasnet.py:16-46`) that matter here."""
    N: int = 256
    L: int = 20
    B: int = 256
    H: int = 512
    P: int = 3
    X: int = 8
    R: int = 4
    spk_emb_dim: int = 256
    norm: str = "gLN"
    spk_fuse_type: str = "concatConv"   # or concat / additive / multiply / FiLM (separation.py:116-135)
    multi_fuse: bool = True
    use_spk_transform: bool = False
    joint_training: bool = False   # True: SpEx+ speaker encoder on the shared encoder (tasnet/speaker.py)
    multi_task: bool = False       # True: speaker-classification head on the embedding (convtasnet.py:116-117)
    spksInTrain: int = 251

    @property
    def stride(self):
        return self.L // 2


def _block_names(cfg: ConvTasNetConfig):
    """(kind, prefix, dilation) of every block in execution order (`separation.py:84-160,168-186`)."""
    out = []
    if cfg.multi_fuse and cfg.spk_fuse_type == "concatConv":
        for r in range(cfg.R):
            out.append(("fuse", f"separation.separation.{2 * r}.", 1))
            for j, x in enumerate(range(1, cfg.X)):
                out.append(("plain", f"separation.separation.{2 * r + 1}.separation.{j}.", 2 ** x))
    elif cfg.multi_fuse:
        # SpeakerFuseLayer - PReLU - norm - Separation(1, X) with all X dilations, four list entries per repeat
        for r in range(cfg.R):
            out.append(("fuselayer", f"separation.separation.{4 * r}.", 0))
            out.append(("prelu", f"separation.separation.{4 * r + 1}.", 0))
            out.append(("norm", f"separation.separation.{4 * r + 2}.", 0))
            for j, x in enumerate(range(0, cfg.X)):
                out.append(("plain", f"separation.separation.{4 * r + 3}.separation.{j}.", 2 ** x))
    else:
        # the reference first appends the fuse block to a ModuleList and then REPLACES the list by one
        # Separation (`separation.py:146-160`): the fuse block is dropped and forward() would call
        # Separation(x, spk_embedding).  Not a usable configuration; the product refuses it too.
        raise NotImplementedError("multi_fuse=False is broken in the reference (separation.py:146-186)")
    return out


def param_shapes(cfg: ConvTasNetConfig) -> Dict[str, tuple]:
    """Name -> shape of every parameter, in the reference's registration order."""
    N, B, H, P, E = cfg.N, cfg.B, cfg.H, cfg.P, cfg.spk_emb_dim
    s: Dict[str, tuple] = {}
    for name, L in (("short", cfg.L), ("middle", 80), ("long", 160)):
        s[f"encoder.encoder_1d_{name}.weight"] = (N, 1, L)
        s[f"encoder.encoder_1d_{name}.bias"] = (N,)
    s["encoder.ln.weight"] = (3 * N,)
    s["encoder.ln.bias"] = (3 * N,)
    s["encoder.proj.weight"] = (B, 3 * N, 1)
    s["encoder.proj.bias"] = (B,)
    if cfg.joint_training:  # ResNet4SpExplus(in_channel=N), tasnet/speaker.py:49-64 (hard-wired to N = 256)
        assert cfg.N == 256, "ResNet4SpExplus hard-codes 3*256 input channels (tasnet/speaker.py:55)"
        q = "spk_model.aux_enc3."
        s[q + "0.weight"] = (3 * N,)
        s[q + "0.bias"] = (3 * N,)
        s[q + "1.weight"] = (256, 3 * 256, 1)
        s[q + "1.bias"] = (256,)
        for i, (ci, co) in zip((2, 3, 4), ((256, 256), (256, 512), (512, 512))):
            s[q + f"{i}.conv1.weight"] = (co, ci, 1)
            s[q + f"{i}.conv2.weight"] = (co, co, 1)
            for bn in ("batch_norm1", "batch_norm2"):
                s[q + f"{i}.{bn}.weight"] = (co,)
                s[q + f"{i}.{bn}.bias"] = (co,)
                s[q + f"{i}.{bn}.running_mean"] = (co,)
                s[q + f"{i}.{bn}.running_var"] = (co,)
                s[q + f"{i}.{bn}.num_batches_tracked"] = ()
            s[q + f"{i}.prelu1.weight"] = (1,)
            s[q + f"{i}.prelu2.weight"] = (1,)
            if ci != co:
                s[q + f"{i}.conv_downsample.weight"] = (co, ci, 1)
        s[q + "5.weight"] = (E, 512, 1)
        s[q + "5.bias"] = (E,)
        if cfg.multi_task:
            s["pred_linear.weight"] = (cfg.spksInTrain, E)
            s["pred_linear.bias"] = (cfg.spksInTrain,)
    if cfg.use_spk_transform:  # speaker.py:26-43
        s["spk_transform.transforms.0.weight"] = (128, E, 1)
        s["spk_transform.transforms.0.bias"] = (128,)
        s["spk_transform.transforms.1.weight"] = (128, 128, 1)
        s["spk_transform.transforms.1.bias"] = (128,)
        s["spk_transform.transforms.3.weight"] = (E, 128, 1)
        s["spk_transform.transforms.3.bias"] = (E,)
    nshape = (H, 1) if cfg.norm == "gLN" else (H,)
    for kind, p, _ in _block_names(cfg):
        if kind == "fuselayer":                       # speaker.py:63-79 (FiLM: norm.py:84-111)
            if cfg.spk_fuse_type == "FiLM":
                s[p + "fc.gamma_fcs.0.weight"], s[p + "fc.gamma_fcs.0.bias"] = (B, E), (B,)
                s[p + "fc.beta_fcs.0.weight"], s[p + "fc.beta_fcs.0.bias"] = (B, E), (B,)
            else:
                s[p + "fc.linear.weight"] = (B, E + B) if cfg.spk_fuse_type == "concat" else (B, E)
                s[p + "fc.linear.bias"] = (B,)
        elif kind == "prelu":
            s[p + "weight"] = (1,)
        elif kind == "norm":
            s[p + "weight"] = (B, 1) if cfg.norm == "gLN" else (B,)
            s[p + "bias"] = (B, 1) if cfg.norm == "gLN" else (B,)
        elif kind == "fuse":
            s[p + "conv1x1.weight"] = (H, B + E, 1)
            s[p + "conv1x1.bias"] = (H,)
            s[p + "prelu1.weight"] = (1,)
            s[p + "lnorm1.weight"] = nshape
            s[p + "lnorm1.bias"] = nshape
            s[p + "dconv.weight"] = (H, 1, P)
            s[p + "dconv.bias"] = (H,)
            s[p + "prelu2.weight"] = (1,)
            s[p + "lnorm2.weight"] = nshape
            s[p + "lnorm2.bias"] = nshape
            s[p + "sconv.weight"] = (B, H, 1)
            s[p + "sconv.bias"] = (B,)
        else:
            s[p + "conv1x1.weight"] = (H, B, 1)
            s[p + "conv1x1.bias"] = (H,)
            s[p + "PReLU_1.weight"] = (1,)
            s[p + "norm_1.weight"] = nshape
            s[p + "norm_1.bias"] = nshape
            s[p + "dwconv.weight"] = (H, 1, P)
            s[p + "dwconv.bias"] = (H,)
            s[p + "PReLU_2.weight"] = (1,)
            s[p + "norm_2.weight"] = nshape
            s[p + "norm_2.bias"] = nshape
            s[p + "Output.weight"] = (B, H, 1)
            s[p + "Output.bias"] = (B,)
    for i in (1, 2, 3):
        s[f"decoder.mask{i}.weight"] = (N, B, 1)
        s[f"decoder.mask{i}.bias"] = (N,)
    for i, L in ((1, cfg.L), (2, 80), (3, 160)):
        s[f"decoder.decoder_1d_{i}.weight"] = (N, 1, L)
        s[f"decoder.decoder_1d_{i}.bias"] = (1,)
    return s


import re as _re

_TOP_ENTRY = _re.compile(r"^separation\.separation\.\d+\.(weight|bias)$")


def synth_params(cfg: ConvTasNetConfig, seed: int) -> Dict[str, torch.Tensor]:
    """Deterministic parameter set (fan-in scaled normal; norm gains near 1, PReLU slopes near .25)."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, shp in param_shapes(cfg).items():
        if k.endswith("running_mean"):
            v = torch.zeros(shp)
        elif k.endswith("running_var"):
            v = torch.ones(shp)
        elif k.endswith("num_batches_tracked"):
            v = torch.zeros(shp, dtype=torch.long)
        elif k.startswith("pred_linear") and k.endswith("weight"):
            v = torch.randn(shp, generator=g) / shp[1] ** 0.5
        elif _TOP_ENTRY.match(k):      # PReLU / norm entries of the non-concatConv FuseSeparation list
            if tuple(shp) == (1,):
                v = 0.25 + 0.05 * torch.randn(shp, generator=g)
            else:
                v = (1.0 + 0.1 * torch.randn(shp, generator=g)) if k.endswith("weight") else 0.1 * torch.randn(shp, generator=g)
        elif ".fc." in k and k.endswith("weight"):   # SpeakerFuseLayer / FiLM linears [B, E (+ B)]
            v = torch.randn(shp, generator=g) / shp[1] ** 0.5
        elif "norm" in k or ".ln." in k or k.startswith("spk_model.aux_enc3.0."):
            v = (1.0 + 0.1 * torch.randn(shp, generator=g)) if k.endswith("weight") else 0.1 * torch.randn(shp, generator=g)
        elif "prelu" in k.lower():
            v = 0.25 + 0.05 * torch.randn(shp, generator=g)
        elif k.endswith("bias"):
            v = 0.05 * torch.randn(shp, generator=g)
        else:
            fan_in = shp[1] * shp[2]
            if "decoder_1d" in k:
                fan_in = shp[0]
            v = torch.randn(shp, generator=g) / fan_in ** 0.5
        out[k] = v
    return out


def _norm(cfg: ConvTasNetConfig, x, w, b):
    """select_norm (`norm.py:62-76`) on [R, C, T]."""
    if cfg.norm == "gLN":  # norm.py:29-48
        mean = x.mean((1, 2), keepdim=True)
        var = ((x - mean) ** 2).mean((1, 2), keepdim=True)
        return w * (x - mean) / torch.sqrt(var + LN_EPS) + b
    if cfg.norm == "cLN":  # norm.py:51-59
        return F.layer_norm(x.transpose(1, 2), (x.shape[1],), w, b, LN_EPS).transpose(1, 2)
    raise NotImplementedError(cfg.norm)


def multi_encoder(p, cfg: ConvTasNetConfig, x):
    """`encoder.py:95-114`: x [R, T] -> (e [R, B, T'], w1, w2, w3 [R, N, T'])."""
    s = cfg.stride
    x = x.unsqueeze(1)
    w1 = F.relu(F.conv1d(x, p["encoder.encoder_1d_short.weight"], p["encoder.encoder_1d_short.bias"], stride=s))
    T1 = w1.shape[-1]
    xlen = x.shape[-1]
    l2 = (T1 - 1) * s + 80
    l3 = (T1 - 1) * s + 160
    w2 = F.relu(F.conv1d(F.pad(x, (0, l2 - xlen)), p["encoder.encoder_1d_middle.weight"],
                         p["encoder.encoder_1d_middle.bias"], stride=s))
    w3 = F.relu(F.conv1d(F.pad(x, (0, l3 - xlen)), p["encoder.encoder_1d_long.weight"],
                         p["encoder.encoder_1d_long.bias"], stride=s))
    cat = torch.cat([w1, w2, w3], 1)
    y = F.layer_norm(cat.transpose(1, 2), (cat.shape[1],), p["encoder.ln.weight"], p["encoder.ln.bias"],
                     LN_EPS).transpose(1, 2)
    e = F.conv1d(y, p["encoder.proj.weight"], p["encoder.proj.bias"])
    return e, w1, w2, w3


def conv_block(p, cfg, kind, pre, dil, x, aux):
    """Conv1DBlock4Fuse (`convs.py:143-160`) / Conv1DBlock without skip (`convs.py:84-104`)."""
    n = ("conv1x1", "prelu1", "lnorm1", "dconv", "prelu2", "lnorm2", "sconv") if kind == "fuse" else \
        ("conv1x1", "PReLU_1", "norm_1", "dwconv", "PReLU_2", "norm_2", "Output")
    y = torch.cat([x, aux.repeat(1, 1, x.shape[-1])], 1) if kind == "fuse" else x
    y = F.conv1d(y, p[pre + n[0] + ".weight"], p[pre + n[0] + ".bias"])
    y = _norm(cfg, F.prelu(y, p[pre + n[1] + ".weight"]), p[pre + n[2] + ".weight"], p[pre + n[2] + ".bias"])
    pad = dil * (cfg.P - 1) // 2
    y = F.conv1d(y, p[pre + n[3] + ".weight"], p[pre + n[3] + ".bias"], padding=pad, dilation=dil, groups=cfg.H)
    y = _norm(cfg, F.prelu(y, p[pre + n[4] + ".weight"]), p[pre + n[5] + ".weight"], p[pre + n[5] + ".bias"])
    y = F.conv1d(y, p[pre + n[6] + ".weight"], p[pre + n[6] + ".bias"])
    return x + y


def multi_decoder(p, cfg, e, ws) -> List[torch.Tensor]:
    """`decoder.py:92-114` with actLayer = ReLU (`convtasnet.py:152-160`, activate="relu")."""
    outs = []
    for i, w in zip((1, 2, 3), ws):
        m = F.relu(F.conv1d(e, p[f"decoder.mask{i}.weight"], p[f"decoder.mask{i}.bias"]))
        est = F.conv_transpose1d(w * m, p[f"decoder.decoder_1d_{i}.weight"], p[f"decoder.decoder_1d_{i}.bias"],
                                 stride=cfg.stride).squeeze(1)
        outs.append(est)
    xlen = outs[0].shape[-1]
    return [outs[0], outs[1][:, :xlen], outs[2][:, :xlen]]


BN_EPS, BN_MOMENTUM = 1e-5, 0.1  # nn.BatchNorm1d defaults (tasnet/speaker.py:19-20)


def is_buffer(name: str) -> bool:
    return name.endswith(("running_mean", "running_var", "num_batches_tracked"))


def _res_block(p, q, x, training, new_buffers):
    """ResBlock (`tasnet/speaker.py:31-44`): conv1x1 - BN - PReLU - conv1x1 - BN - (+ residual) - PReLU - MaxPool1d(3)."""
    def bn(name, y):
        rm, rv = p[q + name + ".running_mean"].clone(), p[q + name + ".running_var"].clone()
        out = F.batch_norm(y, rm, rv, p[q + name + ".weight"], p[q + name + ".bias"], training, BN_MOMENTUM, BN_EPS)
        if new_buffers is not None and training:
            new_buffers[q + name + ".running_mean"], new_buffers[q + name + ".running_var"] = rm, rv
        return out
    res = x
    y = F.prelu(bn("batch_norm1", F.conv1d(x, p[q + "conv1.weight"])), p[q + "prelu1.weight"])
    y = bn("batch_norm2", F.conv1d(y, p[q + "conv2.weight"]))
    if (q + "conv_downsample.weight") in p:
        res = F.conv1d(res, p[q + "conv_downsample.weight"])
    return F.max_pool1d(F.prelu(y + res, p[q + "prelu2.weight"]), 3)


def spk_encoder(p, cfg, aux_cat, training=True, new_buffers=None):
    """ResNet4SpExplus (`tasnet/speaker.py:49-64`) on the shared encoder's [w1 | w2 | w3] of the enrollment
    (`convtasnet.py:179-187`): [R, 3N, Te'] -> [R, E]."""
    q = "spk_model.aux_enc3."
    y = F.layer_norm(aux_cat.transpose(1, 2), (aux_cat.shape[1],), p[q + "0.weight"], p[q + "0.bias"],
                     LN_EPS).transpose(1, 2)
    y = F.conv1d(y, p[q + "1.weight"], p[q + "1.bias"])
    for i in (2, 3, 4):
        y = _res_block(p, q + f"{i}.", y, training, new_buffers)
    y = F.conv1d(y, p[q + "5.weight"], p[q + "5.bias"])
    return y.mean(-1)


def convtasnet_forward(p: Dict[str, torch.Tensor], cfg: ConvTasNetConfig, wav: torch.Tensor,
                       emb: torch.Tensor, training=True, new_buffers=None) -> List[torch.Tensor]:
    """`convtasnet.py:162-219`: wav [R, T]; emb [R, E] (fixed embeddings) or, with joint_training, the
    enrollment waveform [R, Tw] -> [est1, est2, est3] (+ [speaker logits] with multi_task)."""
    e, w1, w2, w3 = multi_encoder(p, cfg, wav)
    logits = None
    if cfg.joint_training:
        _, a1, a2, a3 = multi_encoder(p, cfg, emb)
        emb = spk_encoder(p, cfg, torch.cat([a1, a2, a3], 1), training, new_buffers)
        if cfg.multi_task:
            logits = F.linear(emb, p["pred_linear.weight"], p["pred_linear.bias"])
    if cfg.use_spk_transform:  # Conv1d(k=1) chain: the [R, E] form is the same map (speaker.py:45-49)
        emb = spk_transform(p, emb)
    aux = emb.unsqueeze(-1)
    for kind, pre, dil in _block_names(cfg):
        if kind == "fuselayer":                       # speaker.py:81-125 on x [R, B, T'], embed [R, E, 1]
            ft = cfg.spk_fuse_type
            if ft == "FiLM":
                gm = F.linear(emb, p[pre + "fc.gamma_fcs.0.weight"], p[pre + "fc.gamma_fcs.0.bias"]).unsqueeze(-1)
                bt = F.linear(emb, p[pre + "fc.beta_fcs.0.weight"], p[pre + "fc.beta_fcs.0.bias"]).unsqueeze(-1)
                e = (1 + gm) * e + bt
            elif ft == "concat":
                y = torch.cat([e, aux.expand(-1, -1, e.shape[-1])], 1).transpose(1, 2)
                e = F.linear(y, p[pre + "fc.linear.weight"], p[pre + "fc.linear.bias"]).transpose(1, 2)
            else:
                t = F.linear(emb, p[pre + "fc.linear.weight"], p[pre + "fc.linear.bias"]).unsqueeze(-1)
                e = e + t if ft == "additive" else e * t
        elif kind == "prelu":
            e = F.prelu(e, p[pre + "weight"])
        elif kind == "norm":
            e = _norm(cfg, e, p[pre + "weight"], p[pre + "bias"])
        else:
            e = conv_block(p, cfg, kind, pre, dil, e, aux)
    outs = multi_decoder(p, cfg, e, (w1, w2, w3))
    if logits is not None:
        outs.append(logits)
    return outs


def spexplus_loss(outs: List[torch.Tensor], target: torch.Tensor, spk_label: torch.Tensor):
    """`spexplus.yaml:27-30` through `executor.py:108-122`: .8/.1/.1 SI-SDR + 0.5 CrossEntropy(logits, label)."""
    return multiscale_sisdr_loss(outs[:3], target) + 0.5 * F.cross_entropy(outs[3], spk_label)


def multiscale_sisdr_loss(ests: List[torch.Tensor], target: torch.Tensor, weights=(0.8, 0.1, 0.1)):
    """`executor.py:108-122` with `spexplus.yaml:27-30`: sum_i w_i * SISDR(est_i, target[:, :len])."""
    loss = 0.0
    for w, est in zip(weights, ests):
        n = min(est.shape[-1], target.shape[-1])
        loss = loss + w * sisdr_loss(est[:, :n], target[:, :n])
    return loss


def synth_enrollment(R: int, Tw: int, nspk: int, seed: int):
    """Enrollment waveforms [R, Tw] and speaker labels [R] for the joint mode (deterministic)."""
    g = torch.Generator().manual_seed(seed + 1000)
    return 0.1 * torch.randn(R, Tw, generator=g), torch.randint(0, nspk, (R,), generator=g)
