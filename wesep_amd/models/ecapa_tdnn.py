"""wespeaker ECAPA-TDNN speaker encoder on MI355X: module tree and `state_dict` keys
of `wespeaker.models.ecapa_tdnn.ECAPA_TDNN` (ECAPA_TDNN_c512 / _c1024 and their global-context "GLOB" variants with
ASTP pooling), so that `spk_model_init` checkpoints and the speaker half of the reference's published
`bsrnn_ecapa_vox1` model (wesep/cli/hub.py:86-95; recipe alternative examples/librimix/tse/v2/confs/bsrnn.yaml:66-71)
load by name.  wespeaker is a third-party dependency absent from the reference tree: the architecture is restated
from its published definition (Desplanques et al. 2020: Conv1d-ReLU-BN layers, three SE-Res2Blocks of scale 8 with
dilations 2 / 3 / 4, multi-layer aggregation, attentive statistics pooling with global context, BN, Linear) and parity
is against oracle/ecapa_oracle.py -- UNPINNED, like the ResNet.  nn.Conv1d / nn.BatchNorm1d / nn.Linear
objects are parameter containers only; forward is a chain of C-ABI launches (wesep_amd/functional_ecapa.py)."""
import torch
import torch.nn as nn

from .. import functional_ecapa as FE
from ..functional import LinearFn


def _crb(x, R, T, m, training):
    """Conv1dReluBn module -> Conv -> ReLU -> BN launches."""
    if training:
        m.bn.num_batches_tracked += 1
    return FE.Conv1dReluBnFn.apply(x, (R, T, m.conv.dilation[0], training), m.conv.weight, m.conv.bias, m.bn.weight,
                                   m.bn.bias, m.bn.running_mean, m.bn.running_var)


class Conv1dReluBn(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=1, stride=1, padding=0, dilation=1, bias=True):
        super().__init__()
        if stride != 1 or padding != dilation * (kernel_size // 2) or not bias:
            raise NotImplementedError("ECAPA-TDNN Conv1dReluBn: stride 1, 'same' padding and a bias are built")
        self.conv = nn.Conv1d(in_channels, out_channels, kernel_size, stride, padding, dilation, bias=bias)
        self.bn = nn.BatchNorm1d(out_channels)

    def run(self, x, R, T):
        return _crb(x, R, T, self, self.training)


class Res2Conv1dReluBn(nn.Module):
    """Res2Net branch: the channels split in `scale` groups; group i >= 1 adds the previous group's output before its own
    Conv -> ReLU -> BN; the last group passes through."""

    def __init__(self, channels, kernel_size=1, stride=1, padding=0, dilation=1, bias=True, scale=4):
        super().__init__()
        assert channels % scale == 0
        self.scale, self.width = scale, channels // scale
        self.nums = scale if scale == 1 else scale - 1
        self.dilation = dilation
        self.convs = nn.ModuleList([nn.Conv1d(self.width, self.width, kernel_size, stride, padding, dilation, bias=bias)
                                    for _ in range(self.nums)])
        self.bns = nn.ModuleList([nn.BatchNorm1d(self.width) for _ in range(self.nums)])

    def run(self, x, R, T):
        spx = torch.split(x, self.width, 1)
        out, sp = [], None
        for i, (conv, bn) in enumerate(zip(self.convs, self.bns)):
            sp = spx[i].contiguous() if i == 0 else sp + spx[i]
            if self.training:
                bn.num_batches_tracked += 1
            sp = FE.Conv1dReluBnFn.apply(sp, (R, T, conv.dilation[0], self.training), conv.weight, conv.bias, bn.weight,
                                         bn.bias, bn.running_mean, bn.running_var)
            out.append(sp)
        if self.scale != 1:
            out.append(spx[self.nums])
        return torch.cat(out, 1)


class SE_Connect(nn.Module):
    def __init__(self, channels, se_bottleneck_dim=128):
        super().__init__()
        self.linear1 = nn.Linear(channels, se_bottleneck_dim)
        self.linear2 = nn.Linear(se_bottleneck_dim, channels)

    def run(self, x, R, T):
        m = FE.TimeMeanFn.apply(x, (R, T))                                             # [R, C]
        h = torch.relu(LinearFn.apply(m, self.linear1.weight, self.linear1.bias))      # [R, 128]: a few thousand numbers
        s = FE.RowBiasActFn.apply(LinearFn.apply(h, self.linear2.weight, self.linear2.bias), None, 1, 3)
        return FE.GateFn.apply(x, s, (R, T))                                           # x * gate[r, c]


class SE_Res2Block(nn.Module):
    def __init__(self, channels, kernel_size, stride, padding, dilation, scale):
        super().__init__()
        self.se_res2block = nn.Sequential(
            Conv1dReluBn(channels, channels, kernel_size=1, stride=1, padding=0),
            Res2Conv1dReluBn(channels, kernel_size, stride, padding, dilation, scale=scale),
            Conv1dReluBn(channels, channels, kernel_size=1, stride=1, padding=0),
            SE_Connect(channels))

    def run(self, x, R, T):
        y = x
        for m in self.se_res2block:
            y = m.run(y, R, T)
        return x + y


class ASTP(nn.Module):
    """Attentive statistics pooling; global_context_att: the attention also sees the utterance mean / std."""

    def __init__(self, in_dim, bottleneck_dim=128, global_context_att=False, **kwargs):
        super().__init__()
        self.in_dim, self.global_context_att = in_dim, global_context_att
        self.linear1 = nn.Conv1d(in_dim * 3 if global_context_att else in_dim, bottleneck_dim, kernel_size=1)
        self.linear2 = nn.Conv1d(bottleneck_dim, in_dim, kernel_size=1)

    def get_out_dim(self):
        self.out_dim = 2 * self.in_dim
        return self.out_dim

    def run(self, x, R, T):
        from .. import functional_resnet as FR
        C = self.in_dim
        w1 = self.linear1.weight.view(self.linear1.weight.shape[0], -1)
        rb = None
        if self.global_context_att:
            # cat(x, mean.expand, std.expand) W1^T = x Wx^T + (mean Wm^T + std Ws^T): the context is a per-row bias
            ctxt = FR.TstpFn.apply(x, (R, 1, T))                                         # [R, 2C] mean || sqrt(var + 1e-7)
            rb = LinearFn.apply(ctxt, w1[:, C:].contiguous(), self.linear1.bias)        # [R, 128]
            a = LinearFn.apply(x, w1[:, :C].contiguous(), torch.zeros_like(self.linear1.bias))
        else:
            a = LinearFn.apply(x, w1, self.linear1.bias)
        a = FE.RowBiasActFn.apply(a, rb, T, 1)                                            # tanh
        logits = LinearFn.apply(a, self.linear2.weight.view(C, -1), self.linear2.bias)   # [R*T, C]
        return FE.AstpFn.apply(x, logits, (R, T))


class ECAPA_TDNN(nn.Module):
    def __init__(self, channels=512, feat_dim=80, embed_dim=192, pooling_func="ASTP", global_context_att=False,
                 emb_bn=False):
        super().__init__()
        if pooling_func != "ASTP":
            raise NotImplementedError(f"ECAPA-TDNN pooling_func {pooling_func!r}: ASTP (the recipe's) is built")
        self.layer1 = Conv1dReluBn(feat_dim, channels, kernel_size=5, padding=2)
        self.layer2 = SE_Res2Block(channels, kernel_size=3, stride=1, padding=2, dilation=2, scale=8)
        self.layer3 = SE_Res2Block(channels, kernel_size=3, stride=1, padding=3, dilation=3, scale=8)
        self.layer4 = SE_Res2Block(channels, kernel_size=3, stride=1, padding=4, dilation=4, scale=8)
        cat_channels = channels * 3
        out_channels = 512 * 3
        self.conv = nn.Conv1d(cat_channels, out_channels, kernel_size=1)
        self.pool = ASTP(in_dim=out_channels, global_context_att=global_context_att)
        self.pool_out_dim = self.pool.get_out_dim()
        self.bn = nn.BatchNorm1d(self.pool_out_dim)
        self.linear = nn.Linear(self.pool_out_dim, embed_dim)
        self.emb_bn = emb_bn
        self.bn2 = nn.BatchNorm1d(embed_dim) if emb_bn else nn.Identity()
        self.feat_dim, self.embed_dim = feat_dim, embed_dim

    def forward(self, x):
        """x [R, T, F] fbank -> embedding [R, embed_dim]."""
        if not x.is_cuda:
            from .._lib import WesepHipError
            raise WesepHipError("ECAPA-TDNN speaker encoder: wesep_amd has no CPU path")
        R, T, Fq = x.shape
        if Fq % 4:
            raise NotImplementedError("ECAPA-TDNN: feat_dim must be a multiple of 4")
        tr = self.training
        y = x.float().contiguous().view(R * T, Fq)                                    # already channels-last
        out1 = self.layer1.run(y, R, T)
        out2 = self.layer2.run(out1, R, T)
        out3 = self.layer3.run(out2, R, T)
        out4 = self.layer4.run(out3, R, T)
        cat = torch.cat([out2, out3, out4], 1)
        h = FE.LinearReluFn.apply(cat, self.conv.weight.view(self.conv.weight.shape[0], -1), self.conv.bias)
        stats = self.pool.run(h, R, T)
        if tr:
            self.bn.num_batches_tracked += 1
        stats = FE.BatchNormRowsFn.apply(stats, self.bn.weight, self.bn.bias, self.bn.running_mean, self.bn.running_var, tr)
        emb = LinearFn.apply(stats, self.linear.weight, self.linear.bias)
        if self.emb_bn:
            if tr:
                self.bn2.num_batches_tracked += 1
            emb = FE.BatchNormRowsFn.apply(emb, self.bn2.weight, self.bn2.bias, self.bn2.running_mean,
                                           self.bn2.running_var, tr)
        return emb


def ECAPA_TDNN_c512(feat_dim, embed_dim, pooling_func="ASTP", emb_bn=False):
    return ECAPA_TDNN(channels=512, feat_dim=feat_dim, embed_dim=embed_dim, pooling_func=pooling_func, emb_bn=emb_bn)


def ECAPA_TDNN_c1024(feat_dim, embed_dim, pooling_func="ASTP", emb_bn=False):
    return ECAPA_TDNN(channels=1024, feat_dim=feat_dim, embed_dim=embed_dim, pooling_func=pooling_func, emb_bn=emb_bn)


def ECAPA_TDNN_GLOB_c512(feat_dim, embed_dim, pooling_func="ASTP", emb_bn=False):
    return ECAPA_TDNN(channels=512, feat_dim=feat_dim, embed_dim=embed_dim, pooling_func=pooling_func,
                      global_context_att=True, emb_bn=emb_bn)


def ECAPA_TDNN_GLOB_c1024(feat_dim, embed_dim, pooling_func="ASTP", emb_bn=False):
    return ECAPA_TDNN(channels=1024, feat_dim=feat_dim, embed_dim=embed_dim, pooling_func=pooling_func,
                      global_context_att=True, emb_bn=emb_bn)


ECAPA_MODELS = {f.__name__: f for f in (ECAPA_TDNN_c512, ECAPA_TDNN_c1024, ECAPA_TDNN_GLOB_c512, ECAPA_TDNN_GLOB_c1024)}
