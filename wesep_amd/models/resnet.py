"""wespeaker ResNet speaker encoder on MI355X: module tree and `state_dict` keys of
`wespeaker.models.resnet.ResNet` (BasicBlock variants: ResNet18 / ResNet34), so `spk_model_init` checkpoints load.
The package itself is a third-party dependency absent from the reference tree: parity is against the restatement in
oracle/resnet_oracle.py and is UNPINNED.  nn.Conv2d / nn.BatchNorm2d / nn.Linear objects are parameter
containers only; forward is a chain of C-ABI launches (wesep_amd/functional_resnet.py)."""
import torch
import torch.nn as nn

from .. import functional_resnet as FR
from ..functional import LinearFn


class TSTP(nn.Module):
    """Temporal statistics pooling (no parameters): mean || std."""

    def __init__(self, in_dim=0, **kwargs):
        super().__init__()
        self.in_dim = in_dim

    def get_out_dim(self):
        return self.in_dim * 2


class TAP(TSTP):
    """Temporal average pooling: the mean half of TSTP."""

    def get_out_dim(self):
        return self.in_dim


class TSDP(TSTP):
    """Temporal standard-deviation pooling: the std half of TSTP."""

    def get_out_dim(self):
        return self.in_dim


def _pooling_layer(name, in_dim):
    """wespeaker.models.pooling_layers by name: TSTP / TAP / TSDP (one statistics kernel) and ASTP (attentive statistics,
    the ECAPA-TDNN module of models/ecapa_tdnn.py on the [R, C * F', T] view)."""
    if name in ("TSTP", "TAP", "TSDP"):
        return {"TSTP": TSTP, "TAP": TAP, "TSDP": TSDP}[name](in_dim=in_dim)
    if name == "ASTP":
        from .ecapa_tdnn import ASTP
        return ASTP(in_dim=in_dim)
    raise NotImplementedError(f"pooling_func {name!r}: TSTP, TAP, TSDP and ASTP are built (not MHASTP / MQMHASTP)")


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, in_planes, planes, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(in_planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=1, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.shortcut = nn.Sequential()
        if stride != 1 or in_planes != self.expansion * planes:
            self.shortcut = nn.Sequential(
                nn.Conv2d(in_planes, self.expansion * planes, kernel_size=1, stride=stride, bias=False),
                nn.BatchNorm2d(self.expansion * planes))
        self.stride = stride


class Bottleneck(nn.Module):
    """wespeaker Bottleneck (ResNet50 / 101 / 152): 1x1 - 3x3(stride) - 1x1, expansion 4."""
    expansion = 4

    def __init__(self, in_planes, planes, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(in_planes, planes, kernel_size=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, self.expansion * planes, kernel_size=1, bias=False)
        self.bn3 = nn.BatchNorm2d(self.expansion * planes)
        self.shortcut = nn.Sequential()
        if stride != 1 or in_planes != self.expansion * planes:
            self.shortcut = nn.Sequential(
                nn.Conv2d(in_planes, self.expansion * planes, kernel_size=1, stride=stride, bias=False),
                nn.BatchNorm2d(self.expansion * planes))
        self.stride = stride


def _cba(x, res, R, H, W, stride, relu, conv, bn, training):
    if training:
        bn.num_batches_tracked += 1
    return FR.ConvBnActFn.apply(x, res, (R, H, W, stride, relu, training), conv.weight, bn.weight, bn.bias,
                                bn.running_mean, bn.running_var)


class ResNet(nn.Module):
    def __init__(self, block, num_blocks, m_channels=32, feat_dim=40, embed_dim=128, pooling_func="TSTP",
                 two_emb_layer=True):
        super().__init__()
        self.pooling_func = pooling_func
        self.in_planes, self.feat_dim, self.embed_dim = m_channels, feat_dim, embed_dim
        self.stats_dim = int(feat_dim / 8) * m_channels * 8
        self.two_emb_layer = two_emb_layer
        self.conv1 = nn.Conv2d(1, m_channels, kernel_size=3, stride=1, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(m_channels)
        self.layer1 = self._make_layer(block, m_channels, num_blocks[0], stride=1)
        self.layer2 = self._make_layer(block, m_channels * 2, num_blocks[1], stride=2)
        self.layer3 = self._make_layer(block, m_channels * 4, num_blocks[2], stride=2)
        self.layer4 = self._make_layer(block, m_channels * 8, num_blocks[3], stride=2)
        self.pool = _pooling_layer(pooling_func, self.stats_dim * block.expansion)
        self.pool_out_dim = self.pool.get_out_dim()
        self.seg_1 = nn.Linear(self.pool_out_dim, embed_dim)
        if two_emb_layer:
            self.seg_bn_1 = nn.BatchNorm1d(embed_dim, affine=False)
            self.seg_2 = nn.Linear(embed_dim, embed_dim)
        else:
            self.seg_bn_1 = nn.Identity()
            self.seg_2 = nn.Identity()

    def _make_layer(self, block, planes, num_blocks, stride):
        layers = []
        for s in [stride] + [1] * (num_blocks - 1):
            layers.append(block(self.in_planes, planes, s))
            self.in_planes = planes * block.expansion
        return nn.Sequential(*layers)

    def forward(self, x):
        """x [R, T, F] fbank -> (tensor(0.), embed_a [R, embed_dim]), or (embed_a, embed_b) with two_emb_layer."""
        if not x.is_cuda:
            from .._lib import WesepHipError
            raise WesepHipError("ResNet speaker encoder: wesep_amd has no CPU path")
        R, T, Fq = x.shape
        tr = self.training
        y = x.float().transpose(1, 2).contiguous().view(R * Fq * T, 1)      # [R, F, T, 1]
        H, W = Fq, T
        y = _cba(y, None, R, H, W, 1, True, self.conv1, self.bn1, tr)
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            for blk in layer:
                s = blk.stride
                Ho, Wo = (H + 2 - 3) // s + 1, (W + 2 - 3) // s + 1
                sc = y
                if len(blk.shortcut) > 0:
                    sc = _cba(y, None, R, H, W, s, False, blk.shortcut[0], blk.shortcut[1], tr)
                if isinstance(blk, Bottleneck):
                    o = _cba(y, None, R, H, W, 1, True, blk.conv1, blk.bn1, tr)
                    o = _cba(o, None, R, H, W, s, True, blk.conv2, blk.bn2, tr)
                    y = _cba(o, sc, R, Ho, Wo, 1, True, blk.conv3, blk.bn3, tr)
                else:
                    o = _cba(y, None, R, H, W, s, True, blk.conv1, blk.bn1, tr)
                    y = _cba(o, sc, R, Ho, Wo, 1, True, blk.conv2, blk.bn2, tr)
                H, W = Ho, Wo
        if self.pooling_func == "ASTP":      # [R, F', T, C] -> frames [R*T, C * F'] (feature index c * F' + f), then ASTP
            Cc = y.shape[1]
            frames = y.view(R, H, W, Cc).permute(0, 2, 3, 1).reshape(R * W, Cc * H)
            stats = self.pool.run(frames, R, W)
        else:
            stats = FR.TstpFn.apply(y, (R, H, W))                               # mean || std, each [C * F']
            half = stats.shape[1] // 2
            if self.pooling_func == "TAP":
                stats = stats[:, :half].contiguous()
            elif self.pooling_func == "TSDP":
                stats = stats[:, half:].contiguous()
        embed_a = LinearFn.apply(stats, self.seg_1.weight, self.seg_1.bias)
        if not self.two_emb_layer:
            return torch.tensor(0.0), embed_a
        from .. import functional_ecapa as FE
        E = self.embed_dim
        if tr:
            self.seg_bn_1.num_batches_tracked += 1
        ones, zeros = torch.ones(E, device=x.device), torch.zeros(E, device=x.device)
        o = FE.BatchNormRowsFn.apply(torch.relu(embed_a), ones, zeros, self.seg_bn_1.running_mean,
                                     self.seg_bn_1.running_var, tr)          # affine=False; [R, E]: a few thousand numbers
        return embed_a, LinearFn.apply(o, self.seg_2.weight, self.seg_2.bias)


def ResNet18(feat_dim, embed_dim, pooling_func="TSTP", two_emb_layer=True):
    return ResNet(BasicBlock, [2, 2, 2, 2], feat_dim=feat_dim, embed_dim=embed_dim, pooling_func=pooling_func,
                  two_emb_layer=two_emb_layer)


def ResNet34(feat_dim, embed_dim, pooling_func="TSTP", two_emb_layer=True):
    return ResNet(BasicBlock, [3, 4, 6, 3], feat_dim=feat_dim, embed_dim=embed_dim, pooling_func=pooling_func,
                  two_emb_layer=two_emb_layer)


def _bottleneck_resnet(num_blocks):
    def make(feat_dim, embed_dim, pooling_func="TSTP", two_emb_layer=True):
        return ResNet(Bottleneck, num_blocks, feat_dim=feat_dim, embed_dim=embed_dim, pooling_func=pooling_func,
                      two_emb_layer=two_emb_layer)
    return make


ResNet50, ResNet101, ResNet152 = (_bottleneck_resnet(n) for n in ([3, 4, 6, 3], [3, 4, 23, 3], [3, 8, 36, 3]))


def get_speaker_model(model_name: str):
    """`wespeaker.models.speaker_model.get_speaker_model` for the encoders built here."""
    table = {"ResNet18": ResNet18, "ResNet34": ResNet34, "ResNet50": ResNet50, "ResNet101": ResNet101,
             "ResNet152": ResNet152}
    if model_name in table:
        return table[model_name]
    from .ecapa_tdnn import ECAPA_MODELS
    if model_name in ECAPA_MODELS:
        return ECAPA_MODELS[model_name]
    if model_name == "CAMPPlus":
        from .campplus import CAMPPlus
        return CAMPPlus
    raise NotImplementedError(f"speaker model {model_name!r}: the wespeaker ResNets (18 / 34 / 50 / 101 / 152), "
                              "ECAPA-TDNN (c512 / c1024, with or without global context) and CAM++ (CAMPPlus) are built "
                              "(SURVEY.md section 8 row a12)")
