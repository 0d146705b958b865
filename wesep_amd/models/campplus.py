"""wespeaker CAM++ speaker encoder (`CAMPPlus`) on MI355X: module tree and `state_dict` keys of
`wespeaker.models.campplus.CAMPPlus` (FCM convolutional head, D-TDNN backbone of three CAM-dense-TDNN blocks with
context-aware masking, TSTP pooling, dense embedding layer), so that `spk_model_init` checkpoints load by name and
`BSRNN(spk_model="CAMPPlus", ...)` trains jointly (recipe alternative: examples/librimix/tse/v2/confs/bsrnn.yaml:66-74;
call site wesep/models/bsrnn.py:217,352-356).  wespeaker is a third-party dependency absent from the reference tree: the
architecture is restated from its published definition (Wang et al. 2023) and parity is against
oracle/campplus_oracle.py -- UNPINNED, like the ResNet and ECAPA-TDNN; the restatement's parameter count for
80 mel bins / 512-d embeddings is the published 7.18 M.  nn.Conv / nn.BatchNorm objects are parameter containers only;
forward is a chain of C-ABI launches (functional_campplus.py, functional_resnet.py)."""
from collections import OrderedDict

import torch
import torch.nn as nn

from .. import functional_campplus as FC
from .. import functional_resnet as FR

SEG_LEN = 100


def _bn_act(x, bn, training, relu=True):
    if training:
        bn.num_batches_tracked += 1
    return FC.BnActFn.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, training, relu)


def _cba(x, res, R, H, W, stride, relu, conv, bn, training):
    if training:
        bn.num_batches_tracked += 1
    return FR.ConvBnActFn.apply(x, res, (R, H, W, stride, relu, training), conv.weight, bn.weight, bn.bias,
                                bn.running_mean, bn.running_var)


def get_nonlinear(config_str, channels):
    nonlinear = nn.Sequential()
    for name in config_str.split("-"):
        if name == "relu":
            nonlinear.add_module("relu", nn.ReLU(inplace=True))
        elif name == "batchnorm":
            nonlinear.add_module("batchnorm", nn.BatchNorm1d(channels))
        elif name == "batchnorm_":
            nonlinear.add_module("batchnorm", nn.BatchNorm1d(channels, affine=False))
        else:
            raise NotImplementedError(f"CAM++ nonlinearity {name!r}: 'batchnorm-relu' (the default) and 'batchnorm_' are built")
    return nonlinear


def _nonlinear(x, seq, training):
    return _bn_act(x, seq.batchnorm, training, relu=hasattr(seq, "relu"))


class BasicResBlock(nn.Module):
    """FCM residual block: the stride acts on the mel axis only."""
    expansion = 1

    def __init__(self, in_planes, planes, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(in_planes, planes, kernel_size=3, stride=(stride, 1), padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=1, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.shortcut = nn.Sequential()
        if stride != 1 or in_planes != self.expansion * planes:
            self.shortcut = nn.Sequential(
                nn.Conv2d(in_planes, self.expansion * planes, kernel_size=1, stride=(stride, 1), bias=False),
                nn.BatchNorm2d(self.expansion * planes))
        self.stride = stride


class FCM(nn.Module):
    def __init__(self, block=BasicResBlock, num_blocks=(2, 2), m_channels=32, feat_dim=80):
        super().__init__()
        self.in_planes = m_channels
        self.conv1 = nn.Conv2d(1, m_channels, kernel_size=3, stride=1, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(m_channels)
        self.layer1 = self._make_layer(block, m_channels, num_blocks[0], stride=2)
        self.layer2 = self._make_layer(block, m_channels, num_blocks[1], stride=2)
        self.conv2 = nn.Conv2d(m_channels, m_channels, kernel_size=3, stride=(2, 1), padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(m_channels)
        self.out_channels = m_channels * (feat_dim // 8)

    def _make_layer(self, block, planes, num_blocks, stride):
        layers = []
        for s in [stride] + [1] * (num_blocks - 1):
            layers.append(block(self.in_planes, planes, s))
            self.in_planes = planes * block.expansion
        return nn.Sequential(*layers)

    def run(self, y, R, H, W, tr):
        """y [R*H*W, 1] (H mel bins, W frames) -> [R*W, C * H/8] channels-last frames, channel index c * H/8 + h."""
        y = _cba(y, None, R, H, W, 1, True, self.conv1, self.bn1, tr)
        for layer in (self.layer1, self.layer2):
            for blk in layer:
                s = (blk.stride, 1)
                Ho = (H + 2 - 3) // blk.stride + 1
                sc = y
                if len(blk.shortcut) > 0:
                    sc = _cba(y, None, R, H, W, s, False, blk.shortcut[0], blk.shortcut[1], tr)
                o = _cba(y, None, R, H, W, s, True, blk.conv1, blk.bn1, tr)
                y = _cba(o, sc, R, Ho, W, 1, True, blk.conv2, blk.bn2, tr)
                H = Ho
        Ho = (H + 2 - 3) // 2 + 1
        y = _cba(y, None, R, H, W, (2, 1), True, self.conv2, self.bn2, tr)
        Cc = y.shape[1]
        # [R, H', W, C] -> [R, W, C, H']: the reference's reshape of [B, C, H', W] to [B, C * H', W], channels-last
        return y.view(R, Ho, W, Cc).permute(0, 2, 3, 1).reshape(R * W, Cc * Ho)


class TDNNLayer(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, bias=False,
                 config_str="batchnorm-relu"):
        super().__init__()
        if padding < 0:
            assert kernel_size % 2 == 1
            padding = (kernel_size - 1) // 2 * dilation
        if padding != (kernel_size - 1) // 2 * dilation:
            raise NotImplementedError("CAM++ TDNNLayer: 'same' padding (padding = -1) is built")
        self.linear = nn.Conv1d(in_channels, out_channels, kernel_size, stride=stride, padding=padding,
                                dilation=dilation, bias=bias)
        self.nonlinear = get_nonlinear(config_str, out_channels)

    def run(self, x, R, T, tr):
        c = self.linear
        y = FC.Conv1dFn.apply(x, (R, T, c.stride[0], c.dilation[0]), c.weight, c.bias)
        To = y.shape[0] // R
        return _nonlinear(y, self.nonlinear, tr), To


class CAMLayer(nn.Module):
    def __init__(self, bn_channels, out_channels, kernel_size, stride, padding, dilation, bias, reduction=2):
        super().__init__()
        if stride != 1 or padding != (kernel_size - 1) // 2 * dilation:
            raise NotImplementedError("CAM++ CAMLayer: stride 1 and 'same' padding are built")
        self.linear_local = nn.Conv1d(bn_channels, out_channels, kernel_size, stride=stride, padding=padding,
                                      dilation=dilation, bias=bias)
        self.linear1 = nn.Conv1d(bn_channels, bn_channels // reduction, 1)
        self.relu = nn.ReLU(inplace=True)
        self.linear2 = nn.Conv1d(bn_channels // reduction, out_channels, 1)
        self.sigmoid = nn.Sigmoid()

    def run(self, x, R, T):
        from ..functional import LinearFn
        from ..functional_ecapa import RowBiasActFn
        c = self.linear_local
        y = FC.Conv1dFn.apply(x, (R, T, 1, c.dilation[0]), c.weight, c.bias)
        ctx = FC.SegContextFn.apply(x, (R, T, SEG_LEN))                       # [R * nseg, bn]: one row per segment
        h = torch.relu(LinearFn.apply(ctx, self.linear1.weight.view(self.linear1.weight.shape[0], -1), self.linear1.bias))
        m = RowBiasActFn.apply(LinearFn.apply(h, self.linear2.weight.view(self.linear2.weight.shape[0], -1),
                                              self.linear2.bias), None, 1, 3)    # sigmoid
        return FC.SegGateFn.apply(y, m, (R, T, SEG_LEN))


class CAMDenseTDNNLayer(nn.Module):
    def __init__(self, in_channels, out_channels, bn_channels, kernel_size, stride=1, dilation=1, bias=False,
                 config_str="batchnorm-relu", memory_efficient=False):
        super().__init__()
        assert kernel_size % 2 == 1
        padding = (kernel_size - 1) // 2 * dilation
        self.memory_efficient = memory_efficient          # upstream: activation checkpointing; no effect on the result
        self.nonlinear1 = get_nonlinear(config_str, in_channels)
        self.linear1 = nn.Conv1d(in_channels, bn_channels, 1, bias=False)
        self.nonlinear2 = get_nonlinear(config_str, bn_channels)
        self.cam_layer = CAMLayer(bn_channels, out_channels, kernel_size, stride=stride, padding=padding,
                                  dilation=dilation, bias=bias)

    def run(self, x, R, T, tr):
        h = FC.Conv1dFn.apply(_nonlinear(x, self.nonlinear1, tr), (R, T, 1, 1), self.linear1.weight, None)
        return self.cam_layer.run(_nonlinear(h, self.nonlinear2, tr), R, T)


class CAMDenseTDNNBlock(nn.ModuleList):
    def __init__(self, num_layers, in_channels, out_channels, bn_channels, kernel_size, stride=1, dilation=1, bias=False,
                 config_str="batchnorm-relu", memory_efficient=False):
        super().__init__()
        for i in range(num_layers):
            self.add_module("tdnnd%d" % (i + 1),
                            CAMDenseTDNNLayer(in_channels=in_channels + i * out_channels, out_channels=out_channels,
                                              bn_channels=bn_channels, kernel_size=kernel_size, stride=stride,
                                              dilation=dilation, bias=bias, config_str=config_str,
                                              memory_efficient=memory_efficient))

    def run(self, x, R, T, tr):
        for layer in self:
            x = torch.cat([x, layer.run(x, R, T, tr)], dim=1)
        return x


class TransitLayer(nn.Module):
    def __init__(self, in_channels, out_channels, bias=True, config_str="batchnorm-relu"):
        super().__init__()
        self.nonlinear = get_nonlinear(config_str, in_channels)
        self.linear = nn.Conv1d(in_channels, out_channels, 1, bias=bias)

    def run(self, x, R, T, tr):
        return FC.Conv1dFn.apply(_nonlinear(x, self.nonlinear, tr), (R, T, 1, 1), self.linear.weight, self.linear.bias)


class DenseLayer(nn.Module):
    def __init__(self, in_channels, out_channels, bias=False, config_str="batchnorm-relu"):
        super().__init__()
        self.linear = nn.Conv1d(in_channels, out_channels, 1, bias=bias)
        self.nonlinear = get_nonlinear(config_str, out_channels)

    def run(self, x, tr):
        """x [R, in] pooled statistics -> [R, out]."""
        y = FC.Conv1dFn.apply(x, (x.shape[0], 1, 1, 1), self.linear.weight, self.linear.bias)
        return _nonlinear(y, self.nonlinear, tr)


class CAMPPlus(nn.Module):
    def __init__(self, feat_dim=80, embed_dim=512, pooling_func="TSTP", growth_rate=32, bn_size=4, init_channels=128,
                 config_str="batchnorm-relu", memory_efficient=True):
        super().__init__()
        if pooling_func != "TSTP":
            raise NotImplementedError(f"CAM++ pooling_func {pooling_func!r}: TSTP (the upstream default) is built")
        if feat_dim % 8:
            raise NotImplementedError("CAM++: feat_dim must be a multiple of 8 (three mel-axis strides of 2)")
        from .resnet import TSTP
        self.head = FCM(feat_dim=feat_dim)
        channels = self.head.out_channels
        self.xvector = nn.Sequential(OrderedDict([
            ("tdnn", TDNNLayer(channels, init_channels, 5, stride=2, dilation=1, padding=-1, config_str=config_str))]))
        channels = init_channels
        for i, (num_layers, kernel_size, dilation) in enumerate(zip((12, 24, 16), (3, 3, 3), (1, 2, 2))):
            self.xvector.add_module("block%d" % (i + 1),
                                    CAMDenseTDNNBlock(num_layers=num_layers, in_channels=channels,
                                                      out_channels=growth_rate, bn_channels=bn_size * growth_rate,
                                                      kernel_size=kernel_size, dilation=dilation,
                                                      config_str=config_str, memory_efficient=memory_efficient))
            channels = channels + num_layers * growth_rate
            self.xvector.add_module("transit%d" % (i + 1),
                                    TransitLayer(channels, channels // 2, bias=False, config_str=config_str))
            channels //= 2
        self.xvector.add_module("out_nonlinear", get_nonlinear(config_str, channels))
        self.pool = TSTP(in_dim=channels)
        self.pool_out_dim = self.pool.get_out_dim()
        self.xvector.add_module("stats", self.pool)
        self.xvector.add_module("dense", DenseLayer(self.pool_out_dim, embed_dim, config_str="batchnorm_"))
        self.feat_dim, self.embed_dim = feat_dim, embed_dim
        for m in self.modules():
            if isinstance(m, (nn.Conv1d, nn.Linear)):
                nn.init.kaiming_normal_(m.weight.data)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def forward(self, x):
        """x [R, T, F] fbank -> embedding [R, embed_dim]."""
        if not x.is_cuda:
            from .._lib import WesepHipError
            raise WesepHipError("CAM++ speaker encoder: wesep_amd has no CPU path")
        R, T, Fq = x.shape
        tr = self.training
        y = x.float().transpose(1, 2).contiguous().view(R * Fq * T, 1)        # [R, F, T, 1]
        y = self.head.run(y, R, Fq, T, tr)                                    # [R*T, 32 * F/8]
        xv = self.xvector
        y, T = xv.tdnn.run(y, R, T, tr)
        for i in (1, 2, 3):
            y = getattr(xv, "block%d" % i).run(y, R, T, tr)
            y = getattr(xv, "transit%d" % i).run(y, R, T, tr)
        y = _nonlinear(y, xv.out_nonlinear, tr)
        stats = FR.TstpFn.apply(y, (R, 1, T))                                 # [R, 2C] mean || sqrt(var + 1e-7)
        return xv.dense.run(stats, tr)
