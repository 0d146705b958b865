"""Parameter containers of the Conv-TasNet / SpEx+ path with the reference's module and parameter names
(wesep/modules/tasnet/{convs,encoder,decoder,separation}.py, wesep/modules/common/norm.py).  As in
models/bsrnn.py the torch.nn layers only hold parameters (names, shapes, default initialisation);
every forward is a chain of C-ABI launches (wesep_amd/functional_tasnet.py) on channels-last tensors."""
import torch
import torch.nn as nn

from ... import functional_tasnet as FT


class GlobalChannelLayerNorm(nn.Module):
    """gLN parameters (norm.py:7-27): weight / bias [C, 1]."""

    def __init__(self, dim, eps=1e-05):
        super().__init__()
        self.dim, self.eps = dim, eps
        self.weight = nn.Parameter(torch.ones(dim, 1))
        self.bias = nn.Parameter(torch.zeros(dim, 1))


def select_norm(norm, dim):
    """norm.py:62-76.  BatchNorm is not built (needs running statistics kernels)."""
    if norm == "gLN":
        return GlobalChannelLayerNorm(dim)
    if norm == "cLN":
        return nn.LayerNorm(dim, elementwise_affine=True)
    if norm == "BN":
        raise NotImplementedError("ConvTasNet norm='BN' is not built in wesep_amd (gLN / cLN only)")
    raise RuntimeError("Unsupported normalize layer: {}".format(norm))


class Conv1DBlock(nn.Module):
    """convs.py:41-104 (non-causal, skip_con=False): names conv1x1 / PReLU_1 / norm_1 / dwconv / PReLU_2 /
    norm_2 / Output."""

    def __init__(self, in_channels=256, out_channels=512, kernel_size=3, dilation=1, norm="gLN", causal=False,
                 skip_con=False):
        super().__init__()
        if causal or skip_con:
            raise NotImplementedError("ConvTasNet causal / skip_con blocks are not built in wesep_amd")
        self.conv1x1 = nn.Conv1d(in_channels, out_channels, 1)
        self.PReLU_1 = nn.PReLU()
        self.norm_1 = select_norm(norm, out_channels)
        pad = dilation * (kernel_size - 1) // 2
        self.dwconv = nn.Conv1d(out_channels, out_channels, kernel_size, groups=out_channels, padding=pad,
                                dilation=dilation)
        self.PReLU_2 = nn.PReLU()
        self.norm_2 = select_norm(norm, out_channels)
        self.Output = nn.Conv1d(out_channels, in_channels, 1, bias=True)
        self.norm_type, self.dilation = norm, dilation

    def forward(self, x, geo):
        R, Tp = geo
        return FT.ConvBlockFn.apply(
            x, None, (R, Tp, self.norm_type, self.dilation), self.conv1x1.weight, self.conv1x1.bias,
            self.PReLU_1.weight, self.norm_1.weight, self.norm_1.bias, self.dwconv.weight, self.dwconv.bias,
            self.PReLU_2.weight, self.norm_2.weight, self.norm_2.bias, self.Output.weight, self.Output.bias)


class Conv1DBlock4Fuse(nn.Module):
    """convs.py:107-160 (concatConv speaker fusion): names conv1x1 / prelu1 / lnorm1 / dconv / prelu2 /
    lnorm2 / sconv.  conv1x1(cat[x, aux]) = W_x x + (W_e e + b): the second term is one [R, H] GEMM."""

    def __init__(self, in_channels=256, spk_embed_dim=100, conv_channels=512, kernel_size=3, dilation=1,
                 norm="cLN", causal=False):
        super().__init__()
        if causal:
            raise NotImplementedError("ConvTasNet causal blocks are not built in wesep_amd")
        self.conv1x1 = nn.Conv1d(in_channels + spk_embed_dim, conv_channels, 1)
        self.prelu1 = nn.PReLU()
        self.lnorm1 = select_norm(norm, conv_channels)
        pad = dilation * (kernel_size - 1) // 2
        self.dconv = nn.Conv1d(conv_channels, conv_channels, kernel_size, groups=conv_channels, padding=pad,
                               dilation=dilation, bias=True)
        self.prelu2 = nn.PReLU()
        self.lnorm2 = select_norm(norm, conv_channels)
        self.sconv = nn.Conv1d(conv_channels, in_channels, 1, bias=True)
        self.norm_type, self.dilation, self.in_channels = norm, dilation, in_channels

    def forward(self, x, aux, geo):
        """aux: speaker embedding [R, E]."""
        from ...functional import LinearFn
        R, Tp = geo
        w = self.conv1x1.weight
        rb = LinearFn.apply(aux, w[:, self.in_channels:, 0], self.conv1x1.bias)       # [R, H]
        return FT.ConvBlockFn.apply(
            x, rb, (R, Tp, self.norm_type, self.dilation), w, self.conv1x1.bias, self.prelu1.weight,
            self.lnorm1.weight, self.lnorm1.bias, self.dconv.weight, self.dconv.bias, self.prelu2.weight,
            self.lnorm2.weight, self.lnorm2.bias, self.sconv.weight, self.sconv.bias)


class Separation(nn.Module):
    """separation.py:7-54 without skip connections."""

    def __init__(self, R, X, B, H, P, norm="gLN", causal=False, skip_con=False, start_dilation=0):
        super().__init__()
        self.separation = nn.ModuleList([])
        for _ in range(R):
            for x in range(start_dilation, X):
                self.separation.append(Conv1DBlock(B, H, P, 2 ** x, norm, causal, skip_con))

    def forward(self, x, geo):
        for blk in self.separation:
            x = blk(x, geo)
        return x


class FuseSeparation(nn.Module):
    """separation.py:57-186, concatConv + multi_fuse (the SpEx+ configuration)."""

    def __init__(self, R, X, B, H, P, norm="gLN", causal=False, skip_con=False, C_embedding=256,
                 spk_fuse_type="concatConv", multi_fuse=True):
        super().__init__()
        if spk_fuse_type != "concatConv":
            raise NotImplementedError(f"ConvTasNet spk_fuse_type={spk_fuse_type!r}: only concatConv is built")
        if not multi_fuse:
            raise NotImplementedError("ConvTasNet multi_fuse=False is broken in the reference "
                                      "(separation.py:146-186 replaces the ModuleList); not built")
        self.separation = nn.ModuleList([])
        for _ in range(R):
            self.separation.append(Conv1DBlock4Fuse(spk_embed_dim=C_embedding, in_channels=B, conv_channels=H,
                                                    kernel_size=P, norm=norm, causal=causal, dilation=1))
            self.separation.append(Separation(1, X, B, H, P, norm=norm, causal=causal, skip_con=skip_con,
                                              start_dilation=1))

    def forward(self, x, spk_embedding, geo):
        for i, m in enumerate(self.separation):
            x = m(x, spk_embedding, geo) if i % 2 == 0 else m(x, geo)
        return x


class MultiEncoder(nn.Module):
    """encoder.py:66-114."""

    def __init__(self, in_channels, middle_channels, out_channels, kernel_size, stride):
        super().__init__()
        if in_channels != 1:
            raise NotImplementedError("MultiEncoder: single-channel input only")
        self.L1, self.L2, self.L3, self.stride = kernel_size, 80, 160, stride
        self.encoder_1d_short = nn.Conv1d(in_channels, middle_channels, self.L1, stride=stride)
        self.encoder_1d_middle = nn.Conv1d(in_channels, middle_channels, self.L2, stride=stride)
        self.encoder_1d_long = nn.Conv1d(in_channels, middle_channels, self.L3, stride=stride)
        self.ln = nn.LayerNorm(3 * middle_channels, elementwise_affine=True)
        self.proj = nn.Conv1d(3 * middle_channels, out_channels, 1)

    def forward(self, x):
        """x [R, T] -> (e [R*T', B], cat [R*T', 3N] = w1 | w2 | w3, T')."""
        e, cat = FT.MultiEncoderFn.apply(
            x, self.stride, self.encoder_1d_short.weight, self.encoder_1d_short.bias, self.encoder_1d_middle.weight,
            self.encoder_1d_middle.bias, self.encoder_1d_long.weight, self.encoder_1d_long.bias, self.ln.weight,
            self.ln.bias, self.proj.weight, self.proj.bias)
        return e, cat, (x.shape[-1] - self.L1) // self.stride + 1


class MultiDecoder(nn.Module):
    """decoder.py:66-114."""

    def __init__(self, in_channels, middle_channels, out_channels, kernel_size, stride):
        super().__init__()
        B, N, L = in_channels, middle_channels, kernel_size
        self.mask1, self.mask2, self.mask3 = nn.Conv1d(B, N, 1), nn.Conv1d(B, N, 1), nn.Conv1d(B, N, 1)
        self.decoder_1d_1 = nn.ConvTranspose1d(N, out_channels, kernel_size=L, stride=stride, bias=True)
        self.decoder_1d_2 = nn.ConvTranspose1d(N, out_channels, kernel_size=80, stride=stride, bias=True)
        self.decoder_1d_3 = nn.ConvTranspose1d(N, out_channels, kernel_size=160, stride=stride, bias=True)
        self.stride = stride

    def forward(self, e, cat, geo):
        R, Tp = geo
        return list(FT.MultiDecoderFn.apply(
            e, cat, (R, Tp, self.stride), self.mask1.weight, self.mask1.bias, self.mask2.weight, self.mask2.bias,
            self.mask3.weight, self.mask3.bias, self.decoder_1d_1.weight, self.decoder_1d_1.bias,
            self.decoder_1d_2.weight, self.decoder_1d_2.bias, self.decoder_1d_3.weight, self.decoder_1d_3.bias))


class ResBlock(nn.Module):
    """tasnet/speaker.py:7-44 (parameter container)."""

    def __init__(self, in_dims, out_dims):
        super().__init__()
        self.conv1 = nn.Conv1d(in_dims, out_dims, kernel_size=1, bias=False)
        self.conv2 = nn.Conv1d(out_dims, out_dims, kernel_size=1, bias=False)
        self.batch_norm1 = nn.BatchNorm1d(out_dims)
        self.batch_norm2 = nn.BatchNorm1d(out_dims)
        self.prelu1 = nn.PReLU()
        self.prelu2 = nn.PReLU()
        self.downsample = in_dims != out_dims
        if self.downsample:
            self.conv_downsample = nn.Conv1d(in_dims, out_dims, kernel_size=1, bias=False)

    def tensors(self):
        params = [self.conv1.weight, self.conv2.weight, self.batch_norm1.weight, self.batch_norm1.bias,
                  self.batch_norm2.weight, self.batch_norm2.bias, self.prelu1.weight, self.prelu2.weight,
                  self.conv_downsample.weight if self.downsample else None]
        buffers = [self.batch_norm1.running_mean, self.batch_norm1.running_var, self.batch_norm2.running_mean,
                   self.batch_norm2.running_var]
        return params, buffers


class ResNet4SpExplus(nn.Module):
    """tasnet/speaker.py:47-64: speaker encoder of SpEx+ on the shared encoder's [w1 | w2 | w3] (hard-wired to
    3 * 256 input channels like the reference)."""

    def __init__(self, in_channel=256, C_embedding=256):
        super().__init__()
        self.aux_enc3 = nn.Sequential(
            nn.LayerNorm(3 * in_channel, elementwise_affine=True),
            nn.Conv1d(3 * 256, 256, 1),
            ResBlock(256, 256),
            ResBlock(256, 512),
            ResBlock(512, 512),
            nn.Conv1d(512, C_embedding, 1),
        )

    def forward(self, cat_aux, geo):
        """cat_aux [R*T', 768] (channels-last ReLU outputs of the shared encoder), geo = (R, T')."""
        R, Tp = geo
        seq = self.aux_enc3
        params, buffers = [seq[0].weight, seq[0].bias, seq[1].weight, seq[1].bias], []
        for i in (2, 3, 4):
            p, b = seq[i].tensors()
            params += p
            buffers += b
            if self.training:
                seq[i].batch_norm1.num_batches_tracked += 1
                seq[i].batch_norm2.num_batches_tracked += 1
        params += [seq[5].weight, seq[5].bias]
        return FT.SpkEncoderFn.apply(cat_aux, (R, Tp, self.training), buffers, *params)
