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
    """norm.py:62-76."""
    if norm == "gLN":
        return GlobalChannelLayerNorm(dim)
    if norm == "cLN":
        return nn.LayerNorm(dim, elementwise_affine=True)
    if norm == "BN":
        return nn.BatchNorm1d(dim)
    raise RuntimeError("Unsupported normalize layer: {}".format(norm))


def _bn_geo(norm, n1, n2, training):
    """The BatchNorm1d buffers of a block's two norms for ConvBlockFn (None unless norm == 'BN')."""
    if norm != "BN":
        return None
    if training:
        n1.num_batches_tracked += 1
        n2.num_batches_tracked += 1
    return (n1.running_mean, n1.running_var, n2.running_mean, n2.running_var, training)


def apply_norm(m, norm, x, geo, training):
    """select_norm module `m` on channels-last x [R*T', C]."""
    from ... import functional_campplus as FP
    from ... import functional_tfgridnet as FG
    if norm == "gLN":
        return FG.GroupLNFn.apply(x, m.weight.view(-1), m.bias.view(-1), geo)
    if norm == "cLN":
        return FG.RowLNFn.apply(x, m.weight, m.bias)
    if training:
        m.num_batches_tracked += 1
    return FP.BnActFn.apply(x, m.weight, m.bias, m.running_mean, m.running_var, training, False)


class Conv1DBlock(nn.Module):
    """convs.py:41-104: names conv1x1 / PReLU_1 / norm_1 / dwconv / PReLU_2 / norm_2 / (Sc_conv) / Output.  With
    skip_con the block returns (skip, out) like the reference."""

    def __init__(self, in_channels=256, out_channels=512, kernel_size=3, dilation=1, norm="gLN", causal=False,
                 skip_con=False):
        super().__init__()
        self.conv1x1 = nn.Conv1d(in_channels, out_channels, 1)
        self.PReLU_1 = nn.PReLU()
        self.norm_1 = select_norm(norm, out_channels)
        pad = dilation * (kernel_size - 1) // 2 if not causal else dilation * (kernel_size - 1)
        self.dwconv = nn.Conv1d(out_channels, out_channels, kernel_size, groups=out_channels, padding=pad,
                                dilation=dilation)
        self.PReLU_2 = nn.PReLU()
        self.norm_2 = select_norm(norm, out_channels)
        if skip_con:
            self.Sc_conv = nn.Conv1d(out_channels, in_channels, 1, bias=True)
        self.Output = nn.Conv1d(out_channels, in_channels, 1, bias=True)
        self.norm_type, self.dilation, self.causal, self.skip_con = norm, dilation, causal, skip_con

    def forward(self, x, geo):
        R, Tp = geo
        g = (R, Tp, self.norm_type, self.dilation, self.causal, _bn_geo(self.norm_type, self.norm_1, self.norm_2,
                                                                         self.training))
        skip = (self.Sc_conv.weight, self.Sc_conv.bias) if self.skip_con else ()
        res = FT.ConvBlockFn.apply(
            x, None, g, self.conv1x1.weight, self.conv1x1.bias,
            self.PReLU_1.weight, self.norm_1.weight, self.norm_1.bias, self.dwconv.weight, self.dwconv.bias,
            self.PReLU_2.weight, self.norm_2.weight, self.norm_2.bias, self.Output.weight, self.Output.bias, *skip)
        if self.skip_con:
            out, sc = res
            return sc, out
        return res


class Conv1DBlock4Fuse(nn.Module):
    """convs.py:107-160 (concatConv speaker fusion): names conv1x1 / prelu1 / lnorm1 / dconv / prelu2 /
    lnorm2 / sconv.  conv1x1(cat[x, aux]) = W_x x + (W_e e + b): the second term is one [R, H] GEMM."""

    def __init__(self, in_channels=256, spk_embed_dim=100, conv_channels=512, kernel_size=3, dilation=1,
                 norm="cLN", causal=False):
        super().__init__()
        self.conv1x1 = nn.Conv1d(in_channels + spk_embed_dim, conv_channels, 1)
        self.prelu1 = nn.PReLU()
        self.lnorm1 = select_norm(norm, conv_channels)
        pad = dilation * (kernel_size - 1) // 2 if not causal else dilation * (kernel_size - 1)
        self.dconv = nn.Conv1d(conv_channels, conv_channels, kernel_size, groups=conv_channels, padding=pad,
                               dilation=dilation, bias=True)
        self.prelu2 = nn.PReLU()
        self.lnorm2 = select_norm(norm, conv_channels)
        self.sconv = nn.Conv1d(conv_channels, in_channels, 1, bias=True)
        self.norm_type, self.dilation, self.in_channels, self.causal = norm, dilation, in_channels, causal

    def forward(self, x, aux, geo):
        """aux: speaker embedding [R, E]."""
        from ...functional import LinearFn
        R, Tp = geo
        w = self.conv1x1.weight
        rb = LinearFn.apply(aux, w[:, self.in_channels:, 0], self.conv1x1.bias)       # [R, H]
        g = (R, Tp, self.norm_type, self.dilation, self.causal, _bn_geo(self.norm_type, self.lnorm1, self.lnorm2,
                                                                         self.training))
        return FT.ConvBlockFn.apply(
            x, rb, g, w, self.conv1x1.bias, self.prelu1.weight,
            self.lnorm1.weight, self.lnorm1.bias, self.dconv.weight, self.dconv.bias, self.prelu2.weight,
            self.lnorm2.weight, self.lnorm2.bias, self.sconv.weight, self.sconv.bias)


class Separation(nn.Module):
    """separation.py:7-54: with skip_con the sum of the blocks' skip outputs is returned (the last block's residual
    output is dropped, as in the reference)."""

    def __init__(self, R, X, B, H, P, norm="gLN", causal=False, skip_con=False, start_dilation=0):
        super().__init__()
        self.separation = nn.ModuleList([])
        for _ in range(R):
            for x in range(start_dilation, X):
                self.separation.append(Conv1DBlock(B, H, P, 2 ** x, norm, causal, skip_con))
        self.skip_con = skip_con

    def forward(self, x, geo):
        if self.skip_con:
            total = None
            for blk in self.separation:
                skip, x = blk(x, geo)
                total = skip if total is None else total + skip
            return total
        for blk in self.separation:
            x = blk(x, geo)
        return x


class _FuseLayer(nn.Module):
    """SpeakerFuseLayer container on channels-last [R*T', B] (speaker.py:63-125, 3-D branch; FiLM: norm.py:84-137):
    `fc.linear` (concat: over [x | e]; additive / multiply: e -> B) or `fc.gamma_fcs.0` / `fc.beta_fcs.0`."""

    def __init__(self, embed_dim, feat_dim, fuse_type):
        super().__init__()
        from ..common.speaker import LinearLayer
        self.fuse_type, self.feat_dim = fuse_type, feat_dim
        if fuse_type == "FiLM":
            from ...models.dpccn import _FiLM
            self.fc = _FiLM(feat_dim, embed_dim)
        else:
            self.fc = LinearLayer(embed_dim + feat_dim if fuse_type == "concat" else embed_dim, feat_dim)

    def forward(self, x, emb, geo):
        from ...functional import LinearFn
        from ... import functional_ecapa as FE
        from ... import functional_dpccn as FD
        B = self.feat_dim
        if self.fuse_type == "FiLM":
            gm = LinearFn.apply(emb, self.fc.gamma_fcs[0].weight, self.fc.gamma_fcs[0].bias) + 1.0
            bt = LinearFn.apply(emb, self.fc.beta_fcs[0].weight, self.fc.beta_fcs[0].bias)
            return FE.RowBiasAddFn.apply(FE.GateFn.apply(x, gm, geo), bt, geo)
        w, b = self.fc.linear.weight, self.fc.linear.bias
        if self.fuse_type == "concat":      # Linear(cat[x, e]) = x Wx^T + (e We^T + b): the second term is a per-row bias
            rb = LinearFn.apply(emb, w[:, B:].contiguous(), b)
            y = FD.Conv1x1ResFn.apply(x, w[:, :B].contiguous(), torch.zeros_like(b), None)
            return FE.RowBiasAddFn.apply(y, rb, geo)
        s = LinearFn.apply(emb, w, b)                                                   # [R, B]
        return FE.GateFn.apply(x, s, geo) if self.fuse_type == "multiply" else FE.RowBiasAddFn.apply(x, s, geo)


class _PReLU(nn.PReLU):
    def forward(self, x, geo=None):
        from ... import functional_tfgridnet as FG
        return FG.PReluFn.apply(x, self.weight)


class FuseSeparation(nn.Module):
    """separation.py:57-186 with multi_fuse: concatConv (the SpEx+ configuration: a fusing first block per repeat), or
    SpeakerFuseLayer (concat / additive / multiply / FiLM) - PReLU - norm - all X blocks per repeat."""

    def __init__(self, R, X, B, H, P, norm="gLN", causal=False, skip_con=False, C_embedding=256,
                 spk_fuse_type="concatConv", multi_fuse=True):
        super().__init__()
        if spk_fuse_type not in ("concatConv", "concat", "additive", "multiply", "FiLM"):
            raise NotImplementedError(f"ConvTasNet spk_fuse_type={spk_fuse_type!r}")
        if not multi_fuse:
            raise NotImplementedError("ConvTasNet multi_fuse=False is broken in the reference "
                                      "(separation.py:146-186 replaces the ModuleList); not built")
        self.spk_fuse_type, self.norm_type, self.B = spk_fuse_type, norm, B
        self.separation = nn.ModuleList([])
        for _ in range(R):
            if spk_fuse_type == "concatConv":
                self.separation.append(Conv1DBlock4Fuse(spk_embed_dim=C_embedding, in_channels=B, conv_channels=H,
                                                        kernel_size=P, norm=norm, causal=causal, dilation=1))
                self.separation.append(Separation(1, X, B, H, P, norm=norm, causal=causal, skip_con=skip_con,
                                                  start_dilation=1))
            else:
                self.separation.append(_FuseLayer(C_embedding, B, spk_fuse_type))
                self.separation.append(_PReLU())
                self.separation.append(select_norm(norm, B))
                self.separation.append(Separation(1, X, B, H, P, norm=norm, causal=causal, skip_con=skip_con))

    def _norm(self, m, x, geo):
        return apply_norm(m, self.norm_type, x, geo, self.training)

    def forward(self, x, spk_embedding, geo):
        if self.spk_fuse_type == "concatConv":
            for i, m in enumerate(self.separation):
                x = m(x, spk_embedding, geo) if i % 2 == 0 else m(x, geo)
            return x
        for i, m in enumerate(self.separation):
            k = i % 4
            if k == 0:
                x = m(x, spk_embedding, geo)
            elif k == 1:
                x = m(x)
            elif k == 2:
                x = self._norm(m, x, geo)
            else:
                x = m(x, geo)
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


class DeepEncoder(nn.Module):
    """encoder.py:10-63: strided Conv1d, then four dilated k = 3 convolutions (1 / 2 / 4 / 8) each followed by a PReLU --
    on channels-last frames, every convolution one split-bf16 GEMM (functional_campplus.Conv1dFn)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride):
        super().__init__()
        if in_channels != 1:
            raise NotImplementedError("DeepEncoder: single-channel input only")
        layers = [nn.Conv1d(in_channels, out_channels, kernel_size, stride=stride)]
        for d in (1, 2, 4, 8):
            layers += [nn.Conv1d(out_channels, out_channels, kernel_size=3, stride=1, dilation=d, padding=d), nn.PReLU()]
        self.sequential = nn.Sequential(*layers)
        self.stride = stride

    def forward(self, x):
        """x [R, T] -> ([R*T', N], T')."""
        from ... import functional_campplus as FP
        from ... import functional_tfgridnet as FG
        seq = self.sequential
        y = FT.PlainEncoderFn.apply(x, self.stride, False, seq[0].weight, seq[0].bias)
        R = x.shape[0]
        Tp = y.shape[0] // R
        for i in (1, 3, 5, 7):
            y = FP.Conv1dFn.apply(y, (R, Tp, 1, seq[i].dilation[0]), seq[i].weight, seq[i].bias)
            y = FG.PReluFn.apply(y, seq[i + 1].weight)
        return y, Tp


class DeepDecoder(nn.Module):
    """decoder.py:7-63: four dilated k = 3 ConvTranspose1d (8 / 4 / 2 / 1) each followed by a PReLU, then the strided
    synthesis ConvTranspose1d.  A stride-1 transposed convolution is the convolution with the flipped, channel-swapped
    kernel (same 'same' padding), so the same GEMM serves it."""

    def __init__(self, N, kernel_size=16, stride=16 // 2):
        super().__init__()
        layers = []
        for d in (8, 4, 2, 1):
            layers += [nn.ConvTranspose1d(N, N, kernel_size=3, stride=1, dilation=d, padding=d), nn.PReLU()]
        layers.append(nn.ConvTranspose1d(N, 1, kernel_size=kernel_size, stride=stride, bias=True))
        self.sequential = nn.Sequential(*layers)
        self.stride = stride

    def forward(self, x, geo):
        """x [R*T', N] -> [R, (T' - 1) * stride + L]."""
        from ... import functional_campplus as FP
        from ... import functional_tfgridnet as FG
        R, Tp = geo
        seq = self.sequential
        for i in (0, 2, 4, 6):
            w = seq[i].weight.permute(1, 0, 2).flip(2).contiguous()            # [out, in, k] of the equivalent Conv1d
            x = FP.Conv1dFn.apply(x, (R, Tp, 1, seq[i].dilation[0]), w, seq[i].bias)
            x = FG.PReluFn.apply(x, seq[i + 1].weight)
        return FT.TransDecoderFn.apply(x, (R, Tp, self.stride), seq[8].weight, seq[8].bias)


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
