"""DPCCN on MI355X: constructor arguments, module tree and `state_dict` keys of the
reference `wesep.models.dpccn.DPCCN` (wesep/models/dpccn.py:15-290); `forward` is a chain of C-ABI launches
(wesep_amd/functional_dpccn.py) on channels-last [B*T*F, C] grids.  nn.Conv2d / nn.ConvTranspose2d / nn.Conv1d objects
are parameter containers only.

Built: fixed embeddings (`joint_training=False`) and joint training with a wespeaker ResNet18/34 on fbank or raw
enrollment audio (the same encoder / front-end as BSRNN, models/resnet.py); multiply / additive speaker fusion.
Not built (raise NotImplementedError): kernel sizes / strides other than the
defaults the reference's `_build_*` helpers are written for."""
import torch
import torch.nn as nn

from .. import functional as F_
from .. import functional_dpccn as FD
from ..modules.common.speaker import LinearLayer, SpeakerTransform


class Conv2dBlock(nn.Module):
    """conv2d - ELU - InstanceNorm2d (convs.py:28-50)."""

    def __init__(self, in_dims=16, out_dims=32, kernel_size=(3, 3), stride=(1, 1), padding=(1, 1)):
        super().__init__()
        self.conv2d = nn.Conv2d(in_dims, out_dims, kernel_size, stride, padding)
        self.stride = tuple(stride)

    def forward(self, x, geo):
        """x [B*H*W, Cin], geo (B, H, W) -> (y, (B, Ho, Wo))."""
        B, H, W = geo
        sh, sw = self.stride
        y = FD.Conv2dFn.apply(x, self.conv2d.weight, self.conv2d.bias, (B, H, W, sh, sw))
        Ho, Wo = (H + 2 - 3) // sh + 1, (W + 2 - 3) // sw + 1
        return FD.elu_inorm(y, (B, Ho * Wo), "pre"), (B, Ho, Wo)


class ConvTrans2dBlock(nn.Module):
    """conv_transpose2d - ELU - InstanceNorm2d (convs.py:53-77)."""

    def __init__(self, in_dims=32, out_dims=16, kernel_size=(3, 3), stride=(1, 2), padding=(1, 0),
                 output_padding=(0, 0)):
        super().__init__()
        self.convtrans2d = nn.ConvTranspose2d(in_dims, out_dims, kernel_size, stride, padding, output_padding)
        self.stride = tuple(stride)

    def forward(self, x, geo):
        B, H, W = geo
        sh, sw = self.stride
        y = FD.ConvTranspose2dFn.apply(x, self.convtrans2d.weight, self.convtrans2d.bias, (B, H, W, sh, sw))
        Ht, Wt = (H - 1) * sh - 2 + 3, (W - 1) * sw - 2 + 3
        return FD.elu_inorm(y, (B, Ht * Wt), "pre"), (B, Ht, Wt)


class DenseBlock(nn.Module):
    """Five densely connected Conv2dBlocks (convs.py:80-112); the concatenations are column concatenations."""

    def __init__(self, in_dims, out_dims, mode="enc", **kargs):
        super().__init__()
        if mode not in ["enc", "dec"]:
            raise RuntimeError("The mode option must be 'enc' or 'dec'!")
        n = 1 if mode == "enc" else 2
        self.conv1 = Conv2dBlock(in_dims=in_dims * n, out_dims=in_dims, **kargs)
        self.conv2 = Conv2dBlock(in_dims=in_dims * (n + 1), out_dims=in_dims, **kargs)
        self.conv3 = Conv2dBlock(in_dims=in_dims * (n + 2), out_dims=in_dims, **kargs)
        self.conv4 = Conv2dBlock(in_dims=in_dims * (n + 3), out_dims=in_dims, **kargs)
        self.conv5 = Conv2dBlock(in_dims=in_dims * (n + 4), out_dims=out_dims, **kargs)

    def forward(self, x, geo):
        import os
        convs = (self.conv1, self.conv2, self.conv3, self.conv4, self.conv5)
        if os.environ.get("WESEP_DENSE_FUSED", "1") != "0" and all(c.stride == (1, 1) for c in convs):
            params = [t for c in convs for t in (c.conv2d.weight, c.conv2d.bias)]
            return FD.DenseBlockFn.apply(x, geo, *params), geo       # one buffer for the growing map, no torch.cat
        feats = [x]
        for conv in convs:
            y, _ = conv(torch.cat(feats, 1) if len(feats) > 1 else feats[0], geo)
            feats.append(y)
        return feats[-1], geo


class _EncStage(nn.Sequential):
    def forward(self, x, geo):
        for m in self:
            x, geo = m(x, geo)
        return x, geo


class TCNBlock(nn.Module):
    """IN - ELU - depthwise dilated conv - IN - ELU - 1x1 conv, + residual (convs.py:115-152); causal: every depthwise
    tap at or before t (padding dil * (k - 1), tail cut, convs.py:126-127,147-148)."""

    def __init__(self, in_dims=384, out_dims=384, kernel_size=3, dilation=1, causal=False):
        super().__init__()
        pad = dilation * (kernel_size - 1) // 2 if not causal else dilation * (kernel_size - 1)
        self.dconv1 = nn.Conv1d(in_dims, out_dims, kernel_size, padding=pad, dilation=dilation, groups=in_dims, bias=True)
        self.dconv2 = nn.Conv1d(in_dims, out_dims, 1, bias=True)
        self.dilation, self.causal = dilation, causal

    def forward(self, x, geo):
        """x [B*L, D], geo (B, L)."""
        B, Lr = geo
        y = FD.elu_inorm(x, (B, Lr), "post")
        y = FD.DwConvFn.apply(y, self.dconv1.weight, self.dconv1.bias, (B, Lr, self.dilation, self.causal))
        y = FD.elu_inorm(y, (B, Lr), "post")
        return FD.Conv1x1ResFn.apply(y, self.dconv2.weight, self.dconv2.bias, x)


class _FiLM(nn.Module):
    """FiLM container (wesep/modules/common/norm.py:84-137, one layer): `gamma_fcs.0` / `beta_fcs.0` map the embedding to
    one scale / shift per frequency bin, zero-initialised like the reference."""

    def __init__(self, feat_size, embed_size):
        super().__init__()
        self.gamma_fcs = nn.ModuleList([nn.Linear(embed_size, feat_size)])
        self.beta_fcs = nn.ModuleList([nn.Linear(embed_size, feat_size)])
        for m in (self.gamma_fcs[0], self.beta_fcs[0]):
            nn.init.zeros_(m.weight)
            nn.init.zeros_(m.bias)


class _Fuse(nn.Module):
    """SpeakerFuseLayer container (speaker.py:63-79): `fc.linear` maps the embedding to one factor per frequency bin;
    FiLM: `fc.gamma_fcs.0` / `fc.beta_fcs.0`."""

    def __init__(self, embed_dim, feat_dim, fuse_type):
        super().__init__()
        if fuse_type not in ("multiply", "additive", "FiLM", "concat"):
            raise NotImplementedError(f"DPCCN spk_fuse_type={fuse_type!r}")
        self.fuse_type = fuse_type
        if fuse_type == "FiLM":
            self.fc = _FiLM(feat_dim, embed_dim)
        else:
            self.fc = LinearLayer(embed_dim + feat_dim if fuse_type == "concat" else embed_dim, feat_dim)


def fuse_bins(fuse, x, emb, geo):
    """SpeakerFuseLayer on x [B*T*F, C] with one factor / offset per (row, frequency bin) (speaker.py:102-125 on the
    [B, C, F, T] view; FiLM: norm.py:116-134, x = (1 + gamma(e)) x + beta(e))."""
    B, T, Fq = geo
    if fuse.fuse_type == "concat":
        # Linear over the FREQUENCY axis of cat[x, e] (speaker.py:95-101): out[b, c, :, t] = Wx x[b, c, :, t] + (We e + bias).
        # The contraction index is the spatial axis of the channels-last layout, so the tile is transposed to rows
        # (b, t, c) x F (zero-padded to a multiple of 4 floats), one GEMM + a per-row bias, and transposed back.
        from .. import functional_ecapa as FE
        C = x.shape[1]
        Fp = -(-Fq // 4) * 4
        w, b = fuse.fc.linear.weight, fuse.fc.linear.bias
        xt = torch.nn.functional.pad(x.view(B, T, Fq, C).permute(0, 1, 3, 2).reshape(B * T * C, Fq), (0, Fp - Fq))
        wx = torch.nn.functional.pad(w[:, :Fq], (0, Fp - Fq, 0, Fp - Fq))
        y = FD.Conv1x1ResFn.apply(xt.contiguous(), wx.contiguous(), torch.zeros(Fp, device=x.device), None)
        rb = torch.nn.functional.pad(F_.LinearFn.apply(emb, w[:, Fq:].contiguous(), b), (0, Fp - Fq))
        y = FE.RowBiasAddFn.apply(y, rb, (B, T * C))
        return y[:, :Fq].reshape(B, T, C, Fq).permute(0, 1, 3, 2).reshape(B * T * Fq, C).contiguous()
    if fuse.fuse_type == "FiLM":
        gm = F_.LinearFn.apply(emb, fuse.fc.gamma_fcs[0].weight, fuse.fc.gamma_fcs[0].bias) + 1.0    # [B, F]
        bt = F_.LinearFn.apply(emb, fuse.fc.beta_fcs[0].weight, fuse.fc.beta_fcs[0].bias)
        return FD.ScaleBFFn.apply(FD.ScaleBFFn.apply(x, gm, (B, T, Fq, 0)), bt, (B, T, Fq, 1))
    s = F_.LinearFn.apply(emb, fuse.fc.linear.weight, fuse.fc.linear.bias)                         # [B, F]
    return FD.ScaleBFFn.apply(x, s, (B, T, Fq, 0 if fuse.fuse_type == "multiply" else 1))


class DPCCN(nn.Module):
    def __init__(self, win=512, stride=128, spk_emb_dim=256, sr=16000, use_spk_transform=False,
                 spk_fuse_type="multiply", feature_dim=257, kernel_size=(3, 3), stride1=(1, 1), stride2=(1, 2),
                 paddings=(1, 1), output_padding=(0, 0), tcn_dims=384, tcn_blocks=10, tcn_layers=2, causal=False,
                 pool_size=(4, 8, 16, 32), multi_fuse=False, joint_training=True, multi_task=False, spksInTrain=251,
                 spk_model=None, spk_model_init=None, spk_model_freeze=False, spk_args=None, spk_feat=False,
                 feat_type="consistent"):
        super().__init__()
        if (tuple(kernel_size), tuple(stride1), tuple(stride2), tuple(paddings), tuple(output_padding)) != \
                ((3, 3), (1, 1), (1, 2), (1, 1), (0, 0)):
            raise NotImplementedError("DPCCN: kernel 3x3, strides (1,1)/(1,2), padding (1,1) only (the reference defaults)")
        if feature_dim != win // 2 + 1:
            raise RuntimeError("DPCCN: feature_dim must be win // 2 + 1")
        if joint_training and not spk_feat and feat_type != "consistent":
            raise NotImplementedError("DPCCN joint training with spk_feat=False: feat_type='consistent' only")
        self.win_len, self.hop_size, self.spk_emb_dim = win, stride, spk_emb_dim
        self.joint_training, self.spk_feat, self.feat_type = joint_training, spk_feat, feat_type
        self.spk_model_freeze, self.multi_task = spk_model_freeze, multi_task
        self.pool_size = tuple(pool_size)
        self.conv2d = nn.Conv2d(2, 16, kernel_size, stride1, paddings)
        enc = dict(kernel_size=kernel_size, stride=stride2, padding=paddings)
        self.encoder = nn.ModuleList([DenseBlock(16, 16, "enc")])
        for i in range(4):
            self.encoder.append(_EncStage(Conv2dBlock(in_dims=16 if i == 0 else 32, out_dims=32, **enc),
                                          DenseBlock(32, 32, "enc")))
        self.encoder.append(Conv2dBlock(in_dims=32, out_dims=64, **enc))
        self.encoder.append(Conv2dBlock(in_dims=64, out_dims=128, **enc))
        self.encoder.append(Conv2dBlock(in_dims=128, out_dims=384, **enc))
        self.spk_transform = SpeakerTransform() if use_spk_transform else nn.Identity()
        if joint_training:                  # dpccn.py:66-101, the same encoder / front-end as BSRNN
            from .resnet import get_speaker_model
            self.spk_model = get_speaker_model(spk_model)(**(spk_args or {}))
            if spk_model_init:
                pretrained = torch.load(spk_model_init, map_location="cpu")
                state = self.spk_model.state_dict()
                for key in state.keys():
                    if key in pretrained.keys():
                        state[key] = pretrained[key]
                    else:
                        print("not %s loaded" % key)
                self.spk_model.load_state_dict(state)
            if spk_model_freeze:
                for param in self.spk_model.parameters():
                    param.requires_grad = False
            if not spk_feat:
                from ..modules.common.frontend import MelSpectrogram, PreEmphasis
                self.preEmphasis = PreEmphasis()
                self.spk_encoder = MelSpectrogram(sample_rate=sr, n_fft=win, win_length=win, hop_length=stride, f_min=20,
                                                  n_mels=(spk_args or {})["feat_dim"])
            else:
                self.preEmphasis = nn.Identity()
                self.spk_encoder = nn.Identity()
            self.pred_linear = nn.Linear(spk_emb_dim, spksInTrain) if multi_task else nn.Identity()
        self.spk_fuse = _Fuse(spk_emb_dim, feature_dim, spk_fuse_type)
        self.tcn_layers = nn.Sequential(*[
            nn.Sequential(*[TCNBlock(in_dims=tcn_dims, out_dims=tcn_dims, causal=causal, dilation=2 ** b)
                            for b in range(tcn_blocks)]) for _ in range(tcn_layers)])
        dec = dict(kernel_size=kernel_size, stride=stride2, padding=paddings, output_padding=output_padding)
        self.decoder = nn.ModuleList([ConvTrans2dBlock(in_dims=384 * 2, out_dims=128, **dec),
                                      ConvTrans2dBlock(in_dims=128 * 2, out_dims=64, **dec),
                                      ConvTrans2dBlock(in_dims=64 * 2, out_dims=32, **dec)])
        for i in range(4):
            self.decoder.append(_EncStage(DenseBlock(32, 64, "dec"),
                                          ConvTrans2dBlock(in_dims=64, out_dims=32 if i != 3 else 16, **dec)))
        self.decoder.append(DenseBlock(16, 32, "dec"))
        self.avg_pool = nn.ModuleList([nn.Sequential(nn.AvgPool2d(sz), nn.Conv2d(32, 8, 1, 1)) for sz in pool_size])
        self.avg_proj = nn.Conv2d(32 + 8 * len(pool_size), 32, 1, 1)
        self.deconv2d = nn.ConvTranspose2d(32, 2, kernel_size, stride1, paddings)

    def forward(self, input, aux):
        """input [B, T] mixture; aux [B, E] (fixed) or fbank [B, Te, 80] / raw audio [B, Tw] (joint) ->
        (est [B, T], 0-d dummy or speaker logits)  (dpccn.py:206-290)."""
        if input.dim() != 2:
            raise RuntimeError("DPCCN expects a [batch, samples] mixture")
        wav = input.float().contiguous()
        B, nsample = wav.shape
        d = wav.device
        spec, Tf = FD.stft_ri(wav, self.win_len, self.hop_size)
        Fq = self.win_len // 2 + 1
        x4 = torch.zeros(B * Tf * Fq, 4, device=d, dtype=torch.float32)        # (re, im, 0, 0): 16-byte channel groups
        x4[:, :2] = spec[:, :2 * Fq].reshape(B * Tf * Fq, 2)
        geo = (B, Tf, Fq)
        c2 = self.conv2d
        w4 = torch.cat([c2.weight, torch.zeros(16, 2, 3, 3, device=d, dtype=torch.float32)], 1)
        out = FD.Conv2dFn.apply(x4, w4, c2.bias, (B, Tf, Fq, 1, 1))
        out, geo = self.encoder[0](out, geo)
        logits = torch.zeros((), device=d)      # (a fill on the stream; torch.tensor(0.0, device=d) synchronises it: models/bsrnn.py)
        emb = aux.float().contiguous()
        if self.joint_training:
            if not self.spk_feat:
                from ..modules.common.frontend import fbank_frontend
                emb = fbank_frontend(emb, self.preEmphasis, self.spk_encoder)
            o = self.spk_model(emb)
            emb = o[-1] if isinstance(o, tuple) else o
            # pred_linear is nn.Identity without multi_task: the reference then returns the embedding (dpccn.py:247)
            logits = (F_.LinearFn.apply(emb, self.pred_linear.weight, self.pred_linear.bias) if self.multi_task
                      else emb)
        emb = self.spk_transform(emb)
        out = fuse_bins(self.spk_fuse, out, emb, (B, Tf, Fq))
        skips = [(out, geo)]
        for enc in list(self.encoder)[1:]:
            out, geo = enc(out, geo)
            skips.append((out, geo))
        Bq, T2, F2 = geo
        for layer in self.tcn_layers:
            for blk in layer:
                out = blk(out, (B, T2 * F2))
        skips = skips[::-1]
        for idx, dec in enumerate(self.decoder):
            out, geo = dec(torch.cat([skips[idx][0], out], 1), geo)
        Bq, T3, F3 = geo
        pools = []
        for sz, avg in zip(self.pool_size, self.avg_pool):
            a = FD.AvgPoolFn.apply(out, (B, T3, F3, sz))
            a = FD.Conv2dFn.apply(a, avg[1].weight, avg[1].bias, (B, T3 // sz, F3 // sz, 1, 1))
            pools.append(FD.BilinearFn.apply(a, (B, T3 // sz, F3 // sz, T3, F3)))
        out = FD.Conv2dFn.apply(torch.cat([out, *pools], 1), self.avg_proj.weight, self.avg_proj.bias, (B, T3, F3, 1, 1))
        out = FD.ConvTranspose2dFn.apply(out, self.deconv2d.weight, self.deconv2d.bias, (B, T3, F3, 1, 1))  # [B*T*F, 2]
        if (T3, F3) != (Tf, Fq):
            raise RuntimeError("DPCCN: decoder grid does not match the spectrogram")
        ld = -(-2 * Fq // 4) * 4
        est_spec = torch.zeros(B * Tf, ld, device=d, dtype=torch.float32)
        est_spec[:, :2 * Fq] = out.reshape(B * Tf, 2 * Fq)
        est = FD.IstftFn.apply(est_spec, (B, Tf, nsample, self.win_len, self.hop_size))
        return est, logits
