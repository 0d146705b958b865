"""TF-GridNet on MI355X: constructor arguments, module tree and `state_dict` keys of the
reference `wesep.models.tfgridnet.TFGridNet` (wesep/models/tfgridnet.py:23-302) and `GridNetBlock`
(wesep/modules/tfgridnet/gridnet_block.py); `forward` is a chain of C-ABI launches (wesep_amd/functional_tfgridnet.py,
functional_dpccn.py) on channels-last [B*T*Q, C] grids.  nn.LSTM / nn.Conv2d / nn.LayerNorm objects are parameter
containers only.

Built: single microphone, one source, multiply / additive fusion, PReLU activation, fixed embeddings or joint training
with the wespeaker ResNet18/34 (fbank or raw enrollment audio), any emb_ks / emb_hs, lstm_hidden_units <= 256.
Not built (raise): multi-microphone input, n_srcs > 1, concat / FiLM fusion, eps != 1e-5."""
import math
import os

import torch
import torch.nn as nn
from torch.nn import init
from torch.nn.parameter import Parameter

from .. import dev
from .. import functional as F_
from .. import functional_dpccn as FD
from .. import functional_tfgridnet as FG
from ..modules.common.speaker import LinearLayer, SpeakerTransform


class LayerNormalization4DCF(nn.Module):
    """gridnet_block.py:230-255 (parameter container): gamma / beta [1, C, 1, F]."""

    def __init__(self, input_dimension, eps=1e-5):
        super().__init__()
        assert len(input_dimension) == 2
        param_size = [1, input_dimension[0], 1, input_dimension[1]]
        self.gamma = Parameter(torch.Tensor(*param_size).to(torch.float32))
        self.beta = Parameter(torch.Tensor(*param_size).to(torch.float32))
        init.ones_(self.gamma)
        init.zeros_(self.beta)
        self.eps = eps


class AllHeadPReLULayerNormalization4DCF(nn.Module):
    """gridnet_block.py:258-284 (parameter container): gamma / beta [1, H, E, 1, F], act = PReLU(H)."""

    def __init__(self, input_dimension, eps=1e-5):
        super().__init__()
        assert len(input_dimension) == 3
        H, E, n_freqs = input_dimension
        param_size = [1, H, E, 1, n_freqs]
        self.gamma = Parameter(torch.Tensor(*param_size).to(torch.float32))
        self.beta = Parameter(torch.Tensor(*param_size).to(torch.float32))
        init.ones_(self.gamma)
        init.zeros_(self.beta)
        self.act = nn.PReLU(num_parameters=H, init=0.25)
        self.eps, self.H, self.E, self.n_freqs = eps, H, E, n_freqs


def _heads(x, B, T, Q, nh, ch, norm):
    """x [B*T*Q, nh*ch] -> per head [B*T, Q*ch] rows after PReLU(head slope) + LN over (ch, Q) with the head's affine."""
    xh = x.view(B, T, Q, nh, ch).permute(3, 0, 1, 2, 4).contiguous()            # [nh, B, T, Q, ch]
    out = []
    for h in range(nh):
        v = FG.PReluFn.apply(xh[h].reshape(B * T * Q, ch), norm.act.weight[h:h + 1])
        g = norm.gamma[0, h, :, 0, :].t().reshape(-1)                           # [Q*ch], index q*ch + e
        b = norm.beta[0, h, :, 0, :].t().reshape(-1)
        out.append(FG.RowLNFn.apply(v.view(B * T, Q * ch), g, b))
    return out


class GridNetBlock(nn.Module):
    def __getitem__(self, key):
        return getattr(self, key)

    def __init__(self, emb_dim, emb_ks, emb_hs, n_freqs, hidden_channels, n_head=4, approx_qk_dim=512,
                 activation="prelu", eps=1e-5):
        super().__init__()
        if activation != "prelu":
            raise NotImplementedError("TF-GridNet: activation 'prelu' only (the reference asserts the same)")
        if abs(eps - 1e-5) > 1e-12:
            raise NotImplementedError("TF-GridNet: eps = 1e-5 only (norm kernels)")
        in_channels = emb_dim * emb_ks
        for path in ("intra", "inter"):
            setattr(self, f"{path}_norm", nn.LayerNorm(emb_dim, eps=eps))
            setattr(self, f"{path}_rnn", nn.LSTM(in_channels, hidden_channels, 1, batch_first=True, bidirectional=True))
            if emb_ks == emb_hs:
                setattr(self, f"{path}_linear", nn.Linear(hidden_channels * 2, in_channels))
            else:
                setattr(self, f"{path}_linear", nn.ConvTranspose1d(hidden_channels * 2, emb_dim, emb_ks, stride=emb_hs))
        E = math.ceil(approx_qk_dim * 1.0 / n_freqs)
        assert emb_dim % n_head == 0
        if (E * n_freqs) % 4 or (emb_dim // n_head * n_freqs) % 4:
            raise NotImplementedError("TF-GridNet: per-head widths E * n_freqs and (emb_dim / n_head) * n_freqs must be "
                                      "multiples of 4 (16-byte rows of the norm kernels); the shipped configuration has 520 / 780")
        self.add_module("attn_conv_Q", nn.Conv2d(emb_dim, n_head * E, 1))
        self.add_module("attn_norm_Q", AllHeadPReLULayerNormalization4DCF((n_head, E, n_freqs), eps=eps))
        self.add_module("attn_conv_K", nn.Conv2d(emb_dim, n_head * E, 1))
        self.add_module("attn_norm_K", AllHeadPReLULayerNormalization4DCF((n_head, E, n_freqs), eps=eps))
        self.add_module("attn_conv_V", nn.Conv2d(emb_dim, n_head * emb_dim // n_head, 1))
        self.add_module("attn_norm_V", AllHeadPReLULayerNormalization4DCF((n_head, emb_dim // n_head, n_freqs), eps=eps))
        self.add_module("attn_concat_proj", nn.Sequential(nn.Conv2d(emb_dim, emb_dim, 1), nn.PReLU(),
                                                          LayerNormalization4DCF((emb_dim, n_freqs), eps=eps)))
        self.emb_dim, self.emb_ks, self.emb_hs, self.n_head, self.E = emb_dim, emb_ks, emb_hs, n_head, E
        self.hidden = hidden_channels

    def _padded_lstm(self, path, device):
        C, ks, hs = self.emb_dim, self.emb_ks, self.emb_hs
        rnn = self[f"{path}_rnn"]
        perm = None
        if ks != hs:      # F.unfold orders a window as (channel, position); the row view as (position, channel)
            perm = (torch.arange(C, device=device).unsqueeze(0) * ks + torch.arange(ks, device=device).unsqueeze(1)).reshape(-1)
        wf, hf, bf = FG.pad_lstm(rnn.weight_ih_l0, rnn.weight_hh_l0, rnn.bias_ih_l0, rnn.bias_hh_l0, perm)
        wr, hr, br = FG.pad_lstm(rnn.weight_ih_l0_reverse, rnn.weight_hh_l0_reverse, rnn.bias_ih_l0_reverse,
                                 rnn.bias_hh_l0_reverse, perm)
        return wf, hf, bf, wr, hr, br

    def make_carriers(self):
        """Blocked path only: {path: (the eight weight tensors of functional_tfgridnet.BlstmLinearBlkFn, (dummy, box) | None)}
        for the intra- and the inter-frame BLSTM -- the weight-gradient carriers of functional.WGradCarrierFn.  Must be called
        for EVERY block before the first block's forward (the carriers need the lowest sequence numbers of the step's graph)."""
        C, ks, hs, h = self.emb_dim, self.emb_ks, self.emb_hs, self.hidden
        if not FG.blocked_path_ok(C, ks, hs):
            return None
        out = {}
        for path in ("intra", "inter"):
            lin = self[f"{path}_linear"]
            wf, hf, bf, wr, hr, br = self._padded_lstm(path, lin.weight.device)
            w = (wf, wr, bf, br, hf, hr, FG.pad_hidden_cols(lin.weight, h), lin.bias)
            out[path] = (w, F_.make_wgrad_carrier(w, blocked=True))
        return out

    def _rnn_path(self, path, x, nseq, Lr, strided=None, prep=None):
        """x [nseq*Lr, C]: LayerNorm -> windows -> BLSTM -> ConvTranspose1d / Linear -> + x  (gridnet_block.py:139-160).
        strided = (div, s1, s2, step_rows): the sequences are a strided row set of x instead of contiguous runs (blocked
        path only: functional_tfgridnet.BlstmLinearBlkFn).  prep: this path's entry of `make_carriers`."""
        C, ks, hs, h = self.emb_dim, self.emb_ks, self.emb_hs, self.hidden
        norm, rnn, lin = self[f"{path}_norm"], self[f"{path}_rnn"], self[f"{path}_linear"]
        y = FG.RowLNFn.apply(x, norm.weight, norm.bias)
        if FG.blocked_path_ok(C, ks, hs):     # the recipe's geometry: the pBSRNN blocked-layout recurrences
            if prep is None:
                wf, hf, bf, wr, hr, br = self._padded_lstm(path, x.device)
                w, carrier = (wf, wr, bf, br, hf, hr, FG.pad_hidden_cols(lin.weight, h), lin.bias), None
            else:
                w, carrier = prep
            dummy, box = carrier if carrier is not None else (None, None)
            return FG.BlstmLinearBlkFn.apply(y, x, (nseq, Lr) + tuple(strided or ()), dummy, box, *w)
        wf, hf, bf, wr, hr, br = self._padded_lstm(path, x.device)
        if strided is not None:
            raise dev.L.WesepHipError("TF-GridNet: strided sequence maps are a feature of the blocked-layout path")
        hcat = FG.BlstmFn.apply(y, (nseq, Lr, C, ks, hs), torch.cat([wf, wr], 0), torch.cat([bf, br], 0), hf, hr)
        n = (Lr - ks) // hs + 1
        if ks == hs:      # Linear(2h -> ks*C): frames tile the sequence without overlap
            W = FG.pad_hidden_cols(lin.weight, h)
            o = FD.Conv1x1ResFn.apply(hcat, W, lin.bias, None).view(nseq * Lr, C)
        else:             # ConvTranspose1d weight [2h, C, ks] -> rows i*C + c, columns = padded hidden
            Wt = FG.pad_hidden_cols(lin.weight.permute(2, 1, 0).reshape(ks * C, 2 * h), h)
            o = FG.Deconv1dFn.apply(hcat, (nseq, Lr, C, ks, hs, n), Wt)
            o = FG.AddRowVecFn.apply(o, lin.bias)
        return o + x

    def forward(self, x, geo, prep=None):
        """x [B*T*Q, C], geo (B, T, Q) -> same.  prep: `make_carriers()` of this block, or None."""
        prep = prep or {}
        B, oT, oQ = geo
        C, ks, hs, nh, E = self.emb_dim, self.emb_ks, self.emb_hs, self.n_head, self.E
        olp = ks - hs
        T = math.ceil((oT + 2 * olp - ks) / hs) * hs + ks
        Q = math.ceil((oQ + 2 * olp - ks) / hs) * hs + ks
        if (T, Q, olp) == (oT, oQ, 0):      # emb_ks == emb_hs and both axes already whole windows (the recipe): no padding,
            h = x.view(B, oT, oQ, C)        # and F.pad with all-zero pads would still copy the 150 MB map
        else:
            h = torch.nn.functional.pad(x.view(B, oT, oQ, C), (0, 0, olp, Q - oQ - olp, olp, T - oT - olp))
        h = self._rnn_path("intra", h.reshape(B * T * Q, C), B * T, Q, prep=prep.get("intra")).view(B, T, Q, C)
        if FG.blocked_path_ok(C, ks, hs) and os.environ.get("WESEP_TFG_STRIDED", "1") != "0":
            # inter-frame path IN PLACE on the [B, T, Q, C] map: sequence (b, q), step t -> row (b * T + t) * Q + q (round 4;
            # the reference permutes to [B, Q, T, C] and back, gridnet_block.py:163-180: two copies of the map forward, two backward)
            h = self._rnn_path("inter", h.view(B * T * Q, C), B * Q, T, strided=(Q, T * Q, 1, Q),
                               prep=prep.get("inter")).view(B, T, Q, C)
            inter = h[:, olp:olp + oT, olp:olp + oQ, :].contiguous().view(B * oT * oQ, C)
        else:
            h = h.transpose(1, 2).contiguous()                                      # [B, Q, T, C]
            h = self._rnn_path("inter", h.view(B * Q * T, C), B * Q, T, prep=prep.get("inter")).view(B, Q, T, C)
            inter = h.transpose(1, 2)[:, olp:olp + oT, olp:olp + oQ, :].contiguous().view(B * oT * oQ, C)
        M = B * oT * oQ
        cq, ck, cv = self["attn_conv_Q"], self["attn_conv_K"], self["attn_conv_V"]
        cp = C // nh
        D = oQ * E
        # all heads x batch rows of the block in one grouped launch per product (G = nh * B problems of [oT, oT]).
        # The key / value time axis is zero-padded to a multiple of 4 floats (16-byte rows for the GEMM operand loads);
        # padded key columns get a -1e30 bias in the logits' epilogue, i.e. exactly zero attention weight.
        G = nh * B
        Tp = -(-oT // 4) * 4
        if dev.heads_ok(oQ, nh, E) and dev.heads_ok(oQ, nh, cp) and os.environ.get("WESEP_TFG_HEADS_FUSED", "1") != "0":
            # one projection GEMM for Q, K and V (the block output is read once, its gradient is one GEMM instead of three
            # and two full-size sums), then one kernel per projection: PReLU + head LayerNorm + the head-major layout
            Wc = torch.cat([cq.weight, ck.weight, cv.weight], 0)
            bc = torch.cat([cq.bias, ck.bias, cv.bias], 0)
            qkv = FD.Conv1x1ResFn.apply(inter, Wc, bc, None)
            par = []
            for nm, ch in (("attn_norm_Q", E), ("attn_norm_K", E), ("attn_norm_V", cp)):
                norm = self[nm]
                par += [norm.act.weight, norm.gamma[0, :, :, 0, :].permute(0, 2, 1).reshape(nh, oQ * ch),
                        norm.beta[0, :, :, 0, :].permute(0, 2, 1).reshape(nh, oQ * ch)]
            Qa, Ka, Va = FG.QKVHeadsFn.apply(qkv, (B, oT, Tp, oQ, nh, E, cp), *par)
        else:
            q = FD.Conv1x1ResFn.apply(inter, cq.weight, cq.bias, None)
            k = FD.Conv1x1ResFn.apply(inter, ck.weight, ck.bias, None)
            v = FD.Conv1x1ResFn.apply(inter, cv.weight, cv.bias, None)
            qh = _heads(q, B, oT, oQ, nh, E, self["attn_norm_Q"])
            kh = _heads(k, B, oT, oQ, nh, E, self["attn_norm_K"])
            vh = _heads(v, B, oT, oQ, nh, cp, self["attn_norm_V"])
            Qa = torch.stack(qh, 0).view(G, oT, D)                                                      # group h * B + b
            Ka = torch.nn.functional.pad(torch.stack(kh, 0).view(G, oT, D), (0, 0, 0, Tp - oT))
            Va = torch.nn.functional.pad(torch.stack(vh, 0).view(G, oT, oQ * cp), (0, 0, 0, Tp - oT))
        mask = torch.zeros(Tp, device=x.device, dtype=torch.float32)
        mask[oT:] = -1e30
        logits = FG.BatchedMatmulNTFn.apply(Qa, Ka, mask)                                           # [G, oT, Tp]
        att = FG.SoftmaxFn.apply(logits.view(G * oT, Tp), 1.0 / math.sqrt(D)).view(G, oT, Tp)
        if FG.BatchedMatmulNTFn.nn_ok(Tp, oQ * cp, oT):          # att x V with V as the head kernel wrote it (round 6)
            ov = FG.BatchedMatmulNNFn.apply(att, Va)                                                # [G, oT, oQ * cp]
        else:
            ov = FG.BatchedMatmulNTFn.apply(att, Va.transpose(1, 2).contiguous(), None)
        o = ov.view(nh, B, oT, oQ, cp).permute(1, 2, 3, 0, 4).reshape(M, C)                         # channel h*cp + c
        proj = self["attn_concat_proj"]
        o = FD.Conv1x1ResFn.apply(o, proj[0].weight, proj[0].bias, None)
        o = FG.PReluFn.apply(o, proj[1].weight)
        g = proj[2].gamma[0, :, 0, :].t().reshape(-1)                                               # [Q*C], index q*C + c
        bt = proj[2].beta[0, :, 0, :].t().reshape(-1)
        o = FG.RowLNFn.apply(o.view(B * oT, oQ * C), g, bt).view(M, C)
        return o + inter


class _Fuse(nn.Module):
    def __init__(self, embed_dim, feat_dim, fuse_type):
        super().__init__()
        if fuse_type not in ("multiply", "additive", "FiLM", "concat"):
            raise NotImplementedError(f"TF-GridNet spk_fuse_type={fuse_type!r}")
        self.fuse_type = fuse_type
        from .dpccn import _FiLM
        if fuse_type == "FiLM":
            self.fc = _FiLM(feat_dim, embed_dim)
        else:
            self.fc = LinearLayer(embed_dim + feat_dim if fuse_type == "concat" else embed_dim, feat_dim)


class TFGridNet(nn.Module):
    def __init__(self, n_srcs=1, sr=16000, n_fft=128, stride=64, window="hann", n_imics=1, n_layers=6,
                 lstm_hidden_units=192, attn_n_head=4, attn_approx_qk_dim=512, emb_dim=48, emb_ks=4, emb_hs=1,
                 activation="prelu", eps=1.0e-5, spk_emb_dim=256, use_spk_transform=False, spk_fuse_type="multiply",
                 joint_training=True, multi_task=False, spksInTrain=251, spk_model=None, spk_model_init=None,
                 spk_model_freeze=False, spk_args=None, spk_feat=False, feat_type="consistent"):
        super().__init__()
        if n_srcs < 1 or n_imics < 1 or window != "hann":
            raise NotImplementedError("TF-GridNet: the hann window is built")
        if emb_dim % 4 or lstm_hidden_units > 256:
            raise NotImplementedError("TF-GridNet: emb_dim % 4 == 0 and lstm_hidden_units <= 256")
        if joint_training and not spk_feat and feat_type != "consistent":
            raise NotImplementedError("TF-GridNet joint training with spk_feat=False: feat_type='consistent' only")
        self.n_srcs, self.n_fft, self.stride, self.n_imics, self.n_layers = n_srcs, n_fft, stride, n_imics, n_layers
        self.spk_emb_dim, self.joint_training, self.spk_feat, self.feat_type = spk_emb_dim, joint_training, spk_feat, feat_type
        self.spk_model_freeze, self.multi_task = spk_model_freeze, multi_task
        assert n_fft % 2 == 0
        n_freqs = n_fft // 2 + 1
        self.spk_transform = SpeakerTransform() if use_spk_transform else nn.Identity()
        if joint_training:
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
                self.spk_encoder = MelSpectrogram(sample_rate=sr, n_fft=n_fft, win_length=n_fft, hop_length=stride,
                                                  f_min=20, n_mels=(spk_args or {})["feat_dim"])
            else:
                self.preEmphasis = nn.Identity()
                self.spk_encoder = nn.Identity()
            self.pred_linear = nn.Linear(spk_emb_dim, spksInTrain) if multi_task else nn.Identity()
        self.spk_fuse = _Fuse(spk_emb_dim, n_freqs, spk_fuse_type)
        self.conv = nn.Sequential(nn.Conv2d(2 * n_imics, emb_dim, (3, 3), padding=(1, 1)), nn.GroupNorm(1, emb_dim, eps=eps))
        self.blocks = nn.ModuleList([GridNetBlock(emb_dim, emb_ks, emb_hs, n_freqs, lstm_hidden_units, n_head=attn_n_head,
                                                  approx_qk_dim=attn_approx_qk_dim, activation=activation, eps=eps)
                                     for _ in range(n_layers)])
        self.deconv = nn.ConvTranspose2d(emb_dim, n_srcs * 2, (3, 3), padding=(1, 1))

    def _run_blocks(self, h, emb, geo):
        """The block stack on h [B*T*F, C] with one embedding per row of the batch (tfgridnet.py:272-276)."""
        from .dpccn import fuse_bins
        # weight-gradient carriers of all twelve BLSTMs first: autograd then runs them after every block's backward, and the
        # weight-gradient GEMMs run on the side stream under the inter-frame BPTTs (functional.WGradCarrierFn)
        preps = [blk.make_carriers() for blk in self.blocks]
        for blk, prep in zip(self.blocks, preps):
            h = fuse_bins(self.spk_fuse, h, emb, geo)                # the same fusion before every block
            h = blk(h, geo, prep)
        return h

    def forward(self, input, embeddings):
        """input [B, N] mixture ([B, N, M] with n_imics = M > 1); embeddings [B, E] (fixed) or fbank / raw audio (joint)
        -> (est [B, N] or [B, n_srcs, N], dummy or logits) (tfgridnet.py:197-302)."""
        M, S = self.n_imics, self.n_srcs
        if input.dim() != (2 if M == 1 else 3) or (M > 1 and input.shape[2] != M):
            raise RuntimeError(f"TFGridNet expects a [batch, samples] mixture (n_imics = 1) or [batch, samples, {M}]")
        wav = input.float().contiguous()
        B, n = wav.shape[0], wav.shape[1]
        d = wav.device
        if n % 4:
            raise NotImplementedError("TF-GridNet: the number of samples must be a multiple of 4 (16-byte rows)")
        with torch.no_grad():   # RMS normalisation by the (unbiased) standard deviation over samples and microphones
            from .. import dev
            st = torch.empty(B, 2, device=d, dtype=torch.float32)
            dev.flat_stats(wav.view(B, n * M), B, n * M, st, 0.0)
            std = torch.sqrt(1.0 / (st[:, 1] ** 2) * (n * M / (n * M - 1.0)))        # [B]
            inv = (1.0 / std).repeat_interleave(M).view(B * M, 1).contiguous()
            rows = wav if M == 1 else wav.transpose(1, 2).contiguous().view(B * M, n)   # [B, N, M] -> [B*M, N]
            x = torch.empty_like(rows)
            dev.scale_bf_fwd(rows, inv, B * M, 1, 1, n, 0, x)
        spec, Tf = FD.stft_ri(x, self.n_fft, self.stride)
        Fq = self.n_fft // 2 + 1
        Cp = -(-2 * M // 4) * 4                  # the implicit-patch operand moves 16-byte channel groups
        x4 = torch.zeros(B * Tf * Fq, Cp, device=d, dtype=torch.float32)
        # channels (re_0 .. re_{M-1}, im_0 .. im_{M-1}) like cat((real, imag), 1) of [B, M, T, F] (tfgridnet.py:241-244)
        x4[:, :2 * M] = spec[:, :2 * Fq].reshape(B, M, Tf, Fq, 2).permute(0, 2, 3, 4, 1).reshape(B * Tf * Fq, 2 * M)
        c0, gn = self.conv[0], self.conv[1]
        C = c0.weight.shape[0]
        w4 = torch.cat([c0.weight, torch.zeros(C, Cp - 2 * M, 3, 3, device=d, dtype=torch.float32)], 1)
        h = FD.Conv2dFn.apply(x4, w4, c0.bias, (B, Tf, Fq, 1, 1))
        h = FG.GroupLNFn.apply(h, gn.weight, gn.bias, (B, Tf * Fq))
        logits = torch.zeros((), device=d)      # (a fill on the stream; torch.tensor(0.0, device=d) synchronises it: models/bsrnn.py)
        emb = embeddings.float().contiguous()
        if self.joint_training:
            if not self.spk_feat:
                from ..modules.common.frontend import fbank_frontend
                emb = fbank_frontend(emb, self.preEmphasis, self.spk_encoder)
            o = self.spk_model(emb)
            emb = o[-1] if isinstance(o, tuple) else o
            # pred_linear is nn.Identity without multi_task: the reference then returns the embedding (tfgridnet.py:267)
            logits = (F_.LinearFn.apply(emb, self.pred_linear.weight, self.pred_linear.bias) if self.multi_task
                      else emb)
        emb = self.spk_transform(emb)
        if h.is_cuda:
            F_.reset_deferred_wgrads(h.device)
        # (Two halves of the batch on two HIP streams -- one half's inter-frame recurrence beside the other half's GEMMs -- was
        # built and measured in round 4: 516 ms instead of 325 ms per step at 8 rows, 761 ms with four quarters; the latency
        # chains of the halves add up instead of overlapping.  DESIGN section 10.)
        h = self._run_blocks(h, emb, (B, Tf, Fq))
        out = FD.ConvTranspose2dFn.apply(h, self.deconv.weight, self.deconv.bias, (B, Tf, Fq, 1, 1))   # [B*T*F, 2S]
        ld = -(-2 * Fq // 4) * 4
        est_spec = torch.zeros(B * S * Tf, ld, device=d, dtype=torch.float32)
        # channel 2s + (0 re | 1 im) of source s (tfgridnet.py:280-282): one spectrogram row block per (b, s)
        est_spec[:, :2 * Fq] = out.view(B, Tf, Fq, S, 2).permute(0, 3, 1, 2, 4).reshape(B * S * Tf, 2 * Fq)
        est = FD.IstftFn.apply(est_spec, (B * S, Tf, n, self.n_fft, self.stride))
        est = FD.ScaleBFFn.apply(est.contiguous(), std.repeat_interleave(S).view(B * S, 1).contiguous(), (B * S, 1, 1, 0))
        return (est if S == 1 else est.view(B, S, n)), logits
