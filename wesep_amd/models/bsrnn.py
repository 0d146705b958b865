"""pBSRNN on MI355X: same constructor arguments, module tree and `state_dict` keys as the
reference `wesep.models.bsrnn.BSRNN` (wesep/models/bsrnn.py:151-298), so checkpoints and
`wesep/bin/train.py`-style callers are interchangeable -- but `forward` is a chain of
C-ABI launches into libwesep_hip.so (wesep_amd/functional.py), not ATen operators.

The torch.nn layer objects below (GroupNorm / LSTM / Linear / Conv1d) are PARAMETER CONTAINERS
ONLY: they give the reference's parameter names, shapes and default initialisation; their
own forward() is never called.  There is no CPU path: a CPU tensor raises.
"""
from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from .. import functional as F_
from ..modules.common.speaker import SpeakerFuseLayer, SpeakerTransform

_EPS = float(torch.finfo(torch.float32).eps)


class ResRNN(nn.Module):
    """GroupNorm(1,N) -> BLSTM(N -> 2N x2) -> Linear(4N -> N) -> + input  (bsrnn.py:16-46).
    `forward(z, view)`: z is the Z-layout tensor [R, K, Tf, N]; view 'time' runs the recurrence
    along Tf for every (r, k) (band_rnn), 'band' along K for every (r, t) (band_comm)."""

    def __init__(self, input_size, hidden_size, bidirectional=True):
        super().__init__()
        if not bidirectional:
            raise NotImplementedError("wesep_amd ResRNN kernels are bidirectional (reference default)")
        self.input_size, self.hidden_size, self.eps = input_size, hidden_size, _EPS
        self.norm = nn.GroupNorm(1, input_size, self.eps)
        self.rnn = nn.LSTM(input_size, hidden_size, 1, batch_first=True, bidirectional=True)
        self.proj = nn.Linear(hidden_size * 2, input_size)
        self._packs = F_.PackCache()     # derived weight forms, rebuilt when the weights change (not state)

    def invalidate_packs(self):
        """Drop the cached derived weight forms (MFMA-fragment packs, concatenated W_ih, ...).  The cache follows the
        weights through torch's version counters and the optimizer's weight epoch (functional.PackCache), which cover
        `optimizer.step()`, `load_state_dict`, `.to()` / `_apply` and every in-place op on the Parameter itself -- but
        NOT writes through `param.data` (EMA / model averaging / weight clipping code of the form
        `p.data.mul_(..)`): those do not move `_version`.  Call this (or `wesep_amd.dev.bump_weight_epoch()`) after
        such a write."""
        self._packs = F_.PackCache()

    def _apply(self, fn, recurse=True):   # .to() / .cuda() / .float(): new storages, new packs
        self._packs = F_.PackCache()
        return super()._apply(fn, recurse)

    def _load_from_state_dict(self, *args, **kwargs):
        self._packs = F_.PackCache()
        return super()._load_from_state_dict(*args, **kwargs)

    def _wparams(self):
        r = self.rnn
        return (r.weight_ih_l0, r.weight_hh_l0, r.bias_ih_l0, r.bias_hh_l0,
                r.weight_ih_l0_reverse, r.weight_hh_l0_reverse, r.bias_ih_l0_reverse, r.bias_hh_l0_reverse,
                self.proj.weight, self.proj.bias)

    def make_carrier(self):
        """Weight-gradient carrier of this layer (functional.WGradCarrierFn) or None."""
        return F_.make_wgrad_carrier(self._wparams())

    def forward(self, z, view="time", carrier=None):
        return F_.resrnn(z, view, self.norm.weight, self.norm.bias, *self._wparams(), carrier=carrier,
                         cache=self._packs)


class BSNet(nn.Module):
    """band_rnn then band_comm (bsrnn.py:55-83); no permute copies: both are views of Z."""

    def __init__(self, in_channel, nband=7, bidirectional=True):
        super().__init__()
        self.nband = nband
        self.feature_dim = in_channel // nband
        self.band_rnn = ResRNN(self.feature_dim, self.feature_dim * 2, bidirectional)
        self.band_comm = ResRNN(self.feature_dim, self.feature_dim * 2, bidirectional)

    def forward(self, z, dummy: Optional[torch.Tensor] = None, carriers=None):
        c_t, c_b = carriers if carriers is not None else (None, None)
        return self.band_comm(self.band_rnn(z, "time", c_t), "band", c_b)


class FuseSeparation(nn.Module):
    """bsrnn.py:86-148 (same `separation` ModuleList indexing, hence the same state_dict keys)."""

    def __init__(self, nband=7, num_repeat=6, feature_dim=128, spk_emb_dim=256, spk_fuse_type="concat",
                 multi_fuse=True):
        super().__init__()
        self.multi_fuse, self.nband, self.feature_dim = multi_fuse, nband, feature_dim
        self.separation = nn.ModuleList([])
        fuse = lambda: SpeakerFuseLayer(embed_dim=spk_emb_dim, feat_dim=feature_dim, fuse_type=spk_fuse_type)
        if multi_fuse:
            for _ in range(num_repeat):
                self.separation.append(fuse())
                self.separation.append(BSNet(nband * feature_dim, nband))
        else:
            self.separation.append(fuse())
            for _ in range(num_repeat):
                self.separation.append(BSNet(nband * feature_dim, nband))

    def make_carriers(self, device):
        """Weight-gradient carriers of every ResRNN (functional.WGradCarrierFn).  Autograd runs a node the later the EARLIER it
        was created: made before every ResRNN, they run after every ResRNN's backward, so the side-stream weight-gradient GEMMs
        overlap the following layers.  BSRNN.forward makes them before the band split too (round 6): the band split's backward
        -- 0.7 ms of BN weight gradients -- then runs BEFORE them, under the side stream's last jobs, instead of behind the wait
        for those jobs (profiles/r06_side_stream_tax.md: the main queue sat idle for 1.45 ms there)."""
        if device.type == "cuda":
            F_.reset_deferred_wgrads(device)
            if torch.is_grad_enabled():      # the step's weight packs, ahead of the layers, on the idle side stream (round 6)
                F_.prefetch_packs([(m._packs, m._wparams()) for l in self.separation if isinstance(l, BSNet)
                                   for m in (l.band_rnn, l.band_comm)])
        return {i: (l.band_rnn.make_carrier(), l.band_comm.make_carrier())
                for i, l in enumerate(self.separation) if isinstance(l, BSNet)}

    def forward(self, z, spk_embedding, nch=None, carriers=None):
        if carriers is None:
            carriers = self.make_carriers(z.device)
        for i, layer in enumerate(self.separation):
            if isinstance(layer, BSNet):
                z = layer(z, spk_embedding, carriers[i])
            else:
                z = layer(z, spk_embedding)
        return z


class BSRNN(nn.Module):
    def __init__(self, spk_emb_dim=256, sr=16000, win=512, stride=128, feature_dim=128, num_repeat=6,
                 use_spk_transform=True, use_bidirectional=True, spk_fuse_type="concat", multi_fuse=True,
                 joint_training=True, multi_task=False, spksInTrain=251, spk_model=None,
                 spk_model_init=None, spk_model_freeze=False, spk_args=None, spk_feat=False,
                 feat_type="consistent"):
        super().__init__()
        if (win, stride) != (512, 128):
            raise NotImplementedError("wesep_amd STFT kernels are built for win=512, stride=128")
        if feature_dim != 128:
            raise NotImplementedError("wesep_amd LSTM kernels are built for feature_dim=128 (hidden 256)")
        if joint_training and not spk_feat and feat_type != "consistent":
            raise NotImplementedError("joint_training with spk_feat=False: only feat_type='consistent' exists in the "
                                      "reference (bsrnn.py:231) and is built")
        self.sr, self.win, self.stride = sr, win, stride
        self.group = win // 2
        self.enc_dim = win // 2 + 1
        self.feature_dim = feature_dim
        self.eps = _EPS
        self.spk_emb_dim = spk_emb_dim
        self.joint_training, self.spk_feat, self.feat_type = joint_training, spk_feat, feat_type
        self.spk_model_freeze, self.multi_task = spk_model_freeze, multi_task

        # band table, bsrnn.py:190-209
        nyq = sr / 2.0
        bw = lambda hz: int(np.floor(hz / nyq * self.enc_dim))
        self.band_width = [bw(100)] * 15 + [bw(200)] * 10 + [bw(500)] * 5 + [bw(2000)]
        self.band_width.append(self.enc_dim - int(np.sum(self.band_width)))
        self.nband = len(self.band_width)
        if max(self.band_width) * 2 > 128:
            raise NotImplementedError("band wider than 64 bins")

        self.spk_transform = SpeakerTransform() if use_spk_transform else nn.Identity()

        if joint_training:                  # bsrnn.py:216-250, registration order of the reference
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
            if not spk_feat:                # raw enrollment audio: in-model fbank front-end (bsrnn.py:231-242)
                from ..modules.common.frontend import MelSpectrogram, PreEmphasis
                self.preEmphasis = PreEmphasis()
                self.spk_encoder = MelSpectrogram(sample_rate=sr, n_fft=win, win_length=win, hop_length=stride,
                                                  f_min=20, n_mels=(spk_args or {})["feat_dim"])
            else:
                self.preEmphasis = nn.Identity()
                self.spk_encoder = nn.Identity()
            self.pred_linear = nn.Linear(spk_emb_dim, spksInTrain) if multi_task else nn.Identity()

        self.BN = nn.ModuleList([
            nn.Sequential(nn.GroupNorm(1, b * 2, self.eps), nn.Conv1d(b * 2, feature_dim, 1))
            for b in self.band_width])
        self.separator = FuseSeparation(nband=self.nband, num_repeat=num_repeat, feature_dim=feature_dim,
                                        spk_emb_dim=spk_emb_dim, spk_fuse_type=spk_fuse_type,
                                        multi_fuse=multi_fuse)
        self.mask = nn.ModuleList([
            nn.Sequential(nn.GroupNorm(1, feature_dim, _EPS), nn.Conv1d(feature_dim, feature_dim * 4, 1),
                          nn.Tanh(), nn.Conv1d(feature_dim * 4, feature_dim * 4, 1), nn.Tanh(),
                          nn.Conv1d(feature_dim * 4, b * 4, 1))
            for b in self.band_width])
        self._plans = {}

    # -- parameter lists in the order the grouped kernels expect ---------------------------------
    def _bn_params(self):
        out = []
        for seq in self.BN:
            out += [seq[0].weight, seq[0].bias, seq[1].weight, seq[1].bias]
        return out

    def _mask_params(self):
        out = []
        for seq in self.mask:
            out += [seq[0].weight, seq[0].bias, seq[1].weight, seq[1].bias, seq[3].weight, seq[3].bias,
                    seq[5].weight, seq[5].bias]
        return out

    def _plan(self, device):
        key = (device.type, device.index)
        if key not in self._plans:
            self._plans[key] = F_.BandPlan(self.band_width, self.feature_dim, device)
        return self._plans[key]

    def _speaker(self, embeddings):
        """enrollment -> (fused-in embedding [R, E], second output) (bsrnn.py:339-360)."""
        # dummy, bsrnn.py:339-340.  torch.zeros: a fill on the stream -- torch.tensor(0.0, device=...) is a pageable host-to-device copy
        # that SYNCHRONISES the stream (the host sat out the whole previous step here, 71 of its 86 ms per step; round 6)
        predict_speaker_lable = torch.zeros((), device=embeddings.device)
        if self.joint_training:             # fbank [R, Te, F] -> wespeaker encoder -> embedding (bsrnn.py:341-357)
            if not self.spk_feat:           # raw enrollment waveform [R, Tw] -> log-mel, CMN (no_grad, :343-350)
                from ..modules.common.frontend import fbank_frontend
                embeddings = fbank_frontend(embeddings, self.preEmphasis, self.spk_encoder)
            out = self.spk_model(embeddings.float().contiguous())
            embeddings = out[-1] if isinstance(out, tuple) else out
            # pred_linear is nn.Identity without multi_task: the reference then returns the embedding (bsrnn.py:357)
            predict_speaker_lable = (F_.LinearFn.apply(embeddings, self.pred_linear.weight, self.pred_linear.bias)
                                     if self.multi_task else embeddings)
        return self.spk_transform(embeddings.float().contiguous()), predict_speaker_lable

    def forward(self, input, embeddings):
        """input: mixture [R, T] fp32; embeddings: [R, spk_emb_dim] (fixed) or fbank [R, Te, 80] (joint training)
        -> (est [R, T], 0-d dummy | speaker logits (multi_task) | the embedding (joint, no multi_task))."""
        if input.dim() != 2:
            raise RuntimeError("BSRNN expects a [batch, samples] mixture")
        wav = input.float().contiguous()
        plan = self._plan(wav.device)
        carriers = self.separator.make_carriers(wav.device) if hasattr(self.separator, "make_carriers") else None
        z, xbs = F_.BandSplitFn.apply(wav, plan, *self._bn_params())
        e, predict_speaker_lable = self._speaker(embeddings)
        z = self.separator(z, e, carriers=carriers) if carriers is not None else self.separator(z, e)
        est = F_.MaskDecodeFn.apply(z, xbs, plan, wav.shape[1], *self._mask_params())
        return est, predict_speaker_lable
