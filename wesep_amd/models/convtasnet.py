"""Conv-TasNet / SpEx+ on MI355X: constructor arguments, module tree and `state_dict` keys of the
reference `wesep.models.convtasnet.ConvTasNet` (wesep/models/convtasnet.py:14-219): the fixed-embedding
mode (`joint_training=False`: `forward(wav [R, T], emb [R, E])`, BASELINE.json configs[0]) and the SpEx+
joint mode (`joint_training=True, spk_feat=False`: `forward(wav, enrollment wav [R, Tw])`, speaker encoder
`ResNet4SpExplus` on the shared encoder, optional multi-task speaker logits as a fourth output).  `forward` is a chain of C-ABI launches (wesep_amd/functional_tasnet.py).

Built: Multi (SpEx+), Deep and plain (classic Conv-TasNet: one Conv1d + ReLU, mask, one ConvTranspose1d) encoder /
decoder pairs, every speaker-fusion type (concatConv / concat / additive / multiply / FiLM) with multi_fuse, gLN / cLN /
BN, causal blocks, skip connections, ReLU (all) or sigmoid (Deep / plain) masks, optional SpeakerTransform.  What still
raises NotImplementedError: mixing a Multi end with a non-Multi one and joint training on a non-Multi encoder (both broken
in the reference itself), activate='softmax' (over the batch axis in the reference) and multi_fuse=False.  Joint training
runs the SpEx+ encoder on the shared Multi encoder (spk_feat=False) or any wespeaker encoder of models/resnet.py on
fbank enrollment features (spk_feat=True)."""
import torch
import torch.nn as nn

from ..modules.common.speaker import SpeakerTransform
from ..functional import LinearFn
from ..modules.tasnet import (DeepDecoder, DeepEncoder, FuseSeparation, MultiDecoder, MultiEncoder, ResNet4SpExplus,
                              apply_norm, select_norm)


class ConvTasNet(nn.Module):
    def __init__(self, N=512, L=16, B=128, H=512, P=3, X=8, R=3, spk_emb_dim=256, norm="gLN", activate="relu",
                 causal=False, skip_con=False, spk_fuse_type="concatConv", multi_fuse=True,
                 use_spk_transform=True, encoder_type="Multi", decoder_type="Multi", joint_training=True,
                 multi_task=False, spksInTrain=251, spk_model=None, spk_model_init=None, spk_model_freeze=False,
                 spk_args=None, spk_feat=False, feat_type="consistent"):
        super().__init__()
        if joint_training and not spk_feat and feat_type != "consistent":
            raise NotImplementedError("ConvTasNet joint training on raw enrollment audio: feat_type='consistent' (the "
                                      "SpEx+ speaker encoder on the shared encoder) is what the reference builds "
                                      "(convtasnet.py:95-99)")
        multi = encoder_type == "Multi"
        if multi != (decoder_type == "Multi"):
            raise NotImplementedError("ConvTasNet: a 'Multi' encoder needs the 'Multi' decoder and vice versa (the "
                                      "reference's forward fails otherwise, convtasnet.py:171-201)")
        if joint_training and not spk_feat and not multi:
            raise NotImplementedError("ConvTasNet joint training: ResNet4SpExplus takes the Multi encoder's 3 x 256 "
                                      "channels (tasnet/speaker.py:52-53); other encoders do not fit it in the reference")
        if activate not in ("relu", "sigmoid") or (multi and activate != "relu"):
            raise NotImplementedError("ConvTasNet: activate='relu' (all decoders) or 'sigmoid' (Deep / plain) is built")
        self.encoder_type, self.decoder_type, self.norm_type, self.activate = encoder_type, decoder_type, norm, activate
        self.joint_training, self.multi_task, self.spk_feat = joint_training, multi_task, spk_feat
        self.stride = L // 2
        if multi:
            self.encoder = MultiEncoder(in_channels=1, middle_channels=N, out_channels=B, kernel_size=L, stride=L // 2)
        else:
            if encoder_type == "Deep":
                self.encoder = DeepEncoder(1, N, L, stride=L // 2)
            else:
                self.encoder = nn.Sequential(nn.Conv1d(1, N, L, stride=L // 2, padding=0), nn.ReLU())
            self.LayerN_S = select_norm(norm, N)
            self.BottleN_S = nn.Conv1d(N, B, 1)
        if joint_training:                 # registration order of the reference: encoder, spk_model, pred_linear
            if spk_feat:                   # a wespeaker encoder on fbank enrollment features (convtasnet.py:100-115)
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
                    if spk_model_freeze:   # only with an initialisation file, like the reference (convtasnet.py:112-114)
                        for param in self.spk_model.parameters():
                            param.requires_grad = False
            else:
                self.spk_model = ResNet4SpExplus(in_channel=N, C_embedding=spk_emb_dim)
            if multi_task:
                self.pred_linear = nn.Linear(spk_emb_dim, spksInTrain)
        self.spk_transform = SpeakerTransform() if use_spk_transform else nn.Identity()
        self.separation = FuseSeparation(R, X, B, H, P, norm=norm, causal=causal, skip_con=skip_con,
                                         C_embedding=spk_emb_dim, spk_fuse_type=spk_fuse_type,
                                         multi_fuse=multi_fuse)
        if multi:
            self.decoder = MultiDecoder(in_channels=B, middle_channels=N, out_channels=1, kernel_size=L, stride=L // 2)
        else:
            if decoder_type == "Deep":
                self.decoder = DeepDecoder(N, L, stride=L // 2)
            else:
                self.decoder = nn.ConvTranspose1d(N, 1, L, stride=L // 2)
            self.gen_masks = nn.Conv1d(B, N, 1)

    def _wespeaker_embedding(self, fbank):
        """fbank [R, Te, F] -> (embedding [R, E], logits or None) (convtasnet.py:188-192)."""
        out = self.spk_model(fbank)
        emb = out[-1] if isinstance(out, tuple) else out
        logits = LinearFn.apply(emb, self.pred_linear.weight, self.pred_linear.bias) if self.multi_task else None
        return emb, logits

    def _single_path(self, x, embeddings):
        """Deep / plain ends (convtasnet.py:174-178,197-201): encoder -> norm -> bottleneck -> separation -> mask ->
        mask * encoder output -> decoder.  Returns what the reference returns: [R, T_out] (Deep decoder squeezes) or
        [R, 1, T_out] (the plain ConvTrans1D does not)."""
        from .. import functional_campplus as FP
        from .. import functional_ecapa as FE
        from .. import functional_tasnet as FT
        R = x.shape[0]
        if self.encoder_type == "Deep":
            w, Tp = self.encoder(x)
        else:
            conv = self.encoder[0]
            w = FT.PlainEncoderFn.apply(x, self.stride, True, conv.weight, conv.bias)
            Tp = w.shape[0] // R
        geo = (R, Tp)
        e = apply_norm(self.LayerN_S, self.norm_type, w, geo, self.training)
        e = FP.Conv1dFn.apply(e, (R, Tp, 1, 1), self.BottleN_S.weight, self.BottleN_S.bias)
        logits = None
        if self.joint_training:
            embeddings, logits = self._wespeaker_embedding(embeddings)
        e = self.separation(e, self.spk_transform(embeddings), geo)
        gw = self.gen_masks.weight.view(self.gen_masks.weight.shape[0], -1)
        if self.activate == "relu":
            m = FE.LinearReluFn.apply(e, gw, self.gen_masks.bias)
        else:
            m = FE.RowBiasActFn.apply(LinearFn.apply(e, gw, self.gen_masks.bias), None, 1, 3)
        s = FT.MulFn.apply(w, m)
        if self.decoder_type == "Deep":
            est = self.decoder(s, geo)
        else:
            est = FT.TransDecoderFn.apply(s, (R, Tp, self.stride), self.decoder.weight, self.decoder.bias).unsqueeze(1)
        return est if logits is None else [est, logits]         # convtasnet.py:203-207

    def forward(self, x, embeddings):
        """x [R, T] (or [T]), embeddings [R, E] -> [est1, est2, est3], each [R, (T'-1)*stride + L]
        (convtasnet.py:162-219); Deep / plain ends: one tensor."""
        if x.dim() >= 3:
            raise RuntimeError("{} accept 1/2D tensor as input, but got {:d}".format("ConvTasNet", x.dim()))
        if x.dim() == 1:
            x = torch.unsqueeze(x, 0)
        x = x.contiguous().float()
        if self.encoder_type != "Multi":
            return self._single_path(x, embeddings.contiguous().float())
        e, cat, Tp = self.encoder(x)
        geo = (x.shape[0], Tp)
        logits = None
        embeddings = embeddings.contiguous().float()
        if self.joint_training and self.spk_feat:
            embeddings, logits = self._wespeaker_embedding(embeddings)
        elif self.joint_training:          # enrollment waveform through the SHARED encoder (convtasnet.py:179-187)
            _, cat_aux, Tpa = self.encoder(embeddings)
            embeddings = self.spk_model(cat_aux, (x.shape[0], Tpa))
            if self.multi_task:
                logits = LinearFn.apply(embeddings, self.pred_linear.weight, self.pred_linear.bias)
        emb = self.spk_transform(embeddings)
        e = self.separation(e, emb, geo)
        s = self.decoder(e, cat, geo)
        if logits is not None:
            s.append(logits)
        return s
