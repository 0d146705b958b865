"""Conv-TasNet / SpEx+ on MI355X: constructor arguments, module tree and `state_dict` keys of the
reference `wesep.models.convtasnet.ConvTasNet` (wesep/models/convtasnet.py:14-219): the fixed-embedding
mode (`joint_training=False`: `forward(wav [R, T], emb [R, E])`, BASELINE.json configs[0]) and the SpEx+
joint mode (`joint_training=True, spk_feat=False`: `forward(wav, enrollment wav [R, Tw])`, speaker encoder
`ResNet4SpExplus` on the shared encoder, optional multi-task speaker logits as a fourth output).  `forward` is a chain of C-ABI launches (wesep_amd/functional_tasnet.py).

Built: Multi encoder / decoder, every speaker-fusion type (concatConv / concat / additive / multiply / FiLM) with
multi_fuse, gLN / cLN, ReLU masks, optional SpeakerTransform.  Everything else of the reference constructor raises
NotImplementedError: Deep / plain encoders, skip connections, causal blocks, norm='BN' in the
separator, and joint training with a wespeaker encoder on fbank features."""
import torch
import torch.nn as nn

from ..modules.common.speaker import SpeakerTransform
from ..functional import LinearFn
from ..modules.tasnet import FuseSeparation, MultiDecoder, MultiEncoder, ResNet4SpExplus


class ConvTasNet(nn.Module):
    def __init__(self, N=512, L=16, B=128, H=512, P=3, X=8, R=3, spk_emb_dim=256, norm="gLN", activate="relu",
                 causal=False, skip_con=False, spk_fuse_type="concatConv", multi_fuse=True,
                 use_spk_transform=True, encoder_type="Multi", decoder_type="Multi", joint_training=True,
                 multi_task=False, spksInTrain=251, spk_model=None, spk_model_init=None, spk_model_freeze=False,
                 spk_args=None, spk_feat=False, feat_type="consistent"):
        super().__init__()
        if joint_training and (spk_feat or feat_type != "consistent"):
            raise NotImplementedError("ConvTasNet joint training with a wespeaker model on fbank features (SURVEY "
                                      "section 8 row a12) is not built; the SpEx+ speaker encoder on the shared "
                                      "encoder (spk_feat=False, feat_type='consistent') is")
        if encoder_type != "Multi" or decoder_type != "Multi":
            raise NotImplementedError("ConvTasNet: only encoder_type = decoder_type = 'Multi' (SpEx+) is built")
        if activate != "relu":
            raise NotImplementedError("ConvTasNet: only activate='relu' is built")
        self.encoder_type, self.decoder_type = encoder_type, decoder_type
        self.joint_training, self.multi_task = joint_training, multi_task
        self.encoder = MultiEncoder(in_channels=1, middle_channels=N, out_channels=B, kernel_size=L, stride=L // 2)
        if joint_training:                 # registration order of the reference: encoder, spk_model, pred_linear
            self.spk_model = ResNet4SpExplus(in_channel=N, C_embedding=spk_emb_dim)
            if multi_task:
                self.pred_linear = nn.Linear(spk_emb_dim, spksInTrain)
        self.spk_transform = SpeakerTransform() if use_spk_transform else nn.Identity()
        self.separation = FuseSeparation(R, X, B, H, P, norm=norm, causal=causal, skip_con=skip_con,
                                         C_embedding=spk_emb_dim, spk_fuse_type=spk_fuse_type,
                                         multi_fuse=multi_fuse)
        self.decoder = MultiDecoder(in_channels=B, middle_channels=N, out_channels=1, kernel_size=L, stride=L // 2)

    def forward(self, x, embeddings):
        """x [R, T] (or [T]), embeddings [R, E] -> [est1, est2, est3], each [R, (T'-1)*stride + L]
        (convtasnet.py:162-219)."""
        if x.dim() >= 3:
            raise RuntimeError("{} accept 1/2D tensor as input, but got {:d}".format("ConvTasNet", x.dim()))
        if x.dim() == 1:
            x = torch.unsqueeze(x, 0)
        x = x.contiguous().float()
        e, cat, Tp = self.encoder(x)
        geo = (x.shape[0], Tp)
        logits = None
        embeddings = embeddings.contiguous().float()
        if self.joint_training:            # enrollment waveform through the SHARED encoder (convtasnet.py:179-187)
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
