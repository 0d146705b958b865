"""Conv-TasNet / SpEx+ on MI355X: constructor arguments, module tree and `state_dict` keys of the
reference `wesep.models.convtasnet.ConvTasNet` (wesep/models/convtasnet.py:14-219) in its
fixed-embedding mode (`joint_training=False`: `forward(wav [R, T], emb [R, E])`), the configuration of
BASELINE.json configs[0].  `forward` is a chain of C-ABI launches (wesep_amd/functional_tasnet.py).

Built: Multi encoder / decoder, concatConv fusion with multi_fuse, gLN / cLN, ReLU masks, optional
SpeakerTransform.  Everything else of the reference constructor raises NotImplementedError (see
DESIGN.md): Deep / plain encoders, skip connections, causal blocks, BatchNorm, other fusion types, and
joint training (ResNet4SpExplus / wespeaker encoders)."""
import torch
import torch.nn as nn

from ..modules.common.speaker import SpeakerTransform
from ..modules.tasnet import FuseSeparation, MultiDecoder, MultiEncoder


class ConvTasNet(nn.Module):
    def __init__(self, N=512, L=16, B=128, H=512, P=3, X=8, R=3, spk_emb_dim=256, norm="gLN", activate="relu",
                 causal=False, skip_con=False, spk_fuse_type="concatConv", multi_fuse=True,
                 use_spk_transform=True, encoder_type="Multi", decoder_type="Multi", joint_training=True,
                 multi_task=False, spksInTrain=251, spk_model=None, spk_model_init=None, spk_model_freeze=False,
                 spk_args=None, spk_feat=False, feat_type="consistent"):
        super().__init__()
        if joint_training:
            raise NotImplementedError("ConvTasNet joint_training=True (speaker encoder trained jointly, SURVEY "
                                      "section 8 row a12) is not built; pass fixed [R, E] embeddings")
        if encoder_type != "Multi" or decoder_type != "Multi":
            raise NotImplementedError("ConvTasNet: only encoder_type = decoder_type = 'Multi' (SpEx+) is built")
        if activate != "relu":
            raise NotImplementedError("ConvTasNet: only activate='relu' is built")
        self.encoder_type, self.decoder_type = encoder_type, decoder_type
        self.joint_training, self.multi_task = joint_training, multi_task
        self.encoder = MultiEncoder(in_channels=1, middle_channels=N, out_channels=B, kernel_size=L, stride=L // 2)
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
        emb = self.spk_transform(embeddings.contiguous().float())
        e = self.separation(e, emb, geo)
        return self.decoder(e, cat, geo)
