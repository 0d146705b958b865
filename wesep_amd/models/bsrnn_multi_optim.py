"""`BSRNN_Multi`: pBSRNN with SSA multi-optimisation (wesep/models/bsrnn_multi_optim.py:156-472; recipe
examples/librimix/tse/v2/confs/bsrnn_multi_optim.yaml).

Same modules, parameters and `state_dict` keys as `BSRNN`.  In training (grad mode) the forward is two separator
passes over ONE band-split of the mixture (bsrnn_multi_optim.py:407-470):

    s       = decode(separator(z, embed(enrollment)))                    # as BSRNN
    self_s  = decode(separator(z, embed(s.detach())))                    # the estimate is its own enrollment
    return s, self_s, predict_speaker_lable, self_predict_speaker_lable  # loss_posi [[0, 1]], weights .4 / .6

and without grad it returns `(s, predict_speaker_lable)` like `BSRNN`.  On the device this reuses every BSRNN kernel:
the band-split features `z` and the mixture's band spectra are produced once and read by both passes (autograd sums
their two gradients into the band-split backward), and the speaker encoder / in-model fbank front-end runs twice.

The reference's second pass is only well-formed with `joint_training=True` (otherwise `self_spk_emb_input` is
unbound, :433) and `spk_feat=False` with `feat_type="consistent"` (the estimate is a waveform, the speaker encoder
wants filterbank features, :412-421) -- which is what the shipped recipe sets; other combinations raise here."""
import torch

from .. import functional as F_
from .bsrnn import BSRNN


class BSRNN_Multi(BSRNN):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        if not self.joint_training or self.spk_feat or self.feat_type != "consistent":
            raise NotImplementedError("BSRNN_Multi: the self-enrollment pass needs joint_training=True, spk_feat=False, "
                                      "feat_type='consistent' (bsrnn_multi_optim.yaml)")

    def forward(self, input, embeddings):
        """input [R, T] mixture, embeddings [R, Tw] enrollment waveform -> grad mode: (s, self_s, second output of
        pass 1, of pass 2); no-grad mode: (s, second output)."""
        if input.dim() != 2:
            raise RuntimeError("BSRNN_Multi expects a [batch, samples] mixture")
        wav = input.float().contiguous()
        plan = self._plan(wav.device)
        mask_params = self._mask_params()
        z, xbs = F_.BandSplitFn.apply(wav, plan, *self._bn_params())
        e, predict_speaker_lable = self._speaker(embeddings)
        s = F_.MaskDecodeFn.apply(self.separator(z, e), xbs, plan, wav.shape[1], *mask_params)
        if not torch.is_grad_enabled():
            return s, predict_speaker_lable
        self_e, self_predict_speaker_lable = self._speaker(s.detach())
        self_s = F_.MaskDecodeFn.apply(self.separator(z, self_e), xbs, plan, wav.shape[1], *mask_params)
        return s, self_s, predict_speaker_lable, self_predict_speaker_lable
