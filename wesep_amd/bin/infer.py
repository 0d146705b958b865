"""Whole-utterance inference / scoring loop on the MI355X engine -- the part of wesep/bin/infer.py
that touches the model (infer.py:72-73, 108-128, 150-175): eval mode, no_grad, batch of the two
target speakers of one mixture, variable length, peak normalisation to 0.9, SI-SNR / SI-SNRi.
Dataset / config / wav-file plumbing stays in wesep (it is not on the device path)."""
import numpy as np
import torch

from ..utils.score import cal_SISNRi


@torch.no_grad()
def extract(model, wav_mix, enroll):
    """wav_mix [B, T], enroll [B, spk_emb_dim] (device tensors) -> numpy [B, T], peak-normalised like
    infer.py:118-128 (only when every row's maximum is positive, exactly as the reference)."""
    model.eval()
    outputs = model(wav_mix.float(), enroll.float())
    if isinstance(outputs, (list, tuple)):
        outputs = outputs[0]
    if torch.min(outputs.max(dim=1).values) > 0:
        outputs = outputs / outputs.abs().max(dim=1, keepdim=True)[0] * 0.9
    return outputs.cpu().numpy()


def extract_engine(engine, wav_mix, enroll, kind=None):
    """The same step on the native runtime (`wesep_amd.engine.Engine`, runtime/libwesep_engine.so): host arrays in,
    numpy [B, T] out, peak-normalised by the same rule.  `kind`: an `ENROLL_*` constant; default by rank
    (2-D embeddings for fixed-embedding models are ENROLL_EMBEDDING, 3-D is fbank; pass ENROLL_WAVE for audio)."""
    from .. import engine as E
    wav_mix = np.ascontiguousarray(wav_mix, dtype=np.float32)
    enroll = np.ascontiguousarray(enroll, dtype=np.float32)
    if kind is None:
        kind = E.ENROLL_FBANK if enroll.ndim == 3 else E.ENROLL_EMBEDDING
    outputs = engine.separate(wav_mix, enroll, kind)
    if outputs.max(axis=1).min() > 0:
        outputs = outputs / np.abs(outputs).max(axis=1, keepdims=True) * 0.9
    return outputs


def evaluate(model, batches, device="cuda"):
    """batches: iterable of dicts with `wav_mix` [B, T], `wav_targets` [B, T], `spk_embeds` [B, E]
    (what tse_collate_fn_2spk yields, infer.py:108-116).  Returns (mean SI-SNR, mean SI-SNRi, count):
    the accumulation of infer.py:150-175."""
    tot, toti, n = 0.0, 0.0, 0
    for b in batches:
        mix = torch.as_tensor(b["wav_mix"]).float().to(device)
        ref = np.asarray(b["wav_targets"], dtype=np.float32)
        est = extract(model, mix, torch.as_tensor(b["spk_embeds"]).float().to(device))
        mixn = mix.cpu().numpy()
        for r in range(est.shape[0]):
            end = min(len(est[r]), len(ref[r]))
            s, si = cal_SISNRi(est[r][:end], ref[r][:end], mixn[r][:end])
            tot, toti, n = tot + s, toti + si, n + 1
    return tot / max(n, 1), toti / max(n, 1), n
