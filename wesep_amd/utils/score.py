"""Evaluation metrics of the inference loop (wesep/utils/score.py:7-36), host-side numpy exactly as
in the reference: they run on waveforms already copied back for writing / scoring."""
import numpy as np


def cal_SISNR(est, ref, eps=1e-8):
    """Scale-invariant SNR in dB of `est` against `ref` (1-D arrays of equal length);
    wesep/utils/score.py:7-21 -- note the reference's placement of eps (inside the log and in both
    denominators), which differs from the training loss."""
    est, ref = np.asarray(est), np.asarray(ref)
    assert len(est) == len(ref)
    est_zm = est - np.mean(est)
    ref_zm = ref - np.mean(ref)
    t = np.sum(est_zm * ref_zm) * ref_zm / (np.linalg.norm(ref_zm) ** 2 + eps)
    return 20 * np.log10(eps + np.linalg.norm(t) / (np.linalg.norm(est_zm - t) + eps))


def cal_SISNRi(est, ref, mix, eps=1e-8):
    """(SI-SNR of est, improvement over the unprocessed mixture); wesep/utils/score.py:24-36."""
    assert len(est) == len(ref) == len(mix)
    s1 = cal_SISNR(est, ref, eps)
    return s1, s1 - cal_SISNR(mix, ref, eps)
