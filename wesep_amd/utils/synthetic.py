"""Synthetic 2-speaker training rows of BASELINE.md section 3 (the shape of what
`tse_collate_fn` hands the executor, wesep/dataset/dataset.py:217-227)."""
import torch


def synth_batch(R: int, T: int, seed: int, emb_dim: int = 256):
    """s1, s2 ~ 0.1*N(0,1); mix = s1+s2 peak-normalised to <= 1; rows interleaved (mix,s1),(mix,s2);
    enrollment embedding N(0,1) [R, emb_dim].  Returns (wav_mix [R,T], wav_targets [R,T], emb)."""
    assert R % 2 == 0
    g = torch.Generator().manual_seed(seed)
    s = 0.1 * torch.randn(R // 2, 2, T, generator=g)
    mix = s.sum(1)
    peak = mix.abs().amax(-1, keepdim=True).clamp_min(1.0)
    mix, s = mix / peak, s / peak[:, None]
    wav_mix = mix[:, None, :].expand(R // 2, 2, T).reshape(R, T).contiguous()
    return wav_mix.float(), s.reshape(R, T).contiguous().float(), torch.randn(R, emb_dim, generator=g).float()
