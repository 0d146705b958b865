"""Diagnostic (GPU): what makes gemm_b2p an aggressor for stft_bandsplit running on another stream?  Replays the
victim beside variations of the aggressor launch (time-view vs band-view arguments, operand contents, grid size)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import bsrnn_oracle as O  # noqa: E402
from wesep_amd import dev  # noqa: E402
from wesep_amd import functional as F_  # noqa: E402
from wesep_amd.dev import BIG, SeqMap  # noqa: E402
from wesep_amd.models import get_model  # noqa: E402

d = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
R, T = 2, 24000
kw = dict(num_repeat=1, spk_fuse_type="multiply", multi_fuse=False)
cfg = O.BSRNNConfig(**kw)
model = get_model("BSRNN")(use_spk_transform=False, joint_training=False, **kw)
model.load_state_dict(O.synth_params(cfg, 1))
model.to(d).eval()
wav, tgt, emb = (t.to(d) for t in O.synth_batch(R, T, 1))
plan = model._plan(d)
Tf, K, Nf, H = 1 + T // 128, 32, 128, 256
xbs = torch.empty(R * Tf, 514, device=d)
s0, s1 = torch.cuda.Stream(device=d), torch.cuda.Stream(device=d)


def victim():
    dev.stft_bandsplit(wav, plan.bands, xbs)


victim()
torch.cuda.synchronize()
ref = xbs.clone()
g = torch.Generator().manual_seed(3)
wpack = torch.empty(Nf * 2 * H, device=d)
dev.pack_w((0.05 * torch.randn(Nf, 2 * H, generator=g)).to(d), Nf, 2 * H, 2 * H, wpack, order=1)
z = torch.randn(R, K, Tf, Nf, generator=g).to(d)
out = torch.empty_like(z)
bias = torch.zeros(Nf, device=d)


def trial(tag, seq, A):
    bad = 0
    for _ in range(N):
        with torch.cuda.stream(s0):
            dev.gemm_b2p(A=A, K=2 * H, sm=seq, Wpack=wpack, C_out=out, ldc=Nf, bias=bias, R=z)
        with torch.cuda.stream(s1):
            victim()
        torch.cuda.synchronize()
        if not torch.equal(xbs, ref):
            bad += 1
            victim()
            torch.cuda.synchronize()
    print(f"{tag}: victim mismatches {bad} of {N}", flush=True)


time_seq = SeqMap(R * K, BIG, 0, Tf, 1, Tf)
band_seq = SeqMap(R * Tf, Tf, K * Tf, 1, Tf, K)
for tag, seq in (("time view (64 seq x 188 steps)", time_seq), ("band view (376 seq x 32 steps)", band_seq),
                 ("band-like, 384 seq (no padded slots)", SeqMap(384, BIG, 0, 32, 1, 32)),
                 ("time-like, 60 seq (padded slots)", SeqMap(60, BIG, 0, Tf, 1, Tf))):
    nb = dev.bl_num_blocks(seq)
    for atag, A in (("randn", torch.randn(nb, 32 * 2 * H, generator=g).to(d)), ("zeros", torch.zeros(nb, 32 * 2 * H, device=d))):
        if "384" in tag or "60 seq" in tag:
            zz = torch.randn(max(384 * 32, 60 * Tf), Nf, generator=g).to(d)
            o2 = torch.empty_like(zz)
            bad = 0
            for _ in range(N):
                with torch.cuda.stream(s0):
                    dev.gemm_b2p(A=A, K=2 * H, sm=seq, Wpack=wpack, C_out=o2, ldc=Nf, bias=bias, R=zz)
                with torch.cuda.stream(s1):
                    victim()
                torch.cuda.synchronize()
                if not torch.equal(xbs, ref):
                    bad += 1
                    victim()
                    torch.cuda.synchronize()
            print(f"{tag}, A {atag}: victim mismatches {bad} of {N}", flush=True)
        else:
            trial(f"{tag}, A {atag}", seq, A)
