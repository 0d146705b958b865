"""Diagnostic (GPU): are library kernels affected too?  (a) victim torch.fft.rfft / torch.matmul beside our gemm_b2p;
(b) our stft_bandsplit beside torch.matmul (rocBLAS / hipBLASLt) and beside our other kernels."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import bsrnn_oracle as O  # noqa: E402
from wesep_amd import dev  # noqa: E402
from wesep_amd.dev import BIG, SeqMap  # noqa: E402
from wesep_amd.models import get_model  # noqa: E402

d = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
R, T = 2, 24000
kw = dict(num_repeat=1, spk_fuse_type="multiply", multi_fuse=False)
model = get_model("BSRNN")(use_spk_transform=False, joint_training=False, **kw).to(d).eval()
wav, tgt, emb = (t.to(d) for t in O.synth_batch(R, T, 1))
plan = model._plan(d)
Tf, K, Nf, H = 1 + T // 128, 32, 128, 256
g = torch.Generator().manual_seed(3)
s0, s1 = torch.cuda.Stream(device=d), torch.cuda.Stream(device=d)
wpack = torch.empty(Nf * 2 * H, device=d)
dev.pack_w((0.05 * torch.randn(Nf, 2 * H, generator=g)).to(d), Nf, 2 * H, 2 * H, wpack, order=1)
seq = SeqMap(R * Tf, Tf, K * Tf, 1, Tf, K)
A = torch.randn(dev.bl_num_blocks(seq), 32 * 2 * H, generator=g).to(d)
z = torch.randn(R, K, Tf, Nf, generator=g).to(d)
out = torch.empty_like(z)
bias = torch.zeros(Nf, device=d)
xbs = torch.empty(R * Tf, 514, device=d)
sig = torch.randn(4096, 512, generator=g).to(d)
ma, mb = torch.randn(2048, 384, generator=g).to(d), torch.randn(384, 1024, generator=g).to(d)
big = torch.randn(4096, 4096, generator=g).to(d)


def b2p():
    dev.gemm_b2p(A=A, K=2 * H, sm=seq, Wpack=wpack, C_out=out, ldc=Nf, bias=bias, R=z)


def stft():
    dev.stft_bandsplit(wav, plan.bands, xbs)
    return xbs


def race(tag, aggr, vict):
    ref = vict().clone()
    torch.cuda.synchronize()
    bad = 0
    for _ in range(N):
        with torch.cuda.stream(s0):
            aggr()
        with torch.cuda.stream(s1):
            o = vict()
        torch.cuda.synchronize()
        bad += int(not torch.equal(o, ref))
    print(f"{tag}: victim mismatches {bad} of {N}", flush=True)


idx = torch.randperm(4096, generator=g).to(d)
race("transpose copy (4-byte strided loads) beside gemm_b2p", b2p, lambda: sig.t().contiguous())
race("strided slice copy beside gemm_b2p", b2p, lambda: sig[:, ::3].contiguous())
race("index_select rows beside gemm_b2p", b2p, lambda: sig.index_select(0, idx))
race("cumsum beside gemm_b2p", b2p, lambda: torch.cumsum(sig, 1))
race("torch.sin beside gemm_b2p", b2p, lambda: torch.sin(sig))
race("sort beside gemm_b2p", b2p, lambda: torch.sort(sig, 1)[0])
race("torch.fft.rfft beside gemm_b2p", b2p, lambda: torch.view_as_real(torch.fft.rfft(sig)))
race("torch.fft.fft (complex) beside gemm_b2p", b2p, lambda: torch.view_as_real(torch.fft.fft(torch.complex(sig, sig))))
race("layer_norm beside gemm_b2p", b2p, lambda: torch.nn.functional.layer_norm(sig, (512,)))
