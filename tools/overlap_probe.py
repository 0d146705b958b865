"""GPU probe: the time-view BPTT (16-sequence workgroups, 128 CUs) beside the weight-gradient GEMMs of the side
stream -- unrestricted, or on a CU-masked stream (hipExtStreamCreateWithCUMask; KFD hands mask bit i to XCD i % 8,
so the low 8k bits are k CUs on every XCD).  Prints isolated times, the BPTT's slow-down and the make-span of
"one BPTT + the weight gradients of two ResRNNs + the next main-stream GEMM".  Not part of the product."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wesep_amd import dev, _lib as L  # noqa: E402
from wesep_amd.functional import _view_maps  # noqa: E402

d = torch.device("cuda:0")
R, K, Tf, N, H = 32, 32, 501, 128, 256
P = R * K * Tf
g = torch.Generator(device="cpu").manual_seed(0)
whf = (torch.randn(4 * H, H, generator=g) * 0.06).to(d)
whr = (torch.randn(4 * H, H, generator=g) * 0.06).to(d)
pf, pb = torch.empty(L.LSTM_PACK_FLOATS, device=d), torch.empty(L.LSTM_PACK_FLOATS, device=d)
_, _, seq, _ = _view_maps("time", R, K, Tf, N)
nb = dev.bl_num_blocks(seq)
gates = torch.randn(nb * 32, 2, 4 * H, device=d)
cbuf, hcat = torch.zeros(nb * 32, 2 * H, device=d), torch.zeros(nb * 32, 2 * H, device=d)
dh = torch.randn(nb * 32, 2 * H, device=d) * 1e-3
mode = 5
dev.lstm_pack(whf, whr, pf, pb, mode)
dev.lstm_fwd(gates, cbuf, hcat, pf, seq, mode)
Gb = dev.bls_pack(torch.randn(nb, 32 * 2048, device=d))
xn = dev.bls_pack(torch.randn(nb, 32 * 128, device=d))
ns, bps = dev.tnb_splits(nb, 8)
slab, bslab = torch.empty(ns, 1024 * 384, device=d), torch.empty(ns, 1024, device=d)
W = torch.randn(128, 2048, device=d) * 0.05
wp = torch.empty(128 * 2048, device=d)
dev.pack_w(W, 128, 2048, 2048, wp, order=1)
dxn = torch.empty(P, 128, device=d)


def bptt():
    dev.lstm_bwd(gates, cbuf, hcat, dh, pb, seq, mode)


TNB_DBG = 0


def tnb():
    dev.gemm_tnb(G=Gb, g_width=2048, g_off=0, g_cols=1024, A0=xn, a0_width=128, a0_off=0, a0_cols=128,
                 A1=hcat, a1_width=512, a1_off=0, a1_cols=256, a1_shift=-1, nblk=nb, L_=seq.L,
                 slab=slab, nsplit=ns, blocks_per_split=bps, bslab=bslab)


def b2p():
    dev.gemm_b2p(A=Gb, K=2048, sm=seq, Wpack=wp, C_out=dxn, ldc=128)


def masked_stream(lo, hi):
    """Stream restricted to the CUs whose mask bits are [lo, hi)."""
    hip = ctypes.CDLL("libamdhip64.so")
    words = (ctypes.c_uint32 * 8)()
    for i in range(lo, hi):
        words[i // 32] |= 1 << (i % 32)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


def ev():
    return torch.cuda.Event(enable_timing=True)


def alone(fn, n=3):
    fn()
    torch.cuda.synchronize()
    a, b = ev(), ev()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


print(f"isolated: bptt {alone(bptt):.3f} ms   tnb<3> {alone(tnb):.3f} ms   b2p dxn {alone(b2p):.3f} ms", flush=True)


def scenario(side, nside=4):
    """main: bptt, b2p   side (after the bptt is launched): nside x tnb.  Times relative to the start:
    bptt end, b2p end, end of every tnb."""
    torch.cuda.synchronize()
    e0, e1, e2 = ev(), ev(), ev()
    es = [ev() for _ in range(nside)]
    e0.record()
    bptt()
    e1.record()
    if side is not None:
        side.wait_event(e0)
        with torch.cuda.stream(side):
            for k in range(nside):
                tnb()
                es[k].record(side)
    b2p()
    e2.record()
    torch.cuda.synchronize()
    return [e0.elapsed_time(e1), e0.elapsed_time(e2)] + ([e0.elapsed_time(e) for e in es] if side is not None else [])


for TNB_DBG, label in ((1, "all cacheable (round 1)"), (0, "G nt"), (2, "G nt + A nt"), (3, "A nt only")):
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        t_side = alone(tnb)
    scenario(side)
    r = [scenario(side) for _ in range(3)]
    best = min(r, key=lambda x: max(x))
    print(f"{label:24s}: tnb alone {t_side:6.3f} ms | bptt end {best[0]:6.3f}  b2p end {best[1]:6.3f}  tnb ends "
          + " ".join(f"{x:6.3f}" for x in best[2:]), flush=True)
