"""GPU probe: the time-view BPTT kernel (16-sequence workgroups) alone and beside a second stream
that keeps the rest of the chip busy (HBM copies or the blocked weight-gradient GEMM).  Not part of
the product."""
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
mode = int(os.environ.get("PROBE_MODE", 5))
dev.lstm_pack(whf, whr, pf, pb, mode)
dev.lstm_fwd(gates, cbuf, hcat, pf, seq, mode)
src, dst = torch.randn(1 << 28, device=d), torch.empty(1 << 28, device=d)  # 1 GiB each
side = torch.cuda.Stream()


def bptt():
    dev.lstm_bwd(gates, cbuf, hcat, dh, pb, seq, mode)


def timed(background):
    """One recurrence launch; the background launches are released right after it (as the step does)."""
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    bptt()
    if background:
        with torch.cuda.stream(side):
            for _ in range(4):
                background()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b)


Gb = torch.randn(nb, 32 * 2048, device=d)
xn = torch.randn(nb, 32 * 128, device=d)
ns, bps = dev.tnb_splits(nb, 8)
slab, bslab = torch.empty(ns, 1024 * 384, device=d), torch.empty(ns, 1024, device=d)


def tnb():
    dev.gemm_tnb(G=Gb, g_width=2048, g_off=0, g_cols=1024, A0=xn, a0_width=128, a0_off=0, a0_cols=128,
                 A1=hcat, a1_width=512, a1_off=0, a1_cols=256, a1_shift=-1, nblk=nb, L_=seq.L,
                 slab=slab, nsplit=ns, blocks_per_split=bps, bslab=bslab)


bptt()
tnb()
print(f"alone            {timed(None):7.3f} ms")
print(f"beside HBM copy  {timed(lambda: dst.copy_(src)):7.3f} ms")
print(f"beside dW GEMM   {timed(tnb):7.3f} ms")
