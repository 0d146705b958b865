"""GPU probe (round 5): the band-view BPTT alone at the headline geometry (R = 32 rows x 501 frames = 16 032 sequences x 32
bands): launch time of the streaming kernel with the three-term split-bf16 product and with rfmt 2 (stored fp16 d(gates) x
fp16 hi + scaled-FP8 lo W_hh), and the distance of the two results.  Not part of the product.

    python tools/r05_band_probe.py [--rows 32]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wesep_amd import _lib as L, dev  # noqa: E402
from wesep_amd.functional import _view_maps  # noqa: E402
from r05_recur_probe import timeit  # noqa: E402

H, N, K = 256, 128, 32


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=32)
    ap.add_argument("--frames", type=int, default=501)
    a = ap.parse_args()
    d = torch.device("cuda:0")
    g = torch.Generator().manual_seed(7)
    R, Tf = a.rows, a.frames
    whf = (0.06 * torch.randn(4 * H, H, generator=g)).to(d)
    whr = (0.06 * torch.randn(4 * H, H, generator=g)).to(d)
    _, _, seq, _ = _view_maps("band", R, K, Tf, N)
    nb = dev.bl_num_blocks(seq)
    mode = L.LSTM_BF16X3_BLK
    pre = torch.randn(nb, 32 * 8 * H, device=d)
    cbuf = torch.zeros(nb, 2 * H // 4, 32, 4, device=d)
    hcat = torch.zeros_like(cbuf)
    dh = torch.randn(nb, 2 * H // 4, 32, 4, device=d) * 1e-3
    amax = dh.abs().max().reshape(1).view(torch.int32).clone()
    gh = torch.zeros(dev.blh_floats(nb, 8 * H), device=d)
    pf, pb, pb8 = (torch.zeros(L.LSTM_PACK_FLOATS, device=d) for _ in range(3))
    dev.lstm_pack(whf, whr, pf, pb, mode)
    dev.lstm_pack_bwd_f8(whf, whr, pb8)
    dev.lstm_fwd(gh, cbuf, hcat, pf, seq, mode, gfmt=L.GATES_H2, gates_in=pre)
    out = {}
    for rf, pk, nm in ((0, pb, "bf16x3 (three terms, 128 KB / wave / step)"), (2, pb8, "fp16 + FP8 lo (two terms, 96 KB)")):
        dgo = torch.zeros_like(gh)
        t = timeit(lambda: dev.lstm_bwd(gh, cbuf, hcat, dh, pk, seq, mode, gfmt=L.GATES_H2F, dgates=dgo, amax=amax, rfmt=rf), n=5)
        out[rf] = dgo.view(torch.float16)[: nb * 32 * 8 * H].float()
        print(f"band BPTT, unorm16 in / fp16 out, {nm:44s} {t:7.3f} ms  {t * 1e3 / K:6.2f} us/step", flush=True)
    e = float((out[2] - out[0]).norm() / out[0].norm())
    print(f"rfmt 2 vs 0: d(gates) rel-L2 {e:.2e}, finite {bool(torch.isfinite(out[2]).all())}")


if __name__ == "__main__":
    main()
