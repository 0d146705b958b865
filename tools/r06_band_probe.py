"""GPU probe (round 6): the band-view recurrences ALONE at the headline geometry (R = 32 rows x 501 frames = 16 032 sequences x 32
bands) -- launch times of
  * the streaming BPTT: three-term product, rfmt 2, rfmt 2 + d(xn) inside (ws_lstm_args.dxn), and ws_gemm_b2p(a_fmt 2) over the
    same d(gates) (the launch the last variant replaces);
  * the fused forward: three-term (hfmt 0), fp16 h (hfmt 1) and fp16 h with the lo term on the FP8 MFMA (hfmt 5, ABI v20), each with the end-of-step wait of rounds 3-5 (vmcnt(0):
    WESEP_FUSED_DRAIN=1) and with the round-6 wait (vmcnt(48): only the DMA of the next step's input).
Not part of the product.

    python tools/r06_band_probe.py [--rows 32]
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
    P = R * K * Tf
    whf = (0.06 * torch.randn(4 * H, H, generator=g)).to(d)
    whr = (0.06 * torch.randn(4 * H, H, generator=g)).to(d)
    wif = (0.08 * torch.randn(4 * H, N, generator=g)).to(d)
    wir = (0.08 * torch.randn(4 * H, N, generator=g)).to(d)
    bias = (0.1 * torch.randn(2 * 4 * H, generator=g)).to(d)
    _, _, seq, _ = _view_maps("band", R, K, Tf, N)
    nb = dev.bl_num_blocks(seq)
    mode = L.LSTM_BF16X3_BLK
    cbuf = torch.zeros(nb, 2 * H // 4, 32, 4, device=d)
    hcat = torch.zeros_like(cbuf)
    gh = torch.zeros(dev.blh_floats(nb, 8 * H), device=d)
    xn = dev.bls_pack(torch.randn(nb, N // 4, 32, 4, device=d))
    # ---- forward ---------------------------------------------------------------------------------------------------
    for hf, nm in ((0, "three terms (bf16x3)"), (1, "fp16 h, two recurrent terms"), (5, "fp16 h, lo term on the FP8 MFMA")):
        fp = torch.empty(L.LSTM_FUSED_PACK_FLOATS, device=d)
        dev.lstm_pack_fused(wif, wir, whf, whr, fp, hfmt=hf)
        for drain in ("1", "0"):
            os.environ["WESEP_FUSED_DRAIN"] = drain
            t = timeit(lambda: dev.lstm_fwd_fused(gh, cbuf, hcat, xn, fp, bias, seq, gfmt=L.GATES_H2F, hfmt=hf), n=5)
            print(f"fused band forward, {nm:30s} end-of-step wait {'vmcnt(0)' if drain == '1' else 'vmcnt(48)'}: {t:7.3f} ms  "
                  f"{t * 1e3 / K:6.2f} us/step-launch", flush=True)
    os.environ["WESEP_FUSED_DRAIN"] = "0"
    # ---- BPTT ------------------------------------------------------------------------------------------------------
    dh = torch.randn(nb, 2 * H // 4, 32, 4, device=d) * 1e-3
    amax = dh.abs().max().reshape(1).view(torch.int32).clone()
    pf, pb, pb8 = (torch.zeros(L.LSTM_PACK_FLOATS, device=d) for _ in range(3))
    dev.lstm_pack(whf, whr, pf, pb, mode)
    dev.lstm_pack_bwd_f8(whf, whr, pb8)
    wcat = torch.cat([wif, wir]).contiguous()
    px = torch.zeros(L.LSTM_DX_PACK_FLOATS, device=d)
    dev.lstm_pack_dx_f8(wcat, px)
    dxn = torch.zeros(2, P, N, device=d)
    dgo = torch.zeros_like(gh)
    for nm, kw in (("bf16x3 (three terms, 128 KB / wave / step)", dict(wp=pb, rfmt=0)),
                   ("fp16 + FP8 lo (two terms, 96 KB)", dict(wp=pb8, rfmt=2)),
                   ("fp16 + FP8 lo term on the FP8 MFMA (rfmt 3)", dict(wp=pb8, rfmt=3)),
                   ("fp16 + FP8 lo + d(xn) inside (144 KB)", dict(wp=pb8, rfmt=2, dxn=dxn, wxpack=px))):
        wp = kw.pop("wp")
        t = timeit(lambda: dev.lstm_bwd(gh, cbuf, hcat, dh, wp, seq, mode, gfmt=L.GATES_H2F, dgates=dgo, amax=amax, **kw), n=5)
        print(f"band BPTT, unorm16 in / fp16 out, {nm:44s} {t:7.3f} ms  {t * 1e3 / K:6.2f} us/step-launch", flush=True)
    wt16 = torch.empty(N * 2 * 4 * H, device=d)
    dev.pack_w(wcat, N, 2 * 4 * H, N, wt16, trans=True, order=1, f16=True)
    c = torch.zeros(P, N, device=d)
    t = timeit(lambda: dev.gemm_b2p(A=dgo, K=2 * 4 * H, sm=seq, Wpack=wt16, C_out=c, ldc=N, a_fmt=2, amax=amax), n=5)
    print(f"ws_gemm_b2p(a_fmt 2) over the same d(gates) (what d(xn) inside replaces): {t:7.3f} ms")
    wt8 = torch.empty(N * 2 * 4 * H, device=d)
    dev.pack_w(wcat, N, 2 * 4 * H, N, wt8, trans=True, order=1, f16=2)
    c3 = torch.zeros(P, N, device=d)
    t = timeit(lambda: dev.gemm_b2p(A=dgo, K=2 * 4 * H, sm=seq, Wpack=wt8, C_out=c3, ldc=N, a_fmt=3, amax=amax), n=5)
    print(f"ws_gemm_b2p(a_fmt 3: the lo term on the FP8 MFMA):                        {t:7.3f} ms; vs a_fmt 2 rel-L2 "
          f"{float((c3 - c).norm() / c.norm()):.2e}")
    e = float(((dxn[0] + dxn[1]) - c).norm() / c.norm())
    print(f"d(xn) inside vs the GEMM: rel-L2 {e:.2e}")


if __name__ == "__main__":
    main()
