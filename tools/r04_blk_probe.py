"""Isolated timings of the blocked-layout GEMMs around the recurrence at the headline geometry (32 rows x 4 s), per storage
format of d(gates) (wesep_hip.h WS_GATES_*): ws_gemm_tnb (g_fmt 0 / 1 / 2; WS_TNB_GDEPTH for the depth of the G prefetch
ring), ws_gemm_b2p as d(xn) (a_fmt 0 / 1 / 2) and as the output projection, ws_gemm_p2b.  One MI355X, nothing else running.

    python tools/r04_blk_probe.py [--view time|band|both] [--reps 20]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timeit(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--view", default="both")
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    from wesep_amd import dev
    from wesep_amd.functional import _view_maps
    d = torch.device("cuda:0")
    R, K, Tf, N, H = 32, 32, 501, 128, 256
    P = R * K * Tf
    amax = torch.tensor([3e-6], dtype=torch.float32).view(torch.int32).to(d)
    for view in (("time", "band") if a.view == "both" else (a.view,)):
        _, smap, seq, _ = _view_maps(view, R, K, Tf, N)
        nb = dev.bl_num_blocks(seq)
        xn = dev.bls_pack(torch.randn(nb, N // 4, 32, 4, device=d))
        h = dev.bls_pack(torch.randn(nb, 2 * H // 4, 32, 4, device=d))
        g32 = torch.randn(nb, 2048 // 4, 32, 4, device=d)
        G = {0: dev.bls_pack(g32), 1: dev.blh_bf16_pack(g32), 2: (g32 * 512).to(torch.float16).contiguous().view(torch.float32)}
        del g32
        ns, bps = dev.tnb_splits(nb, 8)
        slab, bslab = torch.empty(ns, 1024 * 384, device=d), torch.empty(ns, 1024, device=d)
        for fmt, f16mm in ((0, "1"), (1, "1"), (2, "0"), (2, "1")):
            os.environ["WS_TNB_F16"] = f16mm      # g_fmt 2 on the bf16 instruction (3 terms) / the fp16 one (2 terms, default)
            t = timeit(lambda: dev.gemm_tnb(G=G[fmt], g_width=2048, g_off=0, g_cols=1024, A0=xn, a0_width=N, a0_off=0, a0_cols=N,
                                            A1=h, a1_width=2 * H, a1_off=0, a1_cols=H, a1_shift=-1, nblk=nb, L_=seq.L, slab=slab,
                                            nsplit=ns, blocks_per_split=bps, bslab=bslab, g_fmt=fmt,
                                            amax=amax if fmt == 2 else None), a.reps)
            gb = (nb * 32 * 1024 * (4 if fmt == 0 else 2) + nb * 32 * (N + H) * 4) / 1e9
            print(f"tnb  {view} g_fmt={fmt} f16_mfma={f16mm if fmt == 2 else '-'} gdepth={os.environ.get('WS_TNB_GDEPTH', '4')}: {t:7.3f} ms  {gb / t * 1e3:7.0f} GB/s unique  "
                  f"nsplit={ns}", flush=True)
        # a_fmt = 1 (ABI v16): fp16 copies of [xn | h] -- one MFMA per product, 32 instead of 56 KB loaded per block
        xn16 = dev.blh_f16_pack(torch.randn(nb, N // 4, 32, 4, device=d))
        h16 = dev.blh_f16_pack(torch.tanh(torch.randn(nb, 2 * H // 4, 32, 4, device=d)))
        t = timeit(lambda: dev.gemm_tnb(G=G[2], g_width=2048, g_off=0, g_cols=1024, A0=xn16, a0_width=N, a0_off=0, a0_cols=N,
                                        A1=h16, a1_width=2 * H, a1_off=0, a1_cols=H, a1_shift=-1, nblk=nb, L_=seq.L, slab=slab,
                                        nsplit=ns, blocks_per_split=bps, bslab=bslab, g_fmt=2, amax=amax, a_fmt=1), a.reps)
        gb = (nb * 32 * 1024 * 2 + nb * 32 * (N + H) * 2) / 1e9
        print(f"tnb  {view} g_fmt=2 a_fmt=1 (fp16 A): {t:7.3f} ms  {gb / t * 1e3:7.0f} GB/s unique  nsplit={ns}", flush=True)
        del xn16, h16
        W = torch.randn(N, 2048, device=d) * 0.05
        wp = torch.empty(N * 2048, device=d)
        dev.pack_w(W.t().contiguous(), N, 2048, N, wp, trans=True, order=1)
        C1 = torch.empty(P, N, device=d)
        for fmt in (0, 1, 2):
            t = timeit(lambda: dev.gemm_b2p(A=G[fmt], K=2048, sm=seq, Wpack=wp, C_out=C1, ldc=N, a_fmt=fmt,
                                            amax=amax if fmt == 2 else None), a.reps)
            gb = (nb * 32 * 2048 * (4 if fmt == 0 else 2) + P * N * 4) / 1e9
            print(f"b2p  {view} dxn a_fmt={fmt}: {t:7.3f} ms  {gb / t * 1e3:7.0f} GB/s", flush=True)
        del G, slab
        z = torch.randn(P, N, device=d)
        for name, Nout in (("xproj", 2048), ("dhcat", 512)):
            W = torch.randn(Nout, N, device=d) * 0.05
            wp = torch.empty(Nout * N, device=d)
            dev.pack_w(W, Nout, N, N, wp, order=0)
            C2, Ab = torch.empty(nb, 32 * Nout, device=d), torch.empty(nb, 32 * N, device=d)
            am = torch.zeros(1, device=d, dtype=torch.int32)
            for with_amax in (False, True):
                t = timeit(lambda: dev.gemm_p2b(A=z, lda=N, sm=seq, Wpack=wp, N=Nout, C_out=C2, A_bl=Ab,
                                                amax=am if with_amax else None), a.reps)
                gb = (z.numel() + C2.numel() + Ab.numel()) * 4 / 1e9
                print(f"p2b  {view} {name} amax={int(with_amax)}: {t:7.3f} ms  {gb / t * 1e3:7.0f} GB/s", flush=True)
            del C2, Ab
        del xn, h, z, C1


if __name__ == "__main__":
    main()
