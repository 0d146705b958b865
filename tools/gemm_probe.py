"""GPU probe: determinism + timing of the GEMM kernels at the model's shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wesep_amd import dev  # noqa: E402

d = torch.device("cuda:0")
R, K, Tf = int(os.environ.get("PROBE_R", 32)), 32, 501
P = R * K * Tf
torch.manual_seed(0)


def timeit(fn, n=3):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def probe_nt(name, M, N, Kd, norm=False, res=False):
    A = torch.randn(M, Kd, device=d)
    W = torch.randn(N, Kd, device=d) * 0.05
    b = torch.randn(N, device=d)
    Rr = torch.randn(M, N, device=d) if res else None
    kw = {}
    if norm:
        stats = torch.stack([torch.randn(M // Tf, device=d) * 0.1, torch.rand(M // Tf, device=d) + 0.5], 1).contiguous()
        kw = dict(stats=stats, gamma=torch.randn(Kd, device=d), beta=torch.randn(Kd, device=d),
                  stat_map=dev.StatMap(Tf, 1, 1, 0, 0))
    for mode in ("f32", "bf16x3"):
        C1, C2 = torch.empty(M, N, device=d), torch.empty(M, N, device=d)
        run = lambda C: dev.gemm_nt(A=A, a_rows=dev.flat(Kd), M=M, N=N, K=Kd, W=W, ldw=Kd, bias=b, C_out=C,
                                    c_rows=dev.flat(N), R=Rr, mode=mode, **kw)
        run(C1)
        run(C2)
        same = torch.equal(C1, C2)
        msg = ""
        if not same:
            bad = (C1 != C2).nonzero()
            msg = f" MISMATCH n={bad.shape[0]} first={bad[:4].tolist()} maxdiff={float((C1 - C2).abs().max()):.3e}"
        t = timeit(lambda: run(C1))
        gb = (A.numel() + C1.numel() + (Rr.numel() if res else 0)) * 4 / 1e9
        print(f"NT {name:8s} {mode:7s} M={M} N={N} K={Kd}: {t:7.3f} ms  {2.0 * M * N * Kd / t / 1e9:8.1f} TFLOP/s(alg)  "
              f"{gb / t * 1e3:7.1f} GB/s  deterministic={same}{msg}", flush=True)
        del C1, C2


probe_nt("xproj", P, 2048, 128, norm=True)
probe_nt("proj", P, 128, 512, res=True)
probe_nt("dhcat", P, 512, 128)
probe_nt("dxn", P, 128, 2048)

# ---- blocked-layout kernels (time view) -----------------------------------------------------
from wesep_amd.functional import _view_maps  # noqa: E402

for view in ("time", "band"):
    _, smap, seq, _ = _view_maps(view, R, K, Tf, 128)
    nb = dev.bl_num_blocks(seq)
    z = torch.randn(P, 128, device=d)
    for name, Nout in (("xproj", 2048), ("dhcat", 512)):
        W = torch.randn(Nout, 128, device=d) * 0.05
        wp = torch.empty(Nout * 128, device=d)
        dev.pack_w(W, Nout, 128, 128, wp, order=0)
        C1, Ab = torch.empty(nb, 32 * Nout, device=d), torch.empty(nb, 32 * 128, device=d)
        t = timeit(lambda: dev.gemm_p2b(A=z, lda=128, sm=seq, Wpack=wp, N=Nout, C_out=C1, A_bl=Ab))
        gb = (z.numel() + C1.numel() + Ab.numel()) * 4 / 1e9
        print(f"p2b {view} {name}: {t:7.3f} ms  {gb / t * 1e3:7.1f} GB/s", flush=True)
        del C1, Ab
    for name, Kd in (("proj", 512), ("dxn", 2048)):
        W = torch.randn(128, Kd, device=d) * 0.05
        wp = torch.empty(128 * Kd, device=d)
        dev.pack_w(W, 128, Kd, Kd, wp, order=1)
        Ab = torch.randn(nb, 32 * Kd, device=d)
        C1 = torch.empty(P, 128, device=d)
        t = timeit(lambda: dev.gemm_b2p(A=Ab, K=Kd, sm=seq, Wpack=wp, C_out=C1, ldc=128, R=z))
        gb = (Ab.numel() + 2 * C1.numel()) * 4 / 1e9
        print(f"b2p {view} {name}: {t:7.3f} ms  {gb / t * 1e3:7.1f} GB/s", flush=True)
        del Ab, C1
    G = torch.randn(nb, 32 * 2048, device=d)
    xn = torch.randn(nb, 32 * 128, device=d)
    h = torch.randn(nb, 32 * 512, device=d)
    ns, bps = dev.tnb_splits(nb, 8)
    slab, bslab = torch.empty(ns, 1024 * 384, device=d), torch.empty(ns, 1024, device=d)
    for dbg in (0,):
        t = timeit(lambda: dev.gemm_tnb(G=G, g_width=2048, g_off=0, g_cols=1024, A0=xn, a0_width=128, a0_off=0, a0_cols=128,
                                        A1=h, a1_width=512, a1_off=0, a1_cols=256, a1_shift=-1, nblk=nb, L_=seq.L,
                                        slab=slab, nsplit=ns, blocks_per_split=bps, bslab=bslab, dbg=dbg))
        gb = (G.numel() / 2 + xn.numel() + h.numel() / 2) * 4 / 1e9
        print(f"tnb {view} dW(dir) dbg={dbg}: {t:7.3f} ms  {gb / t * 1e3:7.1f} GB/s (unique bytes)  nsplit={ns}", flush=True)
    del G, xn, h, slab
