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
