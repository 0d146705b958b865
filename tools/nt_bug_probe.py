"""GPU probe: characterise the run-to-run mismatches of gemm_nt (mode bf16x3)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wesep_amd import dev  # noqa: E402

d = torch.device("cuda:0")
Tf = 501
torch.manual_seed(0)


def run_case(M, N, Kd, norm, reps=6):
    A = torch.randn(M, Kd, device=d)
    W = torch.randn(N, Kd, device=d) * 0.05
    b = torch.randn(N, device=d)
    kw = {}
    if norm:
        ng = -(-M // Tf)
        stats = torch.stack([torch.randn(ng, device=d) * 0.1, torch.rand(ng, device=d) + 0.5], 1).contiguous()
        kw = dict(stats=stats, gamma=torch.randn(Kd, device=d), beta=torch.randn(Kd, device=d),
                  stat_map=dev.StatMap(Tf, 1, 1, 0, 0))
    ref = torch.empty(M, N, device=d)
    dev.gemm_nt(A=A, a_rows=dev.flat(Kd), M=M, N=N, K=Kd, W=W, ldw=Kd, bias=b, C_out=ref, c_rows=dev.flat(N),
                mode="f32", **kw)
    for r in range(reps):
        C = torch.empty(M, N, device=d)
        dev.gemm_nt(A=A, a_rows=dev.flat(Kd), M=M, N=N, K=Kd, W=W, ldw=Kd, bias=b, C_out=C, c_rows=dev.flat(N),
                    mode="bf16x3", **kw)
        err = (C - ref).abs()
        bad = (err > 1e-2 * ref.abs().max()).nonzero()
        msg = ""
        if bad.shape[0]:
            rows, cols = bad[:, 0], bad[:, 1]
            msg = (f" BAD n={bad.shape[0]} rows[{int(rows.min())}..{int(rows.max())}] (mod128 {int(rows.min()) % 128}..{int(rows.max()) % 128},"
                   f" distinct {rows.unique().numel()}) cols[{int(cols.min())}..{int(cols.max())}] distinct {cols.unique().numel()}"
                   f" maxerr {float(err.max()):.3e}")
        print(f"M={M} N={N} K={Kd} norm={norm} rep{r}: maxerr {float(err.max()):.3e}{msg}", flush=True)
        del C


run_case(513024, 2048, 128, True)
run_case(513024, 2048, 128, False)
run_case(513024, 512, 128, True)
run_case(64128, 2048, 128, True)
