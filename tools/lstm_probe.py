"""GPU probe: time the split-bf16 LSTM recurrence kernels (and their DBG variants) at the
headline size, both views.  Not part of the product; prints one line per configuration."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wesep_amd import dev, _lib as L  # noqa: E402
from wesep_amd.functional import _view_maps  # noqa: E402

d = torch.device("cuda:0")
R, K, Tf, N, H = int(os.environ.get("PROBE_R", 32)), 32, 501, 128, 256
P = R * K * Tf
g = torch.Generator(device="cpu").manual_seed(0)
whf = (torch.randn(4 * H, H, generator=g) * 0.06).to(d)
whr = (torch.randn(4 * H, H, generator=g) * 0.06).to(d)
pf, pb = torch.empty(L.LSTM_PACK_FLOATS, device=d), torch.empty(L.LSTM_PACK_FLOATS, device=d)
gates0 = torch.randn(P, 2, 4 * H, device=d)
gates = torch.empty_like(gates0)
cbuf, hcat = torch.zeros(P, 2 * H, device=d), torch.zeros(P, 2 * H, device=d)
dh = torch.randn(P, 2 * H, device=d) * 1e-3


def timeit(fn, n=2):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


modes = [int(m) for m in os.environ.get("PROBE_MODES", "4,5").split(",")]
for view in ("time", "band"):
    _, _, seq, _ = _view_maps(view, R, K, Tf, N)
    for mode in modes:
        base = mode & 255
        dev.lstm_pack(whf, whr, pf, pb, base)
        gates.copy_(gates0)
        tf = timeit(lambda: dev.lstm_fwd(gates, cbuf, hcat, pf, seq, mode))
        gates.copy_(gates0)
        dev.lstm_fwd(gates, cbuf, hcat, pf, seq, base)
        tb = timeit(lambda: dev.lstm_bwd(gates, cbuf, hcat, dh, pb, seq, mode))
        if base == 4 and dev.lstm_cluster_ok(seq, d):
            for dbg in (0, 1):
                gates.copy_(gates0)
                tc = timeit(lambda: dev.lstm_fwd_cluster(gates, cbuf, hcat, whf, whr, seq, dbg=dbg))
                print(f"{view} CLUSTER fwd dbg={dbg} {tc:8.3f} ms ({tc * 1e3 / seq.L:6.2f} us/step)", flush=True)
        if base == 4 and dev.lstm_cluster_ok(seq, d):
            for dbg in (0, 1):
                gates.copy_(gates0)
                dev.lstm_fwd_cluster(gates, cbuf, hcat, whf, whr, seq)
                tc = timeit(lambda: dev.lstm_bwd_cluster(gates, cbuf, dh, whf, whr, seq, dbg=dbg))
                print(f"{view} CLUSTER bwd dbg={dbg} {tc:8.3f} ms ({tc * 1e3 / seq.L:6.2f} us/step)", flush=True)
        steps = seq.L * (1 if view == "time" else -(-seq.nseq // 32 * 2) // 256)
        print(f"{view} mode={base} dbg={mode >> 8} fwd {tf:8.3f} ms  bwd {tb:8.3f} ms  "
              f"(nseq {seq.nseq}, L {seq.L}; per step-slot fwd {tf * 1e3 / steps:6.2f} us, bwd {tb * 1e3 / steps:6.2f} us)",
              flush=True)
