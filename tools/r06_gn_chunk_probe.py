"""Round 6 probe: would the time view's GroupNorm backward (ws_gn_bwd_reduce + ws_gn_bwd_apply_pg: x and dxn cross HBM twice) gain
from doing both passes over a CHUNK of groups before moving on, so that the second pass finds x / dxn in the 256 MB
infinity cache?  Times the existing two kernels over the whole tensor against the same kernels launched chunk by chunk
(128 MB, 64 MB, 32 MB, 16 MB of x + dxn per chunk) at the headline geometry (1024 groups of 501 x 128).  The chunked form is
a measurement, not a product path: its parameter sums are per chunk.

    python tools/r06_gn_chunk_probe.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from wesep_amd import dev
    d = torch.device("cuda:0")
    G, Tf, N = 1024, 501, 128
    g = torch.Generator().manual_seed(0)
    x = torch.randn(G, Tf, N, generator=g).to(d)
    dxn = (1e-3 * torch.randn(G, Tf, N, generator=g)).to(d)
    res = (1e-3 * torch.randn(G, Tf, N, generator=g)).to(d)
    gamma = (1.0 + 0.1 * torch.randn(N, generator=g)).to(d)
    dx = torch.empty_like(x)
    geo = dev.Geom(G, 1, Tf * N, 0, N, Tf, N)
    stats = torch.empty(G, 2, device=d)
    dev.group_stats(x, geo, stats)
    ab = torch.empty(G, 2, device=d)

    def run(chunk):
        for g0 in range(0, G, chunk):
            n = min(chunk, G - g0)
            cg = dev.Geom(n, 1, Tf * N, 0, N, Tf, N)
            dev.gn_bwd_reduce(x[g0:g0 + n], dxn[g0:g0 + n], stats[g0:g0 + n], cg, ab[g0:g0 + n], gamma=gamma)
            pslab = torch.empty(n + dev.tree_groups(n), 2, N, device=d)
            pout = torch.empty(2, N, device=d)
            cnt = torch.zeros(1 + dev.tree_groups(n), device=d, dtype=torch.int32)
            dev.gn_bwd_apply_pg(x[g0:g0 + n], dxn[g0:g0 + n], stats[g0:g0 + n], ab[g0:g0 + n], cg, dx[g0:g0 + n], gamma, pslab,
                                pout, cnt, res=res[g0:g0 + n])

    ref = None
    for chunk in (1024, 512, 256, 128, 64, 32):
        for _ in range(3):
            run(chunk)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run(chunk)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        if ref is None:
            ref = dx.clone()
        print(f"chunk {chunk:5d} groups ({chunk * Tf * N * 8 / 1e6:7.1f} MB of x + dxn): {ms * 1e3:8.1f} us per backward "
              f"({G // chunk * 2} launches + {G // chunk * 3} allocator / fill calls); dx identical: {torch.equal(dx, ref)}", flush=True)


if __name__ == "__main__":
    main()
