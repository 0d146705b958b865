"""GPU probe (round 5): the time-view recurrences alone at the headline geometry (R = 32 rows x 32 bands = 1024 sequences x
501 steps): launch times of the pair BPTT in its storage / arithmetic variants and of the forward cluster kernels, plus the
in-kernel cycle stamps (dbg 2048) of the pair BPTT per phase.  Not part of the product.

    python tools/r05_recur_probe.py [--rows 32] [--no-stamps] > profiles/r05_recurrence_step_budget.txt
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wesep_amd import _lib as L, dev  # noqa: E402
from wesep_amd.functional import _view_maps  # noqa: E402

H, N, K = 256, 128, 32


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=32)
    ap.add_argument("--frames", type=int, default=501)
    ap.add_argument("--no-stamps", action="store_true")
    a = ap.parse_args()
    d = torch.device("cuda:0")
    g = torch.Generator().manual_seed(7)
    R, Tf = a.rows, a.frames
    whf = (0.06 * torch.randn(4 * H, H, generator=g)).to(d)
    whr = (0.06 * torch.randn(4 * H, H, generator=g)).to(d)
    _, _, seq, _ = _view_maps("time", R, K, Tf, N)
    nb = dev.bl_num_blocks(seq)
    pre = torch.randn(nb, 32 * 8 * H, device=d)
    cbuf = torch.zeros(nb, 2 * H // 4, 32, 4, device=d)
    hcat = torch.zeros_like(cbuf)
    dh = torch.randn(nb, 2 * H // 4, 32, 4, device=d) * 1e-3
    amax = dh.abs().max().reshape(1).view(torch.int32).clone()
    gh = torch.zeros(dev.blh_floats(nb, 8 * H), device=d)
    st = torch.zeros(1, device=d, dtype=torch.int32)
    # ---- forward (cluster kernels) ---------------------------------------------------------------------------------
    g32 = pre.clone()
    t = timeit(lambda: dev.lstm_fwd_cluster(g32, cbuf, hcat, whf, whr, seq, status=st))
    print(f"cluster fwd, fp32 gates (in place)          {t:7.3f} ms  {t * 1e3 / Tf:6.2f} us/step", flush=True)
    t = timeit(lambda: dev.lstm_fwd_cluster(gh, cbuf, hcat, whf, whr, seq, status=st, gfmt=L.GATES_H2, gates_in=pre))
    print(f"cluster fwd, unorm16 gates (fp32 pre-act in) {t:7.3f} ms  {t * 1e3 / Tf:6.2f} us/step", flush=True)
    # second generation: fp16 h, x-projection fused in (from the fp16 normalised input), data-tagged hand-off
    xn16 = dev.bls_pack(torch.randn(nb, N // 4, 32, 4, device=d))       # (BLS pairs in BL(128))
    wcat, bcat = torch.randn(2 * 4 * H * N, device=d) * 0.08, torch.randn(2 * 4 * H, device=d) * 0.1
    for dbg, rf, nm in ((0, 0, ""), (1, 0, " (no wait: dbg 1)"), (0, 1, " FP8 lo term"), (1, 1, " FP8 lo, no wait")):
        t = timeit(lambda: dev.lstm_fwd_cluster2(gh, cbuf, hcat, xn16, wcat, bcat, whf, whr, seq, status=st, dbg=dbg, rfmt=rf))
        print(f"cluster2 fwd (fp16 h, fused x-proj, tagged){nm:18s} {t:7.3f} ms  {t * 1e3 / Tf:6.2f} us/step", flush=True)
    print("status word after the forward kernels", int(st.item()), flush=True)
    for iobit, ionm in ((0, "output stores on the X-waves after their MFMAs"),):
      if not a.no_stamps:
        for rep in range(2):
            dbuf = torch.zeros(Tf * 2 * 8 * 2 + 256 * 4 * 2, device=d)   # (+ the per-workgroup wall-clock rows of round 6)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            dev.lstm_fwd_cluster2(gh, cbuf, hcat, xn16, wcat, bcat, whf, whr, seq, status=st, dbg=2048 + iobit, dbg_buf=dbuf, rfmt=0)
            e1.record()
            torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        ts = dbuf.view(torch.int64)[: Tf * 16].view(Tf, 2, 8).cpu().double()
        span = float(ts[-1, 0, 0] - ts[5, 0, 0]) / (Tf - 6)
        upt = (ms * 1e3 / Tf) / span if span > 0 else float("nan")
        print(f"--- stamps, cluster2 forward, {ionm}: launch {ms:.3f} ms = {ms * 1e3 / Tf:.2f} us/step with stamps; {span:.0f} ticks per step")
        cn = ["loop top", "recurrent MFMAs done", "past barrier 0", "cell update done (h published)", "next x-projection done",
              "X: all eight slices arrived", "X: h image written"]
        for role, rn in ((0, "X-wave 0 (tile 0)"), (1, "M-wave 4 (tile 1)")):
            tt = ts[5:-1, role] - ts[5:-1, role, 0:1]
            nxt = ts[6:, role, 0] - ts[5:-1, role, 0]
            print(f"  {rn}: mean microseconds since the loop top")
            print(f"     {'x feed done':34s} {float(tt[:, 7].mean()) * upt:6.2f}   (sd {float(tt[:, 7].std()) * upt:.2f})")
            for k in range(1, 7):
                if role == 1 and k > 4:
                    continue
                print(f"     {cn[k]:34s} {float(tt[:, k].mean()) * upt:6.2f}   (sd {float(tt[:, k].std()) * upt:.2f})")
            print(f"     {'next loop top (past barrier 2)':34s} {float(nxt.mean()) * upt:6.2f}", flush=True)
    st.zero_()
    dev.lstm_fwd_cluster(gh, cbuf, hcat, whf, whr, seq, status=st, gfmt=L.GATES_H2, gates_in=pre)   # (state for the BPTT below)
    # ---- BPTT (pair kernel) ----------------------------------------------------------------------------------------
    pp, pp16 = torch.empty(L.LSTM_PACK_FLOATS, device=d), torch.empty(L.LSTM_PACK_FLOATS, device=d)
    dev.lstm_pack_pair(whf, whr, pp)
    dev.lstm_pack_pair(whf, whr, pp16, f16=True)
    pp8 = torch.empty(L.LSTM_PACK_FLOATS, device=d)
    dev.lstm_pack_pair(whf, whr, pp8, f16=2)
    pp8mx = torch.empty(L.LSTM_PACK_FLOATS, device=d)
    dev.lstm_pack_pair(whf, whr, pp8mx, f16=3)
    gq = dev.blh_gates_unpack(gh, nb).view_as(pre).contiguous()
    dgo = torch.zeros_like(gh)
    work = gq.clone()

    def f32():
        work.copy_(gq)
        dev.lstm_bwd_pair(work, cbuf, dh, pp, seq, status=st)
    t0 = timeit(lambda: work.copy_(gq))
    t = timeit(f32) - t0
    print(f"pair BPTT, fp32 gates in place, bf16x3       {t:7.3f} ms  {t * 1e3 / Tf:6.2f} us/step", flush=True)
    for rf, pk, nm, dbg in ((0, pp, "bf16x3 recurrence", 0), (1, pp16, "fp16x2 recurrence, tagged", 0),
                            (2, pp8, "fp16x2, FP8 lo plane resident", 0), (3, pp8mx, "fp16 + FP8 MFMA lo term (rfmt 3)", 0)):
        t = timeit(lambda: dev.lstm_bwd_pair(gh, cbuf, dh, pk, seq, status=st, gfmt=L.GATES_H2F, dgates=dgo, amax=amax, rfmt=rf,
                                             dbg=dbg))
        print(f"pair BPTT, unorm16 in / fp16 out, {nm}  {t:7.3f} ms  {t * 1e3 / Tf:6.2f} us/step", flush=True)
    print("status word", int(st.item()), flush=True)
    if a.no_stamps:
        return
    names = ["loop top", "cell backward done", "past S1", "MFMA loop done", "X: flagged / O: partial in LDS",
             "X: next loads requested", "X: partner's flag seen", "X: gather arrived"]
    for rf, pk, nm, xdbg in ((0, pp, "bf16x3, flag hand-off", 0), (1, pp16, "fp16x2 tagged", 0), (2, pp8, "fp16x2 tagged, FP8 lo resident", 0)):
        for rep in range(2):
            dbuf = torch.zeros(Tf * 2 * 8 * 2 + 256 * 4 * 2, device=d)   # (+ the per-workgroup wall-clock rows of round 6)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            dev.lstm_bwd_pair(gh, cbuf, dh, pk, seq, status=st, gfmt=L.GATES_H2F, dgates=dgo, amax=amax, rfmt=rf, dbg=2048 + xdbg,
                              dbg_buf=dbuf)
            e1.record()
            torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        ts = dbuf.view(torch.int64)[: Tf * 16].view(Tf, 2, 8).cpu().double()
        span = float(ts[-1, 0, 0] - ts[5, 0, 0]) / (Tf - 6)
        upt = (ms * 1e3 / Tf) / span if span > 0 else float("nan")
        print(f"--- stamps, WS_GATES_H2F {nm}: launch {ms:.3f} ms = {ms * 1e3 / Tf:.2f} us/step with stamps; {span:.0f} ticks per step")
        for role, rn in ((0, "X-wave 0"), (1, "O-wave 4")):
            tt = ts[5:-1, role] - ts[5:-1, role, 0:1]
            nxt = ts[6:, role, 0] - ts[5:-1, role, 0]
            print(f"  {rn}: mean microseconds since the loop top")
            for k in range(1, 8):
                if role == 1 and k > 4:
                    continue
                print(f"     {names[k]:34s} {float(tt[:, k].mean()) * upt:6.2f}   (sd {float(tt[:, k].std()) * upt:.2f})")
            print(f"     {'next loop top (past S2)':34s} {float(nxt.mean()) * upt:6.2f}", flush=True)


if __name__ == "__main__":
    main()
