"""Diagnosis of the pair BPTT kernel (lstm_pair.hip) on the MI355X: where does its d(gates) differ from the streaming
kernel's, and does every hand-off deliver what was sent?

    python tools/pair_diag.py [--rows 2] [--frames 1,2,3,4,8,70] [--dbg 0,64]

For each number of steps L: the blocked forward (16-sequence kernel), then the streaming BPTT and the pair BPTT on the
same state.  Prints the relative difference per (direction, time step) and -- for the worst step -- per
(member hs, m-tile wx, cell group q4), i.e. per wave role; and compares what each X-wave SENT with what the partner's
X-wave RECEIVED at every step (dbg_buf of ws_lstm_pair_args).  Not part of the product."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wesep_amd import _lib as L, dev  # noqa: E402
from wesep_amd.functional import _view_maps  # noqa: E402


def stamps(R, Tf=501):
    """Phase stamps of one X-wave and one O-wave of pair 0 / member 0 (lstm_pair.hip, variant 2048)."""
    d = torch.device("cuda:0")
    g = torch.Generator().manual_seed(7)
    H, N, K = 256, 128, 32
    whf = (0.06 * torch.randn(4 * H, H, generator=g)).to(d)
    whr = (0.06 * torch.randn(4 * H, H, generator=g)).to(d)
    _, _, seq, _ = _view_maps("time", R, K, Tf, N)
    nb = dev.bl_num_blocks(seq)
    gates = torch.rand(nb, 32 * 8 * H, device=d) * 0.9 + 0.05
    cbuf = torch.randn(nb, 2 * H // 4, 32, 4, device=d) * 0.5
    dh = torch.randn(nb, 2 * H // 4, 32, 4, device=d) * 0.1
    pp = torch.empty(L.LSTM_PACK_FLOATS, device=d)
    dev.lstm_pack_pair(whf, whr, pp)
    for rep in range(2):
        dbuf = torch.zeros(Tf * 2 * 8 * 2 + 256 * 4 * 2, device=d)
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        g2 = gates.clone()
        t0.record()
        dev.lstm_bwd_pair(g2, cbuf, dh, pp, seq, dbg=2048, dbg_buf=dbuf)
        t1.record()
        torch.cuda.synchronize()
    ms = t0.elapsed_time(t1)
    ts = dbuf.view(torch.int64)[: Tf * 16].view(Tf, 2, 8).cpu().double()
    span = float(ts[-1, 0, 0] - ts[5, 0, 0]) / (Tf - 6)
    us_per_tick = (ms * 1e3 / Tf) / span if span > 0 else float("nan")
    print(f"launch {ms:.3f} ms = {ms * 1e3 / Tf:.2f} us/step with stamps; {span:.0f} ticks per step -> {us_per_tick * 1e3:.2f} ns per tick")
    names = ["loop top", "cell backward done", "past S1", "MFMA loop done", "X: flagged / O: partial in LDS",
             "X: next loads requested", "X: partner's flag seen", "X: gather arrived"]
    for role, nm in ((0, "X-wave 0"), (1, "O-wave 4")):
        t = ts[5:-1, role] - ts[5:-1, role, 0:1]
        nxt = ts[6:, role, 0] - ts[5:-1, role, 0]
        print(f"  {nm}: mean microseconds since the loop top")
        for k in range(1, 8):
            if role == 1 and k > 4:
                continue
            print(f"     {names[k]:34s} {float(t[:, k].mean()) * us_per_tick:6.2f}   (sd {float(t[:, k].std()) * us_per_tick:.2f})")
        print(f"     {'next loop top (past S2)':34s} {float(nxt.mean()) * us_per_tick:6.2f}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=2)
    ap.add_argument("--frames", default="1,2,3,4,8,70")
    ap.add_argument("--dbg", default="0,64")
    ap.add_argument("--ts", type=int, default=0, help="rows: print the in-kernel phase stamps (variant 2048) at that size")
    a = ap.parse_args()
    if a.ts:
        return stamps(a.ts)
    d = torch.device("cuda:0")
    g = torch.Generator().manual_seed(7)
    H, N, K = 256, 128, 32
    whf = (0.06 * torch.randn(4 * H, H, generator=g)).to(d)
    whr = (0.06 * torch.randn(4 * H, H, generator=g)).to(d)
    for Tf in [int(x) for x in a.frames.split(",")]:
        R = a.rows
        P = R * K * Tf
        _, _, seq, _ = _view_maps("time", R, K, Tf, N)
        nb = dev.bl_num_blocks(seq)
        ntile = -(-seq.nseq // 32)
        pre = torch.randn(P, 8 * H, generator=g).to(d)
        gates = dev.to_blocked(pre, seq)
        pf, pb, pp = (torch.empty(L.LSTM_PACK_FLOATS, device=d) for _ in range(3))
        dev.lstm_pack(whf, whr, pf, pb, L.LSTM_BF16X3_BLK16)
        dev.lstm_pack_pair(whf, whr, pp)
        cbuf = torch.zeros(nb, 2 * H // 4, 32, 4, device=d)
        hcat = torch.zeros_like(cbuf)
        dev.lstm_fwd(gates, cbuf, hcat, pf, seq, L.LSTM_BF16X3_BLK16)
        dh = dev.to_blocked((0.1 * torch.randn(P, 2 * H, generator=g)).to(d), seq)
        ref = gates.clone()
        dev.lstm_bwd(ref, cbuf, hcat, dh, pb, seq, L.LSTM_BF16X3_BLK16)
        ref_u = dev.bls_unpack(ref).view(ntile, Tf, 2, 4, 64, 32, 4)          # [tile][t][d][gate][quad][slot][r]
        for dbg in [int(x) for x in a.dbg.split(",")]:
            npair = 2 * ntile
            dbuf = torch.full((npair, Tf, 2, 2, 1024, 4), float("nan"), device=d)
            status = torch.zeros(1, device=d, dtype=torch.int32)
            outs = []
            for rep in range(2):
                g2 = gates.clone()
                dev.lstm_bwd_pair(g2, cbuf, dh, pp, seq, status=status, dbg=dbg, dbg_buf=dbuf if rep == 0 else None)
                torch.cuda.synchronize()
                outs.append(g2)
            got = dev.bls_unpack(outs[0]).view(ntile, Tf, 2, 4, 64, 32, 4)
            same = torch.equal(outs[0], outs[1])
            print(f"   NaN count: pair {int(torch.isnan(got).sum())}  streaming {int(torch.isnan(ref_u).sum())}; "
                  f"elements differing between the two pair runs: {int((outs[0] != outs[1]).sum())} of {outs[0].numel()}")
            bad_el = ((got - ref_u).abs() > 1e-3 * ref_u.abs().max()) | torch.isnan(got)
            if bool(bad_el.any()):
                w_ = bad_el.nonzero()
                print(f"   {int(bad_el.sum())} elements off; by direction {[int(bad_el[:, :, k].sum()) for k in range(2)]}, "
                      f"by gate {[int(bad_el[:, :, :, k].sum()) for k in range(4)]}, by hs "
                      f"{[int(bad_el[:, :, :, :, 32 * k:32 * k + 32].sum()) for k in range(2)]}")
                # pattern of the wrong elements in kernel coordinates: quad -> (hs, wx, role, e, half)
                qd = w_[:, 4]
                hs_, ql = qd // 32, qd % 32
                wx_, rem = ql // 8, ql % 8
                role_, e_, half_ = rem // 4, (rem % 4) // 2, rem % 2
                for nm, v in (("role", role_), ("e", e_), ("half", half_), ("wx", wx_), ("r", w_[:, 6]),
                              ("slot%16", w_[:, 5] % 16), ("slot//16", w_[:, 5] // 16)):
                    print(f"      by {nm}: {torch.bincount(v).tolist()}")
                raw_g = outs[0].view(ntile, Tf, 2, 4, 64, 32, 4).view(torch.int32)
                raw_r = ref.view(ntile, Tf, 2, 4, 64, 32, 4).view(torch.int32)
                for row in w_[:6].tolist():
                    print("      raw bits got %08x want %08x" % (int(raw_g[tuple(row)]) & 0xffffffff,
                                                                 int(raw_r[tuple(row)]) & 0xffffffff))
                for row in w_[:6].tolist():
                    print("    [tile, t, d, gate, quad, slot, r] =", row, " got", float(got[tuple(row)]), " want",
                          float(ref_u[tuple(row)]))
            err = (got - ref_u).double()
            tot = float(err.norm() / ref_u.double().norm())
            print(f"L={Tf:3d} dbg={dbg:2d}: rel {tot:.3e}  run-to-run identical {same}  status {int(status.item())}")
            # hand-off: sent by (pair, step, hs) must equal received by (pair, step, 1 - hs)
            sent, recv = dbuf[:, :, :, 0], dbuf[:, :, :, 1]
            bad = (recv != sent.flip(2))
            nb_ = int(bad.sum())
            if nb_:
                where = bad.nonzero()
                steps = sorted(set(where[:, 1].tolist()))
                print(f"   HAND-OFF MISMATCH: {nb_} floats; steps {steps[:12]}{'...' if len(steps) > 12 else ''}; "
                      f"first: pair {int(where[0, 0])} step {int(where[0, 1])} receiver hs {int(where[0, 2])} "
                      f"cell {int(where[0, 3])}")
                # is the received value the one sent two steps earlier (a stale slot)?
                if Tf > 2:
                    stale = (recv[:, 2:] == sent.flip(2)[:, :-2]) & bad[:, 2:]
                    print(f"   of which equal to the value sent two steps earlier (stale slot): {int(stale.sum())}")
            else:
                print("   hand-off: every received partial equals what the partner sent")
            if tot > 1e-4:
                for dd in range(2):
                    per_t = [(float(err[:, t, dd].norm() / (ref_u[:, t, dd].double().norm() + 1e-30))) for t in range(Tf)]
                    order = list(range(Tf - 1, -1, -1)) if dd == 0 else list(range(Tf))
                    print(f"   dir {dd} rel by processing step:", " ".join(f"{per_t[t]:.1e}" for t in order[:10]))
                # structure at the first wrong processing step of direction 0
                t_first = next((t for t in range(Tf - 1, -1, -1)
                                if float(err[:, t, 0].norm() / (ref_u[:, t, 0].double().norm() + 1e-30)) > 1e-4), None)
                if t_first is not None:
                    e = err[:, t_first, 0]                   # [tile][gate][quad 64][slot][r]
                    r = ref_u[:, t_first, 0].double()
                    for hs in range(2):
                        row = []
                        for wx in range(4):
                            for q4 in range(4):
                                qs = [32 * hs + 8 * wx + 2 * q4 + h for h in range(2)]
                                row.append(float(e[:, :, qs].norm() / (r[:, :, qs].norm() + 1e-30)))
                        print(f"   dir 0 t={t_first} hs={hs} by (wx, q4):", " ".join(f"{v:.0e}" for v in row))


if __name__ == "__main__":
    main()
