"""Diagnostic (GPU): the blocked-layout BLSTM + Linear + residual of TF-GridNet (functional_tfgridnet.BlstmLinearBlkFn)
stage by stage against an fp64 torch statement, for the sequence geometries of the failing hardware tests and for
each recurrence selection (cluster / 16-sequence / 32-sequence streaming kernels).
Usage: python tools/diag_tfgrid_blk.py [nseq,L ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wesep_amd import _lib as L, dev  # noqa: E402
from wesep_amd import functional_tfgridnet as FG  # noqa: E402
from wesep_amd.dev import BIG, SeqMap  # noqa: E402

d = torch.device("cuda:0")
H, N = 256, 128


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def reference(y, res, nseq, Lr, P):
    """fp64 BLSTM + Linear + residual with explicit pre-activations (leaf-like, retain_grad)."""
    yd = y.double().view(nseq, Lr, N).requires_grad_(True)
    W = {k: v.double().requires_grad_(True) for k, v in P.items()}
    outs, pres = [], []
    for di, sfx in ((0, "f"), (1, "r")):
        pre = yd @ W["wih_" + sfx].t() + W["b_" + sfx]
        pre.retain_grad()
        pres.append(pre)
        h = torch.zeros(nseq, H, device=d, dtype=torch.float64)
        c = torch.zeros_like(h)
        hs = [None] * Lr
        for t in (range(Lr) if di == 0 else range(Lr - 1, -1, -1)):
            g = pre[:, t] + h @ W["whh_" + sfx].t()
            i, f, gg, o = g.chunk(4, 1)
            c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
            h = torch.sigmoid(o) * torch.tanh(c)
            hs[t] = h
        outs.append(torch.stack(hs, 1))
    hcat = torch.cat(outs, 2)
    hcat.retain_grad()
    out = res.double().view(nseq, Lr, N) + hcat @ W["lin_w"].t() + W["lin_b"]
    return yd, W, pres, hcat, out


def run(nseq, Lr, tag):
    g = torch.Generator().manual_seed(nseq * 1000 + Lr)
    r = lambda *s, sc=1.0: (sc * torch.randn(*s, generator=g)).to(d)
    P = dict(wih_f=r(4 * H, N, sc=0.08), wih_r=r(4 * H, N, sc=0.08), b_f=r(4 * H, sc=0.1), b_r=r(4 * H, sc=0.1),
             whh_f=r(4 * H, H, sc=0.06), whh_r=r(4 * H, H, sc=0.06), lin_w=r(N, 2 * H, sc=0.05), lin_b=r(N, sc=0.1))
    y, res, dout = r(nseq * Lr, N), r(nseq * Lr, N), r(nseq * Lr, N)
    yd, W, pres, hcat_ref, out_ref = reference(y, res, nseq, Lr, P)
    out_ref.backward(dout.double().view(nseq, Lr, N))
    leaves = [P[k].clone().requires_grad_(True) for k in ("wih_f", "wih_r", "b_f", "b_r", "whh_f", "whh_r", "lin_w", "lin_b")]
    yl, rl = y.clone().requires_grad_(True), res.clone().requires_grad_(True)
    out = FG.BlstmLinearBlkFn.apply(yl, rl, (nseq, Lr), None, None, *leaves)
    out.backward(dout)
    torch.cuda.synchronize()
    names = ("wih_f", "wih_r", "b_f", "b_r", "whh_f", "whh_r", "lin_w", "lin_b")
    line = [f"out {rel(out, out_ref.view(-1, N)):.1e}", f"dy {rel(yl.grad, yd.grad.view(-1, N)):.1e}",
            f"dres {rel(rl.grad, dout):.1e}"]
    line += [f"d{n} {rel(l.grad, W[n].grad):.1e}" for n, l in zip(names, leaves)]
    print(f"[{tag}] nseq {nseq} L {Lr}: " + "  ".join(line), flush=True)
    return P, y, res, dout, pres, hcat_ref


def stages(nseq, Lr):
    """The forward / backward launch sequence by hand, each intermediate against the reference."""
    P, y, res, dout, pres, hcat_ref = run(nseq, Lr, "fn")
    pad = 0
    if Lr >= 64 and nseq % 64:
        pad = -(-nseq // 64) * 64 - nseq
    ns = nseq + pad
    z = torch.zeros(pad * Lr, N, device=d)
    yp = torch.cat([y, z], 0).contiguous()
    seq = SeqMap(ns, BIG, 0, Lr, 1, Lr)
    nb = dev.bl_num_blocks(seq)
    G4 = 4 * H
    zero = torch.zeros(G4, device=d)
    wcat, bcat = torch.empty(2 * G4, N, device=d), torch.empty(2 * G4, device=d)
    dev.lstm_cat_ih(P["wih_f"], P["wih_r"], P["b_f"], zero, P["b_r"], zero, N, wcat, bcat)
    wih_pack = torch.empty(2 * G4 * N, device=d)
    dev.pack_w(wcat, 2 * G4, N, N, wih_pack, order=0)
    gates0, xn = torch.empty(nb, 32 * 2 * G4, device=d), torch.empty(nb, 32 * N, device=d)
    dev.gemm_p2b(A=yp, lda=N, sm=seq, Wpack=wih_pack, N=2 * G4, C_out=gates0, bias=bcat, A_bl=xn)
    Pn = ns * Lr
    pre = dev.from_blocked(gates0.view(nb, 2 * G4 // 4, 32, 4), seq, Pn)[:nseq * Lr]
    pre_ref = torch.cat([pres[0].detach().reshape(-1, G4), pres[1].detach().reshape(-1, G4)], 1)
    print(f"   p2b pre-activations {rel(pre, pre_ref):.1e}   xn {rel(dev.from_blocked(xn.view(nb, N // 4, 32, 4), seq, Pn)[:nseq * Lr], y):.1e}",
          flush=True)
    status = torch.zeros(1, device=d, dtype=torch.int32)
    pf, pb = torch.empty(L.LSTM_PACK_FLOATS, device=d), torch.empty(L.LSTM_PACK_FLOATS, device=d)
    for kind in ("cluster", "s16", "s32"):
        gates = gates0.clone()
        cbuf, hcat = torch.zeros(nb, 32 * 2 * H, device=d), torch.zeros(nb, 32 * 2 * H, device=d)
        try:
            if kind == "cluster":
                if ns % 64 or (ns // 32) * 8 > dev.cu_count(d):
                    continue
                dev.lstm_fwd_cluster(gates, cbuf, hcat, P["whh_f"], P["whh_r"], seq, status=status)
            else:
                mode = L.LSTM_BF16X3_BLK16 if kind == "s16" else L.LSTM_BF16X3_BLK
                dev.lstm_pack(P["whh_f"], P["whh_r"], pf, pb, mode)
                dev.lstm_fwd(gates, cbuf, hcat, pf, seq, mode)
            torch.cuda.synchronize()
        except Exception as e:  # noqa: BLE001
            print(f"   fwd {kind}: EXC {e}", flush=True)
            continue
        hc = dev.from_blocked(hcat.view(nb, 2 * H // 4, 32, 4), seq, Pn)[:nseq * Lr]
        print(f"   fwd {kind}: hcat {rel(hc, hcat_ref.detach().reshape(-1, 2 * H)):.1e} status {int(status.item())} "
              f"finite {bool(torch.isfinite(hcat).all())}", flush=True)
        # BPTT with each backward kernel on this forward's state
        wlt_pack = torch.empty(2 * H * N, device=d)
        dev.pack_w(P["lin_w"].contiguous(), 2 * H, N, 2 * H, wlt_pack, trans=True, order=0)
        doutp = torch.cat([dout, z], 0).contiguous()
        dh, dout_bl = torch.empty(nb, 32 * 2 * H, device=d), torch.empty(nb, 32 * N, device=d)
        dev.gemm_p2b(A=doutp, lda=N, sm=seq, Wpack=wlt_pack, N=2 * H, C_out=dh, A_bl=dout_bl)
        dh_ref = hcat_ref.grad.reshape(-1, 2 * H)
        print(f"      p2b d(hcat) {rel(dev.from_blocked(dh.view(nb, 2 * H // 4, 32, 4), seq, Pn)[:nseq * Lr], dh_ref):.1e}",
              flush=True)
        dpre_ref = torch.cat([pres[0].grad.reshape(-1, G4), pres[1].grad.reshape(-1, G4)], 1)
        for bk in ("cluster", "s16", "s32"):
            g2 = gates.clone()
            try:
                if bk == "cluster":
                    if ns % 64 or (ns // 32) * 8 > dev.cu_count(d):
                        continue
                    dev.lstm_bwd_cluster(g2, cbuf, dh.clone(), P["whh_f"], P["whh_r"], seq, status=status)
                else:
                    mode = L.LSTM_BF16X3_BLK16 if bk == "s16" else L.LSTM_BF16X3_BLK
                    dev.lstm_pack(P["whh_f"], P["whh_r"], pf, pb, mode)
                    dev.lstm_bwd(g2, cbuf, hcat, dh.clone(), pb, seq, mode)
                torch.cuda.synchronize()
            except Exception as e:  # noqa: BLE001
                print(f"      bwd {bk}: EXC {e}", flush=True)
                continue
            dg = dev.from_blocked(g2.view(nb, 2 * G4 // 4, 32, 4), seq, Pn)[:nseq * Lr]
            padrows = dev.from_blocked(g2.view(nb, 2 * G4 // 4, 32, 4), seq, Pn)[nseq * Lr:]
            print(f"      bwd {bk}: dgates {rel(dg, dpre_ref):.1e} status {int(status.item())} "
                  f"pad-seq |dgates| max {float(padrows.abs().max()) if padrows.numel() else 0.0:.1e}", flush=True)


if __name__ == "__main__":
    geos = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(202, 65), (130, 101), (195, 66), (256, 65),
                                                                             (64, 70), (100, 20)]
    for nseq, Lr in geos:
        try:
            stages(nseq, Lr)
        except Exception as e:  # noqa: BLE001
            import traceback
            traceback.print_exc()
            print(f"nseq {nseq} L {Lr}: EXC {e}", flush=True)
