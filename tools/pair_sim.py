"""Index-level model of wesep_amd/csrc/lstm_pair.hip (the pair BPTT kernel) on the CPU.

Not a numerical emulation (tests/emu_blk.py has that): this walks the kernel's OWN index arithmetic -- weight pack
layout, MFMA fragment maps, the cell <-> (wave, lane) assignment, BL byte offsets, exchange offsets, rec slots -- lane
by lane, and compares the d(gates) it leaves in the blocked buffer with a plain BPTT.  It exists because GPU time is
scarce: a wrong shift or a swapped role shows up here, not on the MI355X.  Every formula below is transcribed from the
kernel source and cites the variable it mirrors.

    python tools/pair_sim.py            # prints the max relative difference per direction
"""
import numpy as np

H, G4, SQ = 256, 1024, 32


def bf16(x):
    """round-to-nearest-even bf16 of float32 values, returned as float32"""
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def split(x):
    hi = bf16(x)
    return hi, bf16(np.asarray(x, np.float32) - hi)


def pack_pair(whh):
    """lstm_pack_pair_kernel: whh [2][4H][H] -> out[unit][8] (bf16 values as float32)"""
    out = np.zeros((2 * 2 * 8 * 2 * 32 * 64, 8), np.float32)
    idx = np.arange(2 * G4 * H)
    j = idx & 7
    lane = (idx >> 3) & 63
    ks = (idx >> 9) & 31
    w = (idx >> 14) & 7
    hs = (idx >> 17) & 1
    d = idx >> 18
    mt = np.where(w < 4, 4 * (1 - hs) + w, 4 * hs + (w - 4))
    u = 32 * mt + (lane & 31)
    kl = 16 * ks + 8 * (lane >> 5) + j
    row = (kl >> 7) * H + 128 * hs + (kl & 127)
    v = whh[d, row, u]
    hi, lo = split(v)
    unit = ((d * 2 + hs) * 8 + w) * 2 * (32 * 64) + ks * 64 + lane
    out[unit, j] = hi
    out[unit + 32 * 64, j] = lo
    return out


def mfma32(a, b, c):
    """v_mfma_f32_32x32x16_bf16: a, b [64 lanes][8]; c [64 lanes][16].  A[i = l & 31][k = 8 (l >> 5) + j],
    B[k = 8 (l >> 5) + j][n = l & 31], D: lane l reg r -> row (r & 3) + 8 (r >> 2) + 4 (l >> 5), col l & 31"""
    A = np.zeros((32, 16))
    B = np.zeros((16, 32))
    l = np.arange(64)
    for j in range(8):
        A[l & 31, 8 * (l >> 5) + j] = a[:, j]
        B[8 * (l >> 5) + j, l & 31] = b[:, j]
    D = A @ B
    out = c.copy()
    for r in range(16):
        out[:, r] += D[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5), l & 31]
    return out


def bl_off(C, quad, slot):
    """float index of the 4-float cell (column quad, slot) inside one BL(C) block"""
    return (quad * 32 + slot) * 4


def simulate(ntile=1, L=3, seed=0):
    rng = np.random.default_rng(seed)
    nseq = ntile * SQ
    whh = (rng.standard_normal((2, G4, H)) * 0.06).astype(np.float32)
    # activated gates, cells, dh in natural layout [seq][t][dir][...]
    act = rng.uniform(0.05, 0.95, (nseq, L, 2, 4, H)).astype(np.float32)
    act[:, :, :, 2] = act[:, :, :, 2] * 2 - 1
    cs = (rng.standard_normal((nseq, L, 2, H)) * 0.5).astype(np.float32)
    dh = (rng.standard_normal((nseq, L, 2, H)) * 0.1).astype(np.float32)

    # ---- reference BPTT (float64) ----
    ref = np.zeros((nseq, L, 2, 4, H))
    for d in range(2):
        W = whh[d].astype(np.float64)
        dc = np.zeros((nseq, H))
        dhr = np.zeros((nseq, H))
        order = range(L - 1, -1, -1) if d == 0 else range(L)
        for t in order:
            i, f, g, o = (act[:, t, d, k].astype(np.float64) for k in range(4))
            tp = t - 1 if d == 0 else t + 1
            cprev = cs[:, tp, d].astype(np.float64) if 0 <= tp < L else np.zeros((nseq, H))
            tc = np.tanh(cs[:, t, d].astype(np.float64))
            dhv = dh[:, t, d] + dhr
            dov = dhv * tc
            dcv = dc + dhv * o * (1 - tc * tc)
            dc = dcv * f
            dp = np.stack([dcv * g * i * (1 - i), dcv * cprev * f * (1 - f), dcv * i * (1 - g * g), dov * o * (1 - o)], 1)
            ref[:, t, d] = dp
            dhr = dp.reshape(nseq, G4) @ W

    # ---- blocked buffers: block b = tile * L + t ----
    nb = ntile * L
    gates = np.zeros((nb, 32 * 2 * G4), np.float32)
    cbuf = np.zeros((nb, 32 * 2 * H), np.float32)
    dhc = np.zeros((nb, 32 * 2 * H), np.float32)
    for tile in range(ntile):
        for t in range(L):
            b = tile * L + t
            for slot in range(32):
                s = tile * 32 + slot
                col = act[s, t].reshape(2 * G4)                      # column = d*1024 + g*256 + u
                gates[b].reshape(2 * G4 // 4, 32, 4)[:, slot, :] = col.reshape(-1, 4)
                cbuf[b].reshape(2 * H // 4, 32, 4)[:, slot, :] = cs[s, t].reshape(-1, 4)
                dhc[b].reshape(2 * H // 4, 32, 4)[:, slot, :] = dh[s, t].reshape(-1, 4)
    pack = pack_pair(whh)

    lane = np.arange(64)
    n, half = lane & 31, lane >> 5
    for pr in range(2 * ntile):
        d, tile = pr & 1, pr >> 1
        # per member hs, per wave w: state
        st = {}
        for hs in range(2):
            for w in range(8):
                wx, role = w & 3, w >> 2
                q0 = 8 * wx + 4 * role + half
                base = ((d * 2 + hs) * 8 + w) * 2 * (32 * 64)
                wh = pack[base: base + 32 * 64].reshape(32, 64, 8)
                wl = pack[base + 32 * 64: base + 2 * 32 * 64].reshape(32, 64, 8)
                t0 = L - 1 if d == 0 else 0
                st[hs, w] = dict(wx=wx, role=role, q0=q0, wh=wh, wl=wl, dc=np.zeros((2, 64, 4)), mine=np.zeros((2, 64, 4)),
                                 c_cur=np.stack([cbuf[tile * L + t0].reshape(-1, 4)[bl_off(512, d * 64 + 32 * hs + q0 + 2 * e, n) // 4]
                                                 for e in range(2)]))
        rec = {hs: np.zeros((2, 512, 4)) for hs in range(2)}
        xch = np.zeros((2, 2, 1024, 4))       # [parity][dest][cell]
        for step in range(L):
            t = L - 1 - step if d == 0 else step
            has_prev = t > 0 if d == 0 else t < L - 1
            par = step & 1
            b = tile * L + t
            gblk = gates[b].reshape(-1, 4)
            cblk_t = cbuf[b].reshape(-1, 4)
            tp = max(t - 1, 0) if d == 0 else min(t + 1, L - 1)
            cblk_p = cbuf[tile * L + tp].reshape(-1, 4)
            dblk = dhc[b].reshape(-1, 4)
            bimg = {hs: np.zeros((2, 32, 512)) for hs in range(2)}
            # phase C
            for hs in range(2):
                for w in range(8):
                    S = st[hs, w]
                    oth = rec[hs][S["role"]]
                    for e in range(2):
                        q = S["q0"] + 2 * e
                        gq = lambda g: (bl_off(2048, d * 256 + g * 64 + 32 * hs + q, n) // 4)
                        cq = bl_off(512, d * 64 + 32 * hs + q, n) // 4
                        ig, fg, gg, og = (gblk[gq(g)].astype(np.float64) for g in range(4))
                        dhr = S["mine"][e] + oth[(S["wx"] * 2 + e) * 64 + lane]
                        dhv = dblk[cq] + dhr
                        tc = np.tanh(S["c_cur"][e].astype(np.float64))
                        dov = dhv * tc
                        dcv = S["dc"][e] + dhv * og * (1 - tc * tc)
                        S["dc"][e] = dcv * fg
                        cp = cblk_p[cq].astype(np.float64)
                        outs = [dcv * gg * ig * (1 - ig), dcv * (cp if has_prev else 0) * fg * (1 - fg),
                                dcv * ig * (1 - gg * gg), dov * og * (1 - og)]
                        S["c_cur"][e] = cp
                        for g in range(4):
                            hi, lo = split(outs[g].astype(np.float32))
                            for r in range(4):
                                bimg[hs][0][n, g * 128 + 4 * q + r] = hi[:, r]
                                bimg[hs][1][n, g * 128 + 4 * q + r] = lo[:, r]
                            gblk[gq(g)] = (hi + lo)            # BLS value = hi + lo
            # MFMA + exchange
            sums = {}
            for hs in range(2):
                for w in range(8):
                    S = st[hs, w]
                    acc = np.zeros((64, 16))
                    for ks in range(32):
                        kidx = 16 * ks + 8 * half[:, None] + np.arange(8)[None, :]
                        bh = bimg[hs][0][n[:, None], kidx]
                        bl = bimg[hs][1][n[:, None], kidx]
                        acc = mfma32(S["wh"][ks], bh, acc)
                        acc = mfma32(S["wl"][ks], bh, acc)
                        acc = mfma32(S["wh"][ks], bl, acc)
                    sums[hs, w] = acc.reshape(64, 4, 4)          # [lane][q4][r]
            for hs in range(2):
                for w in range(4):                                # X-waves publish
                    for q4 in range(4):
                        cell = (8 * w + half) * 32 + n + q4 * 64  # xc0 / 16 + q4 * 1024 / 16
                        xch[par, 1 - hs, cell] = sums[hs, w][:, q4]
            for hs in range(2):
                for w in range(8):
                    S = st[hs, w]
                    wx = S["wx"]
                    if w < 4:
                        pv = [xch[par, hs, (8 * wx + half) * 32 + n + q4 * 64] for q4 in range(4)]
                        S["mine"][0], S["mine"][1] = pv[0], pv[1]
                        rec[hs][1][(wx * 2) * 64 + lane] = pv[2]
                        rec[hs][1][(wx * 2 + 1) * 64 + lane] = pv[3]
                    else:
                        sm = sums[hs, w]
                        rec[hs][0][(wx * 2) * 64 + lane] = sm[:, 0]
                        rec[hs][0][(wx * 2 + 1) * 64 + lane] = sm[:, 1]
                        S["mine"][0], S["mine"][1] = sm[:, 2], sm[:, 3]
    # ---- compare ----
    res = []
    for d in range(2):
        num = den = 0.0
        for tile in range(ntile):
            for t in range(L):
                blk = gates[tile * L + t].reshape(2 * G4 // 4, 32, 4)
                for slot in range(32):
                    got = blk[:, slot, :].reshape(2, 4, H)[d]
                    want = ref[tile * 32 + slot, t, d]
                    num += ((got - want) ** 2).sum()
                    den += (want ** 2).sum()
        res.append(float(np.sqrt(num / den)))
    return res


if __name__ == "__main__":
    print("pair BPTT index model vs plain BPTT, rel-L2 per direction:", simulate())
