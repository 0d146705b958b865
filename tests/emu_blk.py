"""TEST INFRASTRUCTURE.  CPU emulation of the blocked-layout (BL) entry points of gemm_blk.hip / lstm_bf16*.hip /
lstm_cluster.hip / lstm_fused.hip, on top of tests/emu_dev.py: the plain <-> BL GEMMs, the recurrences on BL
buffers (16-sequence, 32-sequence, cluster and fused-projection variants are one function here: they differ in
scheduling, not in what they compute) and the BL x BL weight-gradient GEMM.  Packed-weight buffers are opaque on the
device; here a registry keyed by the pack buffer's address keeps the logical matrix.

Layout (include/wesep_hip.h): BL(C) block b = tile * L + step holds 32 consecutive sequences; element
(b, slot i, column c) at b*32*C + ((c >> 2)*32 + i)*4 + (c & 3).  Only tests/ import this; the product has no CPU
path."""
import math
import os

import torch

from wesep_amd import dev as real_dev

H, G4 = 256, 1024
_PACKS = {}          # pack buffer address -> logical content


def _ntile(sm):
    return -(-sm.nseq // 32)


def bl_get(buf, ntile, L, C):
    """BL(C) buffer -> [ntile*32 sequences, L steps, C] (padded sequences included)."""
    v = buf.reshape(-1)[: ntile * L * 32 * C].reshape(ntile, L, C // 4, 32, 4)
    return v.permute(0, 3, 1, 2, 4).reshape(ntile * 32, L, C)


def bl_put(buf, x, ntile, L, C):
    v = x.reshape(ntile, 32, L, C // 4, 4).permute(0, 2, 3, 1, 4)
    buf.reshape(-1)[: ntile * L * 32 * C] = v.reshape(-1)


# ---- 2-byte storage of the saved recurrence state (wesep_hip.h WS_GATES_H2 / H2S, ABI v15): BLH(C) = the BL(C) index
#      formula on 2-byte elements; the product hands these buffers over as float32 storage of half the element count
def blh_get(buf, ntile, L, C, dtype):
    v = buf.reshape(-1).view(dtype)[: ntile * L * 32 * C].reshape(ntile, L, C // 4, 32, 4)
    return v.permute(0, 3, 1, 2, 4).reshape(ntile * 32, L, C)


def blh_put(buf, x, ntile, L, C, dtype):
    v = x.reshape(ntile, 32, L, C // 4, 4).permute(0, 2, 3, 1, 4)
    buf.reshape(-1).view(dtype)[: ntile * L * 32 * C] = v.reshape(-1).to(dtype)


def _gate_is_tanh():
    """[2 * 4H] mask of the g gate's columns (column = dir * 4H + gate * H + unit)."""
    m = torch.zeros(2, 4, H, dtype=torch.bool)
    m[:, 2] = True
    return m.reshape(-1)


def gates_put_u16(buf, act, ntile, L):
    """activated gates [S, L, 2 * 4H] -> unorm16 codes (lstm_bf16_common.h enc_u16x4): i, f, o: floor(x * 65535 + 0.5);
    g: floor(x * 32767.5 + 32768)."""
    t = _gate_is_tanh()
    code = torch.where(t, torch.floor(act * 32767.5 + 32768.0), torch.floor(act * 65535.0 + 0.5)).to(torch.int32)
    code = torch.where(code >= 32768, code - 65536, code)          # uint16 bit pattern held in int16
    blh_put(buf, code, ntile, L, 2 * G4, torch.int16)


def gates_get_u16(buf, ntile, L):
    code = blh_get(buf, ntile, L, 2 * G4, torch.int16).to(torch.int32) & 0xFFFF
    t = _gate_is_tanh()
    x = code.float()
    return torch.where(t, x * (1.0 / 32767.5) - 1.0, x * (1.0 / 65535.0))


def _nv(sm):
    return getattr(sm, "nvalid", 0) or sm.nseq


def _valid(sm):
    return (torch.arange(_ntile(sm) * 32) < _nv(sm)).float().view(-1, 1, 1)


def _positions(sm):
    """row of the plain tensor for (sequence, step): [ntile*32, L] (padded sequences clamp to sequence nseq-1)."""
    s = torch.arange(_ntile(sm) * 32).clamp(max=_nv(sm) - 1).view(-1, 1)
    t = torch.arange(sm.L).view(1, -1)
    return (s // sm.div) * sm.s1 + (s % sm.div) * sm.s2 + t * sm.step_rows


def pack_w(W, N, K, ldw, out, trans=False, order=0, w_off=0, f16=False):
    flat_ = W.reshape(-1)
    n, k = torch.arange(N).view(-1, 1), torch.arange(K).view(1, -1)
    _PACKS[out.data_ptr()] = flat_[w_off + (k * ldw + n if trans else n * ldw + k)].clone()      # W'[n][k]


def lstm_cat_ih(wih_f, wih_r, bih_f, bhh_f, bih_r, bhh_r, n_in, wcat, bcat):
    wcat.reshape(2, G4, n_in)[0], wcat.reshape(2, G4, n_in)[1] = wih_f, wih_r
    bcat.reshape(2, G4)[0], bcat.reshape(2, G4)[1] = bih_f + bhh_f, bih_r + bhh_r


def lstm_pack_fused(wih_f, wih_r, whh_f, whh_r, pack, hfmt=0):
    _PACKS[pack.data_ptr()] = tuple(t.clone() for t in (wih_f, wih_r, whh_f, whh_r))


def _skip(run_if):
    """Predicated launches (wesep_hip.h run_if): a no-op unless the word is non-zero."""
    return run_if is not None and int(run_if.reshape(-1)[0]) == 0


def dgates_scale(amax):
    """WS_GATES_H2F (wesep_hip.h, common.h ws_dgates_scale); `amax`: 1-element int32 tensor holding float bits."""
    return real_dev.L.dgates_scale(int(amax.reshape(-1)[0]))


def gemm_p2b(*, A, lda, sm, Wpack, N, C_out, K=128, bias=None, A_bl=None, stats=None, gamma=None, beta=None,
             stat_map=None, run_if=None, amax=None, A_bl16=None):
    if _skip(run_if):
        return
    nt, L = _ntile(sm), sm.L
    pos = _positions(sm)
    rows = A.reshape(-1, lda)[pos.reshape(-1), :K].reshape(nt * 32, L, K)
    if stats is not None:
        d1, m1, d2, m2, base = stat_map
        s = (pos // d1) * m1 + (pos % d2) * m2 + base
        st = stats.reshape(-1, 2)
        rows = (rows - st[s, 0].unsqueeze(-1)) * st[s, 1].unsqueeze(-1) * gamma.reshape(-1)[:K] + beta.reshape(-1)[:K]
    rows = rows * _valid(sm)
    if A_bl is not None:
        bl_put(A_bl, rows, nt, L, K)
    if A_bl16 is not None:                 # ABI v16: the operand once more as fp16 in BLH(K)
        blh_put(A_bl16, rows, nt, L, K, torch.float16)
    if N:
        if _probe() & 8192 and amax is None:           # (numerics probe: a FORWARD x-projection on the fp16 copy of its input)
            rows = rows.half().float()
        out = rows @ _PACKS[Wpack.data_ptr()].t()
        if bias is not None:
            out = out + bias.reshape(-1)[:N]
        bl_put(C_out, out * _valid(sm), nt, L, N)
        if amax is not None:       # atomic max on the float bits of max |C|
            m = (out * _valid(sm)).abs().max().reshape(1).float()
            amax.reshape(-1)[0] = max(int(amax.reshape(-1)[0]), int(m.view(torch.int32)[0]))


def _g16(buf, nt, L, C, fmt, amax):
    """2-byte d(gates): fmt 1 = bf16, fmt 2 = fp16 scaled by dgates_scale(amax)."""
    if fmt == 1:
        return blh_get(buf, nt, L, C, torch.bfloat16).float()
    return blh_get(buf, nt, L, C, torch.float16).float() / dgates_scale(amax)


def gemm_b2p(*, A, K, sm, Wpack, C_out, ldc, N=128, bias=None, R=None, a_fmt=0, amax=None, a16_out=None):
    nt, L = _ntile(sm), sm.L
    if a16_out is not None:                # ABI v16: the split-pair A operand once more as fp16 in BLH(K)
        assert a_fmt == 0
        blh_put(a16_out, bl_get(A, nt, L, K), nt, L, K, torch.float16)
    x = (_g16(A, nt, L, K, a_fmt, amax) if a_fmt else bl_get(A, nt, L, K))[: _nv(sm)]
    out = x @ _PACKS[Wpack.data_ptr()].t()
    if bias is not None:
        out = out + bias.reshape(-1)[:N]
    pos = _positions(sm)[: _nv(sm)].reshape(-1)
    if R is not None:
        out = out + R.reshape(-1, ldc)[pos, :N].reshape(_nv(sm), L, N)
    C_out.reshape(-1, ldc)[pos, :N] = out.reshape(-1, N)


def _probe():
    return int(os.environ.get("WESEP_H2_PROBE", "0"))


def _w_hi_lo8(W):
    """256 W as fp16 hi + e4m3 codes of the residual, one exponent per [32 rows][64 k] fragment (lstm_pack_fused_h8_lo_kernel)."""
    s = 256.0 * W
    hi = s.half().float()
    res = (s - hi).reshape(W.shape[0] // 32, 32, W.shape[1] // 64, 64)
    mx = res.abs().amax(dim=(1, 3), keepdim=True)
    E = torch.where(mx > 0, torch.floor(torch.log2(mx.clamp_min(1e-45))) - 7, torch.zeros_like(mx)).clamp_min(-126)
    lo = (res * torch.exp2(-E)).to(torch.float8_e4m3fn).float() * torch.exp2(E)
    return hi / 256.0, lo.reshape(W.shape) / 256.0


def _recur_fwd(pre, whf, whr, hq16=False, hq8=False):
    """pre [S, L, 2, 4H] pre-activations -> (activated gates, c, h) of the same leading shape.  hq16: fp16 h in the recurrent
    product (ws_lstm_fwd_cluster2).  hq8 (hfmt 5): W_hh = fp16 hi + FP8 lo, the lo term against an e4m3 image of h."""
    S, L = pre.shape[:2]
    act, cs, hs = torch.zeros_like(pre), torch.zeros(S, L, 2, H), torch.zeros(S, L, 2, H)
    for d, W in ((0, whf), (1, whr)):
        h, c = torch.zeros(S, H), torch.zeros(S, H)
        if hq8:
            W, W8 = _w_hi_lo8(W)
        for t in (range(L) if d == 0 else range(L - 1, -1, -1)):
            hq = h.half().float() if hq16 or _probe() & 2048 else h    # (numerics probe: fp16 h in the recurrent product)
            p = pre[:, t, d] + hq @ W.t()
            if hq8:
                p = p + h.to(torch.float8_e4m3fn).float() @ W8.t()
            i, f, g, o = p[:, :H].sigmoid(), p[:, H:2 * H].sigmoid(), p[:, 2 * H:3 * H].tanh(), p[:, 3 * H:].sigmoid()
            c = f * c + i * g
            h = o * c.tanh()
            act[:, t, d], cs[:, t, d], hs[:, t, d] = torch.cat([i, f, g, o], 1), c, h
    return act, cs, hs


def _recur_bwd(act, cs, dh_in, whf, whr, scale=1.0, rq=False):
    """rq: the recurrent product takes the STORED scaled-fp16 d(gates) (ws_lstm_pair_args.rfmt = 1)."""
    S, L = act.shape[:2]
    dpre = torch.zeros_like(act)
    if _probe() & 32768:      # (numerics probe: W_hh of the BPTT as fp16 hi + fp8 (e4m3) lo of 256 w -- ~15 bits)
        def q8(W):
            s = 256.0 * W
            hi = s.half().float()
            lo = ((s - hi) * 4096.0).to(torch.float8_e4m3fn).float() / 4096.0
            return (hi + lo) / 256.0
        whf, whr = q8(whf), q8(whr)
    for d, W in ((0, whf), (1, whr)):
        order = list(range(L)) if d == 0 else list(range(L - 1, -1, -1))
        dh_rec, dc = torch.zeros(S, H), torch.zeros(S, H)
        for idx in range(L - 1, -1, -1):
            t = order[idx]
            a = act[:, t, d]
            i, f, g, o = a[:, :H], a[:, H:2 * H], a[:, 2 * H:3 * H], a[:, 3 * H:]
            c = cs[:, t, d]
            cprev = cs[:, order[idx - 1], d] if idx > 0 else torch.zeros_like(c)
            dh = dh_in[:, t, d] + dh_rec
            tc = c.tanh()
            dcv = dc + dh * o * (1 - tc * tc)
            dp = torch.cat([dcv * g * i * (1 - i), dcv * cprev * f * (1 - f), dcv * i * (1 - g * g),
                            dh * tc * o * (1 - o)], 1)
            dc = dcv * f
            # (numerics probe 4096: the recurrent product takes the STORED scaled-fp16 d(gates), one MFMA operand)
            dq = (dp * scale).half().float() / scale if rq or _probe() & 4096 else dp
            dh_rec = dq @ W
            dpre[:, t, d] = dp
    return dpre


def _fwd_into(gates, cbuf, hcat, pre, whf, whr, sm, gfmt=0, hq16=False, hq8=False):
    nt, L = _ntile(sm), sm.L
    act, cs, hs = _recur_fwd(pre, whf, whr, hq16, hq8)
    v = _valid(sm)
    if gfmt:
        gates_put_u16(gates, act.reshape(nt * 32, L, 2 * G4), nt, L)     # (padded slots: whatever the recurrence made of them)
    else:
        bl_put(gates, (act.reshape(nt * 32, L, 2 * G4)) * v, nt, L, 2 * G4)
    bl_put(cbuf, cs.reshape(nt * 32, L, 2 * H) * v, nt, L, 2 * H)
    bl_put(hcat, hs.reshape(nt * 32, L, 2 * H) * v, nt, L, 2 * H)


def _whh_from_pack(wpack):
    W = wpack.reshape(-1)[: 2 * G4 * H].reshape(2, G4, H)          # emu_dev.lstm_pack stores the raw weights
    return W[0], W[1]


def make_lstm_fwd(plain_fwd):
    def lstm_fwd(gates, cbuf, hcat, wpack, sm, mode=3, run_if=None, gfmt=0, gates_in=None):
        if _skip(run_if):
            return
        if mode not in (4, 5):
            return plain_fwd(gates, cbuf, hcat, wpack, sm, mode)
        nt, L = _ntile(sm), sm.L
        whf, whr = _whh_from_pack(wpack)
        src = gates_in if gfmt else gates
        _fwd_into(gates, cbuf, hcat, bl_get(src, nt, L, 2 * G4).reshape(nt * 32, L, 2, G4), whf, whr, sm, gfmt)
    return lstm_fwd


def make_lstm_bwd(plain_bwd):
    def lstm_bwd(gates, cbuf, hcat, dhcat, wpack, sm, mode=3, gfmt=0, dgates=None, run_if=None, amax=None, rfmt=0, dxn=None,
                 wxpack=None):
        if _skip(run_if):
            return
        if mode not in (4, 5):
            return plain_bwd(gates, cbuf, hcat, dhcat, wpack, sm, mode)
        whf, whr = _whh_from_pack(wpack)
        _bwd_into(gates, cbuf, dhcat, whf, whr, sm, gfmt, dgates, amax, rq=rfmt != 0)
        if dxn is not None:
            # ABI v19: d(xn) of each direction from the STORED scaled-fp16 d(gates) (the kernel's LDS image holds exactly
            # those values) x W_ih as fp16 hi + FP8 lo (lstm_pack_dx_f8), plain rows at the sequence map's positions
            assert rfmt == 2 and gfmt == 3 and mode == 4 and wxpack is not None and not getattr(sm, "nvalid", 0)
            nt, L = _ntile(sm), sm.L
            dg = _g16(dgates if dgates is not None else gates, nt, L, 2 * G4, 2, amax)[: sm.nseq].reshape(sm.nseq, L, 2, G4)
            wq = _PACKS[wxpack.data_ptr()]                         # [2, 4H, 128], quantised
            pos = _positions(sm)[: sm.nseq].reshape(-1)
            for di in (0, 1):
                dxn[di].reshape(-1, 128)[pos] = (dg[:, :, di] @ wq[di]).reshape(-1, 128)
    return lstm_bwd


def _bwd_into(gates, cbuf, dhcat, whf, whr, sm, gfmt=0, dgates=None, amax=None, rq=False):
    """gfmt 0: fp32 gates in, d(gates) in place; 1 (H2): unorm16 gates in, bf16 d(gates) in place; 2 (H2S): unorm16 gates
    in, d(gates) to `dgates` (fp32 here: the emulation does not model the split pair's 2^-17)."""
    nt, L = _ntile(sm), sm.L
    act = (gates_get_u16(gates, nt, L) if gfmt else bl_get(gates, nt, L, 2 * G4)).reshape(nt * 32, L, 2, G4)
    cs = bl_get(cbuf, nt, L, 2 * H).reshape(nt * 32, L, 2, H)
    dh = bl_get(dhcat, nt, L, 2 * H).reshape(nt * 32, L, 2, H)
    dpre = _recur_bwd(act, cs, dh, whf, whr, dgates_scale(amax) if gfmt == 3 else 1.0, rq).reshape(nt * 32, L, 2 * G4) * _valid(sm)
    if gfmt == 1:
        blh_put(dgates if dgates is not None else gates, dpre, nt, L, 2 * G4, torch.bfloat16)
    elif gfmt == 3:      # H2F: fp16(x * S), not clamped (ABI v17: overflow -> Inf -> the optimizer's guard)
        sc = dpre * dgates_scale(amax)
        blh_put(dgates if dgates is not None else gates, sc, nt, L, 2 * G4, torch.float16)
    else:
        bl_put(dgates if gfmt == 2 else gates, dpre, nt, L, 2 * G4)


def lstm_fwd_cluster(gates, cbuf, hcat, whh_f, whh_r, sm, status=None, dbg=0, gfmt=0, gates_in=None):
    """Returns the launch's timeout word like dev.lstm_fwd_cluster; dbg & 8 emulates a timeout: NaN-poisoned outputs
    and a set word, so the caller's predicated fall-back has to produce the result."""
    nt, L = _ntile(sm), sm.L
    if dbg & 8:
        for t in (gates, cbuf, hcat):
            t.fill_(float("nan"))
        return torch.ones(1, dtype=torch.int32)
    src = gates_in if gfmt else gates
    _fwd_into(gates, cbuf, hcat, bl_get(src, nt, L, 2 * G4).reshape(nt * 32, L, 2, G4), whh_f, whh_r, sm, gfmt)
    return torch.zeros(1, dtype=torch.int32)


def lstm_fwd_cluster2(gates, cbuf, hcat, xn, wcat, bcat, whh_f, whh_r, sm, status=None, dbg=0, dbg_buf=None, rfmt=None):
    """ws_lstm_fwd_cluster2: x-projection from the normalised input (BL(128)) inside the recurrence, fp16 h in the recurrent
    product, unorm16 gates out; dbg & 8 emulates a time-out like lstm_fwd_cluster.  rfmt 1 (ABI v20): the lo term of W_hh as
    e4m3 codes against e4m3 of h (one exponent per [32 rows][64 k] here, per wave on the device)."""
    nt, L = _ntile(sm), sm.L
    if dbg & 8:
        for t in (cbuf, hcat):
            t.fill_(float("nan"))
        return torch.ones(1, dtype=torch.int32)
    x = bl_get(xn, nt, L, 128)
    w, b = wcat.reshape(2, G4, 128), bcat.reshape(2, G4)
    pre = torch.stack([x @ w[0].t() + b[0], x @ w[1].t() + b[1]], 2)
    rf = real_dev.cluster2_rfmt() if rfmt is None else rfmt
    _fwd_into(gates, cbuf, hcat, pre, whh_f, whh_r, sm, 1, hq16=True, hq8=bool(rf))
    return torch.zeros(1, dtype=torch.int32)


def lstm_bwd_cluster(gates, cbuf, dhcat, whh_f, whh_r, sm, status=None, dbg=0):
    _bwd_into(gates, cbuf, dhcat, whh_f, whh_r, sm)


def _q8(W):
    """What rfmt 2 keeps of a recurrent weight: fp16 hi of 256 w + e4m3 codes of the remainder over a power-of-two scale that
    puts the largest possible remainder at 256 (per tensor here; the device scales per (direction, half, wave) block --
    ws_lstm_pack_pair_f8 -- or per group of 8 k-steps of it -- ws_lstm_pack_bwd_f8)."""
    s = 256.0 * W.detach()
    hi = s.half().float()
    m = float(s.abs().max())
    S = 2.0 ** (math.floor(math.log2(m)) + 1 - 20) if m > 0 and math.isfinite(m) else 1.0
    lo = ((s - hi) / S).to(torch.float8_e4m3fn).float() * S
    return (hi + lo) / 256.0


def lstm_pack_pair(whh_f, whh_r, pack, f16=False):
    q = _q8 if int(f16) == 2 else (lambda w: w)                                    # raw weights, like emu_dev.lstm_pack
    pack.reshape(-1)[: 2 * G4 * H] = torch.stack([q(whh_f), q(whh_r)]).reshape(-1)


def lstm_pack_bwd_f8(whh_f, whh_r, pack_bwd):
    pack_bwd.reshape(-1)[: 2 * G4 * H] = torch.stack([_q8(whh_f), _q8(whh_r)]).reshape(-1)


def lstm_pack_dx_f8(wcat, pack):
    w = wcat.reshape(2, G4, 128)
    _PACKS[pack.data_ptr()] = torch.stack([_q8(w[0]), _q8(w[1])])


def lstm_bwd_pair(gates, cbuf, dhcat, wpack, sm, status=None, dbg=0, dbg_buf=None, gfmt=0, dgates=None, repairable=False,
                  amax=None, rfmt=0):
    """Returns the launch's timeout word like dev.lstm_bwd_pair; dbg & 8 emulates the forced timeout (NaN-poisoned
    d(gates), both words set)."""
    if dbg & 8:
        (dgates if dgates is not None else gates).fill_(float("nan"))
        if status is not None:
            status.fill_(1)
        return torch.ones(1, dtype=torch.int32)
    _bwd_into(gates, cbuf, dhcat, *_whh_from_pack(wpack), sm, gfmt, dgates, amax, rq=rfmt != 0)
    return torch.zeros(1, dtype=torch.int32)


def lstm_fwd_fused(gates, cbuf, hcat, xn, wpack, bias, sm, gfmt=0, hfmt=0):
    nt, L = _ntile(sm), sm.L
    wih_f, wih_r, whf, whr = _PACKS[wpack.data_ptr()]
    x = bl_get(xn, nt, L, 128)
    if _probe() & 16384:                               # (numerics probe: the band view's fused x-projection on fp16 xn)
        x = x.half().float()
    b = bias.reshape(2, G4)
    pre = torch.stack([x @ wih_f.t() + b[0], x @ wih_r.t() + b[1]], 2)
    _fwd_into(gates, cbuf, hcat, pre, whf, whr, sm, gfmt, hq16=bool(hfmt & 1), hq8=bool(hfmt & 4))   # (hfmt 1: fp16 h in the recurrent product; 5: + FP8 lo term)


def gemm_tnb(*, G, g_width, g_off, g_cols, A0, a0_width, a0_off, a0_cols, nblk, L_, slab, nsplit, blocks_per_split,
             a0_shift=0, A1=None, a1_width=0, a1_off=0, a1_cols=0, a1_shift=0, bslab=None, aslab=None, g_fmt=0, amax=None,
             a_fmt=0):
    nt = nblk // L_
    assert a_fmt == 0 or (g_fmt == 2 and aslab is None)

    def shifted(buf, width, off, cols, shift):
        x = (blh_get(buf, nt, L_, width, torch.float16).float() if a_fmt else bl_get(buf, nt, L_, width))[:, :, off:off + cols]
        out = torch.zeros_like(x)
        if shift == 0:
            return x
        if shift > 0:                      # Acat(b) = A(b + shift)
            out[:, : L_ - shift] = x[:, shift:]
        else:
            out[:, -shift:] = x[:, : L_ + shift]
        return out
    g = (_g16(G, nt, L_, g_width, g_fmt, amax) if g_fmt else bl_get(G, nt, L_, g_width))[:, :, g_off:g_off + g_cols]
    a = shifted(A0, a0_width, a0_off, a0_cols, a0_shift)
    if A1 is not None:
        a = torch.cat([a, shifted(A1, a1_width, a1_off, a1_cols, a1_shift)], 2)
    acols = a.shape[2]
    slab.reshape(-1)[: nsplit * g_cols * acols] = 0.0
    slab.reshape(-1)[: g_cols * acols] = torch.einsum("slg,sla->ga", g, a).reshape(-1)
    if bslab is not None:
        bslab.reshape(-1)[: nsplit * g_cols] = 0.0
        bslab.reshape(-1)[:g_cols] = g.sum((0, 1))
    if aslab is not None:
        aslab.reshape(-1)[: nsplit * acols] = 0.0
        aslab.reshape(-1)[:acols] = a.sum((0, 1))


# ---- GroupNorm(1, C) pieces on the general group geometry of norm.hip (no per-band widths) -------------------------
def _group_index(geo):
    """[ngroups, L, W] element offsets: base(g) = (g / gdiv) * gs1 + (g % gdiv) * gs2, row stride rs."""
    assert geo.band_w is None and geo.nbands == 1
    g = torch.arange(geo.ngroups).view(-1, 1, 1)
    return (g // geo.gdiv) * geo.gs1 + (g % geo.gdiv) * geo.gs2 + torch.arange(geo.L).view(1, -1, 1) * geo.rs + \
        torch.arange(geo.W).view(1, 1, -1)


def make_group_stats(row_stats):
    def group_stats(x, geo, stats, eps=1.1920928955078125e-07):
        if geo.L == 1 and geo.gdiv == 1 and geo.gs2 == 0:
            return row_stats(x, geo, stats, eps)
        v = x.reshape(-1)[_group_index(geo)].reshape(geo.ngroups, -1)
        mean = v.mean(1)
        stats.reshape(-1, 2)[:, 0] = mean
        stats.reshape(-1, 2)[:, 1] = 1.0 / torch.sqrt(((v - mean.unsqueeze(1)) ** 2).mean(1) + eps)
    return group_stats


def make_gn_bwd_reduce(row_reduce):
    def gn_bwd_reduce(x, dxn, stats, geo, ab, gamma=None, gamma_tab=None):
        if geo.L == 1 and geo.gdiv == 1:
            return row_reduce(x, dxn, stats, geo, ab, gamma, gamma_tab)
        assert gamma_tab is None
        idx = _group_index(geo)
        st = stats.reshape(-1, 2)
        xh = (x.reshape(-1)[idx] - st[:, 0].view(-1, 1, 1)) * st[:, 1].view(-1, 1, 1)
        dg = dxn.reshape(-1)[idx] * gamma.reshape(-1)[: geo.W]
        ab.reshape(-1, 2)[:, 0] = dg.reshape(geo.ngroups, -1).mean(1)
        ab.reshape(-1, 2)[:, 1] = (dg * xh).reshape(geo.ngroups, -1).mean(1)
    return gn_bwd_reduce


def gn_bwd_apply(x, dxn, stats, ab, geo, dx, gamma=None, gamma_tab=None, res=None):
    assert gamma_tab is None
    idx = _group_index(geo)
    st, a = stats.reshape(-1, 2), ab.reshape(-1, 2)
    xh = (x.reshape(-1)[idx] - st[:, 0].view(-1, 1, 1)) * st[:, 1].view(-1, 1, 1)
    out = st[:, 1].view(-1, 1, 1) * (dxn.reshape(-1)[idx] * gamma.reshape(-1)[: geo.W] - a[:, 0].view(-1, 1, 1) -
                                     xh * a[:, 1].view(-1, 1, 1))
    if res is not None:
        out = out + res.reshape(-1)[idx]
    dx.reshape(-1)[idx] = out


def gn_param_grad(x, dxn, stats, geo, nsplit, slab):
    idx = _group_index(geo)
    st = stats.reshape(-1, 2)
    xh = (x.reshape(-1)[idx] - st[:, 0].view(-1, 1, 1)) * st[:, 1].view(-1, 1, 1)
    d = dxn.reshape(-1)[idx]
    slab.reshape(-1)[: nsplit * 2 * geo.W] = 0.0
    slab.reshape(-1)[: geo.W] = (d * xh).sum((0, 1))            # d gamma
    slab.reshape(-1)[geo.W: 2 * geo.W] = d.sum((0, 1))          # d beta


def gn_bwd_apply_pg(x, dxn, stats, ab, geo, dx, gamma, pslab, pout, counter, res=None):
    gn_bwd_apply(x, dxn, stats, ab, geo, dx, gamma=gamma, res=res)
    gn_param_grad(x, dxn, stats, geo, 1, pout)


def gn_bwd_fused(x, dxn, stats, geo, gamma, dx, nwg, pslab, res=None, pout=None, counter=None, dxn2=None):
    """norm.hip gn_bwd_fused_kernel: reduce + apply + parameter sums in one call (pslab [nwg, 2, 128]); dxn2: second addend."""
    if dxn2 is not None:
        dxn = dxn + dxn2
    ab = torch.zeros(geo.ngroups, 2)
    make_gn_bwd_reduce(None)(x, dxn, stats, geo, ab, gamma=gamma)
    gn_bwd_apply(x, dxn, stats, ab, geo, dx, gamma=gamma, res=res)
    gn_param_grad(x, dxn, stats, geo, nwg, pslab)
    if pout is not None:                               # the last workgroup's sum over the per-workgroup shares
        pout.reshape(-1)[: 2 * geo.W] = pslab.reshape(-1, 2 * geo.W)[:nwg].sum(0)      # (the rows behind: the kernel's group sums)


def install(monkeypatch):
    """After emu_dev.install: adds the BL entry points (and BL modes of lstm_fwd / lstm_bwd)."""
    import wesep_amd.dev as dev
    monkeypatch.setattr(dev, "lstm_fwd", make_lstm_fwd(dev.lstm_fwd))
    monkeypatch.setattr(dev, "lstm_bwd", make_lstm_bwd(dev.lstm_bwd))
    for fn in (pack_w, lstm_cat_ih, lstm_pack_fused, gemm_p2b, gemm_b2p, lstm_fwd_cluster, lstm_fwd_cluster2, lstm_bwd_cluster,
               lstm_pack_pair, lstm_pack_bwd_f8, lstm_pack_dx_f8, lstm_bwd_pair, lstm_fwd_fused, gemm_tnb):
        monkeypatch.setattr(dev, fn.__name__, fn)
    monkeypatch.setattr(dev, "group_stats", make_group_stats(dev.group_stats))
    monkeypatch.setattr(dev, "gn_bwd_reduce", make_gn_bwd_reduce(dev.gn_bwd_reduce))
    monkeypatch.setattr(dev, "gn_bwd_apply", gn_bwd_apply)
    monkeypatch.setattr(dev, "gn_param_grad", gn_param_grad)
    monkeypatch.setattr(dev, "gn_bwd_fused", gn_bwd_fused)
    monkeypatch.setattr(dev, "gn_bwd_apply_pg", gn_bwd_apply_pg)
    monkeypatch.setattr(dev, "cu_count", lambda device: 256)
    _PACKS.clear()
