"""TEST INFRASTRUCTURE ONLY -- torch-CPU emulation of the `wesep_amd.dev` entry points used by the composed model
paths, so that the HOST LOGIC (shapes, row addressing, weight re-layouts, autograd wiring) of those paths can be
checked against the oracle on a machine without a GPU.  It is installed by the `emulated_dev` fixture with
monkeypatch and never imported by the product: wesep_amd itself has no CPU path (a CPU tensor raises).  The kernels
themselves are only ever validated on the GPU (`-m gpu`) against torch and the oracle."""
import torch
import torch.nn.functional as F


def _rows_off(M, rows):
    div, s1, s2 = rows
    m = torch.arange(M, dtype=torch.long)
    return (m // div) * s1 + (m % div) * s2


def _gather(t, off, M, rows, K):
    flat_ = t.reshape(-1)
    idx = (off + _rows_off(M, rows)).unsqueeze(1) + torch.arange(K).unsqueeze(0)
    return flat_[idx]


def _descs(groups, ngroups, kind):
    import numpy as np
    from wesep_amd import _lib as L
    dt = L.GROUP_NT_DTYPE if kind == "nt" else L.GROUP_TN_DTYPE
    return np.frombuffer(groups.cpu().numpy().tobytes(), dtype=dt)[:ngroups]


def _at(ptr, n):
    """n floats at a raw host address taken from a CPU tensor's data_ptr() (descriptor tables carry pointers)."""
    import ctypes
    import numpy as np
    return torch.from_numpy(np.ctypeslib.as_array((ctypes.c_float * n).from_address(int(ptr))))


def _conv_matrix(x, M, conv):
    """The implicit patch matrix of dev.ConvView, materialised: [M, k*k*C]."""
    mode, H, W, C, Ho, Wo, k, sh, sw, p = conv[:10]
    dil = conv[10] if len(conv) > 10 and conv[10] else 1
    ldp = conv[11] if len(conv) > 11 and conv[11] else C            # pixel stride: the image is the first C of ldp columns
    R = M // (Ho * Wo)
    img = x.reshape(-1)[:R * H * W * ldp].reshape(R, H, W, ldp)[..., :C]
    out = torch.zeros(R, Ho, Wo, k * k, C)

    def src(o, n_out, tap, s, n_in):
        i = torch.arange(n_out)
        if mode == 0:
            v = i * s + tap * dil - p
            ok = (v >= 0) & (v < n_in)
        else:
            q = i + p - tap * dil
            v = torch.div(q, s, rounding_mode="floor")
            ok = (q >= 0) & (q % s == 0) & (v < n_in)
        return v.clamp(0, n_in - 1), ok

    for ky in range(k):
        hh, okh = src(None, Ho, ky, sh, H)
        for kx in range(k):
            ww, okw = src(None, Wo, kx, sw, W)
            g = img[:, hh][:, :, ww]                                    # [R, Ho, Wo, C]
            out[:, :, :, ky * k + kx] = g * (okh[:, None] & okw[None, :])[None, :, :, None]
    return out.reshape(M, k * k * C)


def gemm_nt(*, A, a_rows, M, C_out, c_rows, N=0, K=0, W=None, ldw=0, bias=None, R=None, T=None, stats=None,
            gamma=None, beta=None, stat_map=None, act=0, groups=None, ngroups=0, max_n=0, vec=3, a_off=0, c_off=0,
            w_off=0, mode=None, conv=None):
    if conv is not None:
        assert groups is None and stats is None and a_off == 0
        A, a_rows = _conv_matrix(A, M, conv), (1 << 30, 0, K)
        conv = None
    if groups is not None:   # grouped launch (per-group descriptors with raw pointers, _lib.GROUP_NT_DTYPE)
        assert stats is None and R is None and T is None and W is None
        for gd in _descs(groups, ngroups, "nt"):
            Kg, Ng = int(gd["K"]), int(gd["N"])
            gemm_nt(A=A, a_rows=a_rows, M=M, C_out=C_out, c_rows=c_rows, N=Ng, K=Kg,
                    W=_at(gd["W"], (Kg if vec & 8 else Ng) * int(gd["ldw"])),
                    ldw=int(gd["ldw"]), bias=_at(gd["bias"], Ng) if gd["bias"] else None, act=act,
                    a_off=a_off + int(gd["a_off"]), c_off=c_off + int(gd["c_off"]), vec=vec)
        return
    a = _gather(A, a_off, M, a_rows, K)
    if stats is not None:
        d1, m1, d2, m2, base = stat_map
        m = torch.arange(M)
        s = (m // d1) * m1 + (m % d2) * m2 + base
        st = stats.reshape(-1, 2)
        a = (a - st[s, 0:1]) * st[s, 1:2] * gamma.reshape(-1)[:K] + beta.reshape(-1)[:K]
    if vec & 8:          # W stored transposed: W'[n][k] = W[k * ldw + n] (wesep_hip.h ws_gemm_nt_args.vec bit 3)
        assert N % 4 == 0 and ldw % 4 == 0 and stats is None
        w = W.reshape(-1)[w_off + (torch.arange(N).unsqueeze(1) + torch.arange(K).unsqueeze(0) * ldw)]
    else:
        w = W.reshape(-1)[w_off + (torch.arange(N).unsqueeze(1) * ldw + torch.arange(K).unsqueeze(0))]
    v = a @ w.t()
    if bias is not None:
        v = v + bias.reshape(-1)[:N]
    if act == 1:
        v = torch.tanh(v)
    if act == 2:
        v = torch.relu(v)
    cidx = (c_off + _rows_off(M, c_rows)).unsqueeze(1) + torch.arange(N).unsqueeze(0)
    if T is not None:
        t = T.reshape(-1)[cidx]
        v = v * ((t > 0).float() if act == 4 else (1 - t * t))
    if R is not None:
        v = v + R.reshape(-1)[cidx]
    C_out.reshape(-1)[cidx] = v


def gemm_tn(*, G, g_rows, A, a_rows, M, slab, slab_stride, nsplit, rows_per_split, Nn=0, Kk=0, bslab=None,
            bslab_stride=0, out_off=0, bout_off=0, stats=None, gamma=None, beta=None, stat_map=None, shift_rows=0,
            seq_div=1, seq_len=1, groups=None, ngroups=0, max_n=0, max_k=0, vec=1, g_off=0, a_off=0, mode=None,
            conv=None):
    if conv is not None:
        assert conv[0] == 0 and groups is None and stats is None and not shift_rows and a_off == 0
        A, a_rows = _conv_matrix(A, M, conv), (1 << 30, 0, Kk)
    if groups is not None:   # grouped launch (_lib.GROUP_TN_DTYPE): one output block per group in every split's slab
        assert stats is None and not shift_rows and bslab is None
        for gd in _descs(groups, ngroups, "tn"):
            gemm_tn(G=G, g_rows=g_rows, A=A, a_rows=a_rows, M=M, slab=slab, slab_stride=slab_stride, nsplit=nsplit,
                    rows_per_split=rows_per_split, Nn=int(gd["Nn"]), Kk=int(gd["Kk"]), out_off=out_off + int(gd["out_off"]),
                    g_off=g_off + int(gd["g_off"]), a_off=a_off + int(gd["a_off"]))
        return
    g = _gather(G, g_off, M, g_rows, Nn)
    if shift_rows:      # m' = m + shift_rows, zeroed when the step index ((m // seq_div) % seq_len) leaves the sequence
        m = torch.arange(M)
        step = (m // seq_div) % seq_len + (1 if shift_rows > 0 else -1)
        ok = (step >= 0) & (step < seq_len)
        mm = torch.where(ok, m + shift_rows, m)
        div, s1, s2 = a_rows
        off = a_off + (mm // div) * s1 + (mm % div) * s2
        a = A.reshape(-1)[off.unsqueeze(1) + torch.arange(Kk).unsqueeze(0)] * ok.unsqueeze(1)
    else:
        a = _gather(A, a_off, M, a_rows, Kk)
    if stats is not None:
        d1, m1, d2, m2, base = stat_map
        m = torch.arange(M)
        s = (m // d1) * m1 + (m % d2) * m2 + base
        st = stats.reshape(-1, 2)
        a = (a - st[s, 0:1]) * st[s, 1:2] * gamma.reshape(-1)[:Kk] + beta.reshape(-1)[:Kk]
    sl = slab.reshape(-1)
    for sp in range(nsplit):
        lo, hi = sp * rows_per_split, min(M, (sp + 1) * rows_per_split)
        blk = (g[lo:hi].t() @ a[lo:hi]).reshape(-1) if hi > lo else torch.zeros(Nn * Kk)
        sl[sp * slab_stride + out_off: sp * slab_stride + out_off + Nn * Kk] = blk
        if bslab is not None:
            bslab.reshape(-1)[sp * bslab_stride + bout_off: sp * bslab_stride + bout_off + Nn] = \
                g[lo:hi].sum(0) if hi > lo else 0.0


def reduce_slabs(slab, nsplit, stride, count, out, w=0, ldo=0, out_off=0):
    s = slab.reshape(-1)
    tot = sum(s[k * stride: k * stride + count] for k in range(nsplit))
    i = torch.arange(count)
    o = (i // w) * ldo + (i % w) if w > 0 else i
    out.reshape(-1)[out_off + o] = tot


def transpose(src, rows, cols, lds, dst, src_off=0, dst_off=0):
    idx = src_off + torch.arange(rows).unsqueeze(1) * lds + torch.arange(cols).unsqueeze(0)
    dst.reshape(-1)[dst_off: dst_off + rows * cols] = src.reshape(-1)[idx].t().reshape(-1)


def affine_fwd(z, a, b, a0, rows, rows_per_r, N, out):
    zz = z.reshape(-1)[: rows * N].reshape(rows, N)
    r = torch.arange(rows) // rows_per_r
    sc = a0 + (a.reshape(-1, N)[r] if a is not None else 0.0)
    res = zz * sc + (b.reshape(-1, N)[r] if b is not None else 0.0)
    out.reshape(-1)[: rows * N] = res.reshape(-1)


def chan_sums(g, x, stats, st_div, rows_per_group, ngroups, Cc):
    gg = g.reshape(ngroups, rows_per_group, Cc)
    out = torch.zeros(ngroups, 2, Cc)
    out[:, 0] = gg.sum(1)
    if x is not None:
        xx = x.reshape(ngroups * rows_per_group, Cc)
        if stats is not None:
            s = torch.arange(ngroups * rows_per_group) // st_div
            st = stats.reshape(-1, 2)
            xx = (xx - st[s, 0:1]) * st[s, 1:2]
        out[:, 1] = (gg * xx.reshape(ngroups, rows_per_group, Cc)).sum(1)
    return out


def _nchw(x, B, H, W):
    return x.reshape(B, H, W, -1).permute(0, 3, 1, 2)


def _cl(y):
    B, C, H, W = y.shape
    return y.permute(0, 2, 3, 1).reshape(B * H * W, C)


def im2col_hw(x, R, H, W, Cc, k, sh, sw, p, patches, ldp):
    u = F.unfold(_nchw(x.reshape(R * H * W, Cc), R, H, W), k, padding=p, stride=(sh, sw))   # [R, C*k*k, L]
    L = u.shape[-1]
    u = u.reshape(R, Cc, k * k, L).permute(0, 3, 2, 1).reshape(R * L, k * k * Cc)           # tap-major, channel-minor
    patches.reshape(R * L, ldp)[:, : k * k * Cc] = u


def col2im_hw(dpatches, R, H, W, Cc, k, sh, sw, p, dx):
    Ho, Wo = (H + 2 * p - k) // sh + 1, (W + 2 * p - k) // sw + 1
    u = dpatches.reshape(R, Ho * Wo, k * k, Cc).permute(0, 3, 2, 1).reshape(R, Cc * k * k, Ho * Wo)
    y = F.fold(u, (H, W), k, padding=p, stride=(sh, sw))
    dx.reshape(R * H * W, Cc)[:] = _cl(y)


def elu_fwd(x, y):
    y.copy_(F.elu(x))


def elu_bwd(x, dy, dx):
    dx.copy_(torch.where(x > 0, dy, dy * torch.exp(x)))


def inorm_fwd(x, G, P, Cc, y, eps=1e-5):
    xx = x.reshape(G, P, Cc)
    mean = xx.mean(1)
    var = (xx * xx).mean(1) - mean * mean
    rstd = 1.0 / torch.sqrt(var.clamp_min(0) + eps)
    y.reshape(G, P, Cc)[:] = (xx - mean.unsqueeze(1)) * rstd.unsqueeze(1)
    return torch.stack([mean, rstd], 1).contiguous()


def inorm_bwd(y, dy, stats, G, P, Cc, dx):
    yy, dd = y.reshape(G, P, Cc), dy.reshape(G, P, Cc)
    s0, s1 = dd.mean(1, keepdim=True), (dd * yy).mean(1, keepdim=True)
    dx.reshape(G, P, Cc)[:] = stats[:, 1].unsqueeze(1) * (dd - s0 - yy * s1)


def _dw(xr, w, b, P, dil, Cc, causal):
    if causal:          # convs.py:61-62,91-92: pad dil*(P-1) on both sides, cut the tail
        pad = dil * (P - 1)
        return F.conv1d(xr, w, b, padding=pad, dilation=dil, groups=Cc)[:, :, :xr.shape[2]]
    return F.conv1d(xr, w, b, padding=dil * (P - 1) // 2, dilation=dil, groups=Cc)


def dwconv_fwd(x, stats, gamma, beta, w, b, R, Tp, Cc, P, dil, st_div, y, causal=False):
    s = torch.arange(R * Tp) // st_div
    st = stats.reshape(-1, 2)
    xn = (x.reshape(R * Tp, Cc) - st[s, 0:1]) * st[s, 1:2] * gamma + beta
    xr = xn.reshape(R, Tp, Cc).permute(0, 2, 1)
    o = _dw(xr, w.reshape(Cc, 1, P), b, P, dil, Cc, causal)
    y.reshape(R, Tp, Cc)[:] = o.permute(0, 2, 1)


def dwconv_bwd(dy, x, stats, gamma, beta, w, R, Tp, Cc, P, dil, st_div, dxn, causal=False):
    s = torch.arange(R * Tp) // st_div
    st = stats.reshape(-1, 2)
    xn = ((x.reshape(R * Tp, Cc) - st[s, 0:1]) * st[s, 1:2] * gamma + beta).reshape(R, Tp, Cc).permute(0, 2, 1)
    xn = xn.detach().requires_grad_(True)
    wr = w.reshape(Cc, 1, P).detach().requires_grad_(True)
    br = torch.zeros(Cc, requires_grad=True)
    with torch.enable_grad():
        o = _dw(xn, wr, br, P, dil, Cc, causal)
        o.backward(dy.reshape(R, Tp, Cc).permute(0, 2, 1))
    dxn.reshape(R, Tp, Cc)[:] = xn.grad.permute(0, 2, 1)
    return wr.grad.reshape(Cc, P).contiguous(), br.grad.contiguous()


def avgpool_fwd(x, B, H, W, Cc, sz, y):
    y.reshape(-1, Cc)[:] = _cl(F.avg_pool2d(_nchw(x, B, H, W), sz))


def avgpool_bwd(dy, B, H, W, Cc, sz, dx):
    xr = torch.zeros(B, Cc, H, W, requires_grad=True)
    with torch.enable_grad():
        F.avg_pool2d(xr, sz).backward(_nchw(dy, B, H // sz, W // sz))
    dx.reshape(-1, Cc)[:] = _cl(xr.grad)


def bilinear_fwd(x, B, h, w, H, W, Cc, y):
    y.reshape(-1, Cc)[:] = _cl(F.interpolate(_nchw(x, B, h, w), size=(H, W), mode="bilinear"))


def bilinear_bwd(dy, B, h, w, H, W, Cc, dx):
    xr = torch.zeros(B, Cc, h, w, requires_grad=True)
    with torch.enable_grad():
        F.interpolate(xr, size=(H, W), mode="bilinear").backward(_nchw(dy, B, H, W))
    dx.reshape(-1, Cc)[:] = _cl(xr.grad)


def scale_bf_fwd(x, s, B, T, Fq, Cc, mode, y):
    xx = x.reshape(B, T, Fq, Cc)
    sv = s.reshape(B, 1, Fq, 1)
    y.reshape(B, T, Fq, Cc)[:] = xx * sv if mode == 0 else xx + sv


def scale_bf_bwd(x, dy, s, B, T, Fq, Cc, mode, dx, ds):
    xx, dd = x.reshape(B, T, Fq, Cc), dy.reshape(B, T, Fq, Cc)
    sv = s.reshape(B, 1, Fq, 1)
    dx.reshape(B, T, Fq, Cc)[:] = dd * sv if mode == 0 else dd
    ds.reshape(B, Fq)[:] = (dd * xx).sum((1, 3)) if mode == 0 else dd.sum((1, 3))


def preemph_pad(x, R, T, pad, ldo, coef, out):
    y = x.clone()
    y[:, 1:] = x[:, 1:] - coef * x[:, :-1]
    y[:, 0] = x[:, 0] - coef * x[:, 1]
    out[:, : T + 2 * pad] = F.pad(y.unsqueeze(1), (pad, pad), "reflect").squeeze(1)


def ola_fwd(frames, bias, R, Tp, Lk, hop, Tout, est):
    full = torch.zeros(R, (Tp - 1) * hop + Lk)
    fr = frames.reshape(R, Tp, Lk)
    for t in range(Tp):
        full[:, t * hop: t * hop + Lk] += fr[:, t]
    est[:, :Tout] = full[:, :Tout] + (bias.reshape(-1)[0] if bias is not None else 0.0)


def ola_bwd(dest, R, Tp, Lk, hop, Tout, dframes):
    full = torch.zeros(R, (Tp - 1) * hop + Lk)
    full[:, :Tout] = dest.reshape(R, -1)[:, :Tout]
    dframes.reshape(R, Tp, Lk)[:] = full.unfold(1, Lk, hop)


def total_sum(x):
    return x.sum().reshape(1)


# ---- recurrence (plain row layout, hidden 256; include/wesep_hip.h ws_lstm_*) ------------------------------------
def lstm_pack(whh_f, whh_r, pack_fwd, pack_bwd, mode=3):
    raw = torch.cat([whh_f.reshape(-1), whh_r.reshape(-1)])
    pack_fwd.reshape(-1)[: raw.numel()] = raw
    pack_bwd.reshape(-1)[: raw.numel()] = raw


def _lstm_rows(sm, t):
    s = torch.arange(sm.nseq)
    return (s // sm.div) * sm.s1 + (s % sm.div) * sm.s2 + t * sm.step_rows


def lstm_fwd(gates, cbuf, hcat, wpack, sm, mode=3):
    H = 256
    W = wpack.reshape(-1)[: 2 * 4 * H * H].reshape(2, 4 * H, H)
    g2 = gates.reshape(-1, 2, 4 * H)
    c2, h2 = cbuf.reshape(-1, 2, H), hcat.reshape(-1, 2, H)
    for d in (0, 1):
        h = torch.zeros(sm.nseq, H)
        c = torch.zeros(sm.nseq, H)
        order = range(sm.L) if d == 0 else range(sm.L - 1, -1, -1)
        for t in order:
            r = _lstm_rows(sm, t)
            pre = g2[r, d] + h @ W[d].t()
            i, f, g, o = pre[:, :H].sigmoid(), pre[:, H:2 * H].sigmoid(), pre[:, 2 * H:3 * H].tanh(), pre[:, 3 * H:].sigmoid()
            c = f * c + i * g
            h = o * c.tanh()
            g2[r, d] = torch.cat([i, f, g, o], 1)
            c2[r, d] = c
            h2[r, d] = h


def lstm_bwd(gates, cbuf, hcat, dhcat, wpack, sm, mode=3):
    H = 256
    W = wpack.reshape(-1)[: 2 * 4 * H * H].reshape(2, 4 * H, H)
    g2 = gates.reshape(-1, 2, 4 * H)
    c2, dh2 = cbuf.reshape(-1, 2, H), dhcat.reshape(-1, 2, H)
    for d in (0, 1):
        dh_rec = torch.zeros(sm.nseq, H)
        dc = torch.zeros(sm.nseq, H)
        order = list(range(sm.L)) if d == 0 else list(range(sm.L - 1, -1, -1))
        for idx in range(sm.L - 1, -1, -1):
            t = order[idx]
            r = _lstm_rows(sm, t)
            a = g2[r, d]
            i, f, g, o = a[:, :H], a[:, H:2 * H], a[:, 2 * H:3 * H], a[:, 3 * H:]
            c = c2[r, d]
            cprev = c2[_lstm_rows(sm, order[idx - 1]), d] if idx > 0 else torch.zeros_like(c)
            dh = dh2[r, d] + dh_rec
            tc = c.tanh()
            dcv = dc + dh * o * (1 - tc * tc)
            dpre = torch.cat([dcv * g * i * (1 - i), dcv * cprev * f * (1 - f), dcv * i * (1 - g * g),
                              dh * tc * o * (1 - o)], 1)
            dc = dcv * f
            dh_rec = dpre @ W[d]
            g2[r, d] = dpre


# ---- norms / activations of tasnet.hip, norm.hip ---------------------------------------------------------------
def group_stats(x, geo, stats, eps=1.19e-7):
    assert geo.L == 1 and geo.gdiv == 1 and geo.gs2 == 0        # the row-wise (cLN) geometry only
    xx = x.reshape(-1)[: geo.ngroups * geo.gs1].reshape(geo.ngroups, geo.gs1)[:, : geo.W]
    mean = xx.mean(1)
    var = ((xx - mean.unsqueeze(1)) ** 2).mean(1)
    stats.reshape(-1, 2)[:, 0] = mean
    stats.reshape(-1, 2)[:, 1] = 1.0 / torch.sqrt(var + eps)


def flat_stats(x, ngroups, n_per_group, stats, eps=1e-5):
    xx = x.reshape(ngroups, n_per_group)
    mean = xx.mean(1)
    var = ((xx - mean.unsqueeze(1)) ** 2).mean(1)
    stats.reshape(-1, 2)[:, 0] = mean
    stats.reshape(-1, 2)[:, 1] = 1.0 / torch.sqrt(var + eps)


def gn_bwd_reduce(x, dxn, stats, geo, ab, gamma=None, gamma_tab=None):
    assert geo.L == 1 and geo.gdiv == 1 and gamma_tab is None
    M, Wd = geo.ngroups, geo.W
    st = stats.reshape(-1, 2)
    xh = (x.reshape(M, Wd) - st[:, 0:1]) * st[:, 1:2]
    dg = dxn.reshape(M, Wd) * gamma.reshape(-1)
    ab.reshape(-1, 2)[:, 0] = dg.mean(1)
    ab.reshape(-1, 2)[:, 1] = (dg * xh).mean(1)


def norm_ab(sums, gamma, ngroups, Cc, n_per_group, ab):
    s = sums.reshape(ngroups, 2, Cc)
    ab.reshape(-1, 2)[:, 0] = (s[:, 0] * gamma).sum(1) / n_per_group
    ab.reshape(-1, 2)[:, 1] = (s[:, 1] * gamma).sum(1) / n_per_group


def norm_bwd_apply_cl(x, dxn, stats, ab, gamma, res, rows, Cc, st_div, dx):
    s = torch.arange(rows) // st_div
    st, a = stats.reshape(-1, 2), ab.reshape(-1, 2)
    xh = (x.reshape(rows, Cc) - st[s, 0:1]) * st[s, 1:2]
    r = (dxn.reshape(rows, Cc) * gamma.reshape(-1) - a[s, 0:1] - xh * a[s, 1:2]) * st[s, 1:2]
    if res is not None:
        r = r + res.reshape(rows, Cc)
    dx.reshape(rows, Cc)[:] = r


def prelu_fwd(x, rb, a, rows, Cc, rows_per_r, y):
    v = x.reshape(rows, Cc)
    if rb is not None:
        v = v + rb.reshape(-1, Cc)[torch.arange(rows) // rows_per_r]
        x.reshape(rows, Cc)[:] = v
    y.reshape(rows, Cc)[:] = torch.where(v > 0, v, a.reshape(-1)[0] * v)


def prelu_bwd(pre, dy, a, dx):
    da = (dy * torch.clamp(pre, max=0)).sum().reshape(1)
    dx.copy_(torch.where(pre > 0, dy, a.reshape(-1)[0] * dy))
    return da


def softmax_rows_fwd(x, rows, n, scale, y):
    y.reshape(rows, n)[:] = torch.softmax(scale * x.reshape(rows, n), 1)


def softmax_rows_bwd(y, dy, rows, n, scale, dx):
    yy, dd = y.reshape(rows, n), dy.reshape(rows, n)
    dx.reshape(rows, n)[:] = scale * yy * (dd - (dd * yy).sum(1, keepdim=True))


def conv3x3_pack(W2, Cin, Cout):
    return W2.reshape(Cout, 9 * Cin).contiguous()       # the emulation keeps the plain rows


def conv3x3_pack_srcs(srcs, Cin, Cout, flip=False):
    """ws_conv3x3_pack: W[n][tap][c] = w_k.flat[off + n * s_row + (c - col_off) * s_col + (8 - tap if flip else tap) * s_tap]."""
    W = torch.zeros(Cout, 9, Cin)
    n = torch.arange(Cout).view(-1, 1, 1)
    tap = torch.arange(9).view(1, -1, 1)
    for w, off, s_row, s_col, s_tap, col_off, cols in srcs:
        c = torch.arange(cols).view(1, 1, -1)
        W[:, :, col_off:col_off + cols] = w.reshape(-1)[off + n * s_row + c * s_col + ((8 - tap) if flip else tap) * s_tap]
    return W.reshape(Cout, 9 * Cin).contiguous()


def conv3x3(*, X, ldx, W, ldw, B, H, Wd, Cin, Cout, Y, ldy, bias=None, R=None, x_off=0, y_off=0):
    M = B * H * Wd
    img = X.reshape(-1)[:M * ldx].reshape(B, H, Wd, ldx)[..., x_off:x_off + Cin].permute(0, 3, 1, 2)
    w = W.reshape(-1)[:Cout * ldw].reshape(Cout, ldw)[:, :9 * Cin].reshape(Cout, 3, 3, Cin).permute(0, 3, 1, 2)
    v = F.conv2d(img, w, bias.reshape(-1)[:Cout] if bias is not None else None, padding=1).permute(0, 2, 3, 1).reshape(M, Cout)
    if R is not None:
        v = v + R.reshape(-1)[:M * ldy].reshape(M, ldy)[:, y_off:y_off + Cout]
    Y.reshape(-1)[:M * ldy].reshape(M, ldy)[:, y_off:y_off + Cout] = v


def conv3x3_wgrad_tiles(B, H, Wd):
    return B * (-(-H // 30)) * (-(-Wd // 4))


def conv3x3_wgrad(*, G, ldg, X, ldx, B, H, Wd, Cin, Nn, slab, nsplit, tiles_per_split, bslab=None, sw=1, Wx=0, g_off=0):
    """Slab 0 carries the whole gradient (the split of the pixels is a device detail), the others are zero."""
    assert nsplit * tiles_per_split >= conv3x3_wgrad_tiles(B, H, Wd)
    Wx = Wx or Wd
    assert sw in (1, 2) and (Wx - 1) // sw + 1 == Wd
    M, Mx = B * H * Wd, B * H * Wx
    img = X.reshape(-1)[:Mx * ldx].reshape(B, H, Wx, ldx)[..., :Cin].permute(0, 3, 1, 2)
    g = G.reshape(-1)[:M * ldg].reshape(B, H, Wd, ldg)[..., g_off:g_off + Nn].permute(0, 3, 1, 2)
    # dW[n][c][ky][kx] = sum g[b][n][h][w] * xpad[b][c][h + ky][sw*w + kx]
    dw = torch.nn.grad.conv2d_weight(img, (Nn, Cin, 3, 3), g, stride=(1, sw), padding=1)
    sl = slab.reshape(-1)[:nsplit * Nn * 9 * Cin].reshape(nsplit, Nn * 9 * Cin)
    sl.zero_()
    sl[0] = dw.permute(0, 2, 3, 1).reshape(-1)
    if bslab is not None:
        bs = bslab.reshape(-1)[:nsplit * Nn].reshape(nsplit, Nn)
        bs.zero_()
        bs[0] = g.sum((0, 2, 3))


IN_ELU_PRE, IN_ELU_POST = 1, 2


def _cols(t, rows, ld, off, Cc):
    return t.reshape(-1)[:rows * ld].reshape(rows, ld)[:, off:off + Cc] if ld else t.reshape(rows, Cc)


def in_act_fwd(x, G, P, Cc, flags, y, eps=1e-5, y_ld=0, y_off=0):
    u = x.reshape(G, P, Cc)
    if flags & 1:
        u = F.elu(u)
    mean = u.mean(1)
    var = (u * u).mean(1) - mean * mean
    rstd = 1.0 / torch.sqrt(var.clamp_min(0) + eps)
    n = (u - mean[:, None]) * rstd[:, None]
    _cols(y, G * P, y_ld, y_off, Cc)[:] = (F.elu(n) if flags & 2 else n).reshape(G * P, Cc)
    return torch.stack([mean, rstd], 1)


def in_act_bwd(x, dy, stats, G, P, Cc, flags, dx, dy_ld=0, dy_off=0, dx_ld=0, dx_off=0):
    xx, dd = x.reshape(G, P, Cc), _cols(dy, G * P, dy_ld, dy_off, Cc).reshape(G, P, Cc)
    u = F.elu(xx) if flags & 1 else xx
    mean, rstd = stats[:, 0][:, None], stats[:, 1][:, None]
    n = (u - mean) * rstd
    d = dd * torch.where(n > 0, torch.ones_like(n), torch.exp(n)) if flags & 2 else dd
    r = rstd * (d - d.mean(1, keepdim=True) - n * (d * n).mean(1, keepdim=True))
    if flags & 1:
        r = r * torch.where(xx > 0, torch.ones_like(xx), torch.exp(xx))
    _cols(dx, G * P, dx_ld, dx_off, Cc)[:] = r.reshape(G * P, Cc)


def heads_ok(Q, nh, ch):
    return ch % 4 == 0 and 0 < nh <= 8 and Q * nh * ch <= 9216


def _heads_norm(x, x_ld, x_off, slope, B, T, Q, nh, ch, eps=1e-5):
    xx = _cols(x, B * T * Q, x_ld, x_off, nh * ch).reshape(B, T, Q, nh, ch)
    u = torch.where(xx > 0, xx, slope.view(1, 1, 1, nh, 1) * xx)
    mean = u.mean((2, 4), keepdim=True)
    rstd = 1.0 / torch.sqrt(((u - mean) ** 2).mean((2, 4), keepdim=True) + eps)
    return xx, u, mean, rstd


def heads_fwd(x, x_ld, x_off, slope, gamma, beta, B, T, Tp, Q, nh, ch, y, stats):
    xx, u, mean, rstd = _heads_norm(x, x_ld, x_off, slope, B, T, Q, nh, ch)
    g = gamma.view(nh, Q, ch).permute(1, 0, 2)
    bt = beta.view(nh, Q, ch).permute(1, 0, 2)
    yy = (u - mean) * rstd * g + bt                                           # [B, T, Q, nh, ch]
    out = y.view(nh, B, Tp, Q * ch)
    out.zero_()
    out[:, :, :T] = yy.permute(3, 0, 1, 2, 4).reshape(nh, B, T, Q * ch)
    st = stats.view(nh, B * T, 2)
    st[..., 0] = mean.reshape(B * T, nh).t()
    st[..., 1] = rstd.reshape(B * T, nh).t()


def heads_bwd(x, x_ld, x_off, dy, slope, gamma, stats, B, T, Tp, Q, nh, ch, dx, dx_ld, dx_off):
    xx, u, _, _ = _heads_norm(x, x_ld, x_off, slope, B, T, Q, nh, ch)
    st = stats.view(nh, B, T, 2).permute(1, 2, 0, 3)                         # [B, T, nh, 2]
    mean, rstd = st[..., 0][:, :, None, :, None], st[..., 1][:, :, None, :, None]
    n = (u - mean) * rstd
    dd = dy.view(nh, B, Tp, Q, ch)[:, :, :T].permute(1, 2, 3, 0, 4)          # [B, T, Q, nh, ch]
    g = gamma.view(nh, Q, ch).permute(1, 0, 2)
    d = dd * g
    dl = rstd * (d - d.mean((2, 4), keepdim=True) - n * (d * n).mean((2, 4), keepdim=True))
    a = slope.view(1, 1, 1, nh, 1)
    _cols(dx, B * T * Q, dx_ld, dx_off, nh * ch)[:] = torch.where(xx > 0, dl, a * dl).reshape(B * T * Q, nh * ch)
    dg = (dd * n).sum((0, 1)).permute(1, 0, 2).reshape(nh, Q * ch)
    db = dd.sum((0, 1)).permute(1, 0, 2).reshape(nh, Q * ch)
    ds = torch.where(xx > 0, torch.zeros_like(dl), dl * xx).sum((0, 1, 2, 4))
    return dg, db, ds


def rowln_ok(W):
    return 0 < W <= 256 and W % 4 == 0


def rowln_fwd(x, gamma, beta, M, W, y, stats, eps=1e-5):
    xx = x.reshape(M, W)
    mean = xx.mean(1, keepdim=True)
    var = ((xx - mean) ** 2).mean(1, keepdim=True)
    rstd = 1.0 / torch.sqrt(var + eps)
    y.reshape(M, W)[:] = (xx - mean) * rstd * gamma.reshape(1, W) + beta.reshape(1, W)
    stats.reshape(M, 2)[:, 0] = mean[:, 0]
    stats.reshape(M, 2)[:, 1] = rstd[:, 0]


def rowln_bwd(x, dy, stats, gamma, M, W, dx, res=None):
    xx, dd, st = x.reshape(M, W), dy.reshape(M, W), stats.reshape(M, 2)
    xh = (xx - st[:, :1]) * st[:, 1:2]
    gd = dd * gamma.reshape(1, W)
    tot = torch.stack([dd.sum(0), (dd * xh).sum(0)])
    o = st[:, 1:2] * (gd - gd.mean(1, keepdim=True) - xh * (gd * xh).mean(1, keepdim=True))
    if res is not None:
        o = o + res.reshape(M, W)
    dx.reshape(M, W)[:] = o
    return tot


# ---- Conv-TasNet / speaker-encoder pieces (tasnet.hip, conv2d.hip) -----------------------------------------------
def maskmul_fwd(w, w_off, ldw, m, rows, N, s):
    wv = w.reshape(-1)[w_off + torch.arange(rows).unsqueeze(1) * ldw + torch.arange(N).unsqueeze(0)]
    s.reshape(rows, N)[:] = wv * m.reshape(rows, N)


def maskmul_bwd(ds, w, w_off, ldw, m, rows, N, dw, dw_off, ld_dw, dm):
    wv = w.reshape(-1)[w_off + torch.arange(rows).unsqueeze(1) * ldw + torch.arange(N).unsqueeze(0)]
    mm, dd = m.reshape(rows, N), ds.reshape(rows, N)
    dw.reshape(-1)[dw_off + torch.arange(rows).unsqueeze(1) * ld_dw + torch.arange(N).unsqueeze(0)] = dd * mm
    dm.reshape(rows, N)[:] = torch.where(mm > 0, dd * wv, torch.zeros(()))


def relu_mask(d, y):
    d.mul_((y > 0).float())


def bn_stats(x, M, Cc, running_mean, running_var, stats, eps=1e-5, momentum=0.1):
    if M <= 1:      # like dev.bn_stats / torch
        raise ValueError(f"BatchNorm in training mode needs more than 1 value per channel (got {M} row)")
    xx = x.reshape(M, Cc)
    mean = xx.mean(0)
    var = ((xx - mean) ** 2).mean(0)
    stats.reshape(2, Cc)[0] = mean
    stats.reshape(2, Cc)[1] = 1.0 / torch.sqrt(var + eps)
    if running_mean is not None:
        running_mean.mul_(1 - momentum).add_(momentum * mean)
        running_var.mul_(1 - momentum).add_(momentum * var * (M / (M - 1.0) if M > 1 else 1.0))


def bn_prelu_fwd(x, stats, gamma, beta, res, a, M, Cc, u, y):
    st = stats.reshape(2, Cc)
    v = (x.reshape(M, Cc) - st[0]) * st[1] * gamma + beta
    if res is not None:
        v = v + res.reshape(M, Cc)
    u.reshape(M, Cc)[:] = v
    y.reshape(M, Cc)[:] = torch.where(v > 0, v, a.reshape(-1)[0] * v)


def bn_bwd(x, du, stats, gamma, M, Cc, dx):
    st = stats.reshape(2, Cc)
    xh = (x.reshape(M, Cc) - st[0]) * st[1]
    dd = du.reshape(M, Cc)
    s0, s1 = dd.sum(0), (dd * xh).sum(0)
    dx.reshape(M, Cc)[:] = gamma * st[1] * (dd - s0 / M - xh * s1 / M)
    return torch.stack([s0, s1], 0).contiguous()


def maxpool3_fwd(x, R, T, Cc, y):
    y.reshape(R, T // 3, Cc)[:] = F.max_pool1d(x.reshape(R, T, Cc).permute(0, 2, 1), 3).permute(0, 2, 1)


def maxpool3_bwd(x, dy, R, T, Cc, dx):
    xr = x.reshape(R, T, Cc).permute(0, 2, 1).detach().clone().requires_grad_(True)
    with torch.enable_grad():
        F.max_pool1d(xr, 3).backward(dy.reshape(R, T // 3, Cc).permute(0, 2, 1))
    dx.reshape(R, T, Cc)[:] = xr.grad.permute(0, 2, 1)


def bcast_rows(src, scale, rows_per_r, M, Cc, out):
    out.reshape(M, Cc)[:] = scale * src.reshape(-1, Cc)[torch.arange(M) // rows_per_r]


def cross_entropy(logits, label, loss, dlogits):
    lr = logits.detach().clone().requires_grad_(True)
    with torch.enable_grad():
        v = F.cross_entropy(lr, label)
        v.backward()
    loss.reshape(-1)[0] = v.detach()
    dlogits.copy_(lr.grad)


def im2col(x, R, H, W, Cc, k, s_, p, patches, ldp):
    im2col_hw(x, R, H, W, Cc, k, s_, s_, p, patches, ldp)


def col2im(dpatches, R, H, W, Cc, k, s_, p, dx):
    col2im_hw(dpatches, R, H, W, Cc, k, s_, s_, p, dx)


def tstp_fwd(x, R, Fq, T, Cc, stats, eps=1e-7):
    xx = x.reshape(R, Fq, T, Cc)
    mean = xx.mean(2)                                            # [R, F, C]
    sd = torch.sqrt(xx.var(2, unbiased=True) + eps) if T > 1 else torch.full_like(mean, eps ** 0.5)
    st = stats.reshape(R, 2, Cc, Fq)
    st[:, 0] = mean.permute(0, 2, 1)
    st[:, 1] = sd.permute(0, 2, 1)


def tstp_bwd(x, stats, dstats, R, Fq, T, Cc, dx):
    xx = x.reshape(R, Fq, T, Cc)
    st, ds = stats.reshape(R, 2, Cc, Fq), dstats.reshape(R, 2, Cc, Fq)
    mean, sd = st[:, 0].permute(0, 2, 1).unsqueeze(2), st[:, 1].permute(0, 2, 1).unsqueeze(2)
    gm, gs = ds[:, 0].permute(0, 2, 1).unsqueeze(2), ds[:, 1].permute(0, 2, 1).unsqueeze(2)
    dx.reshape(R, Fq, T, Cc)[:] = gm / T + (gs * (xx - mean) / ((T - 1) * sd) if T > 1 else 0.0)


def power_spec(spec, M, nf, lds, ldp, p):
    sp = spec.reshape(M, lds)
    p.reshape(M, ldp)[:] = 0.0
    p.reshape(M, ldp)[:, :nf] = sp[:, 0:2 * nf:2] ** 2 + sp[:, 1:2 * nf:2] ** 2


def log_eps(x, eps):
    x.copy_(torch.log(x + eps))


def conv_wgrad(*, G, ldg, X, M, Nn, conv, slab, nsplit, tiles_per_split, bslab=None):
    """ws_conv_wgrad: slab[split][n][kk] = sum over the split's rows of G[m][n] * patches(X)[m][kk]."""
    A = _conv_matrix(X, M, conv)
    g = G.reshape(-1)[:M * ldg].reshape(M, ldg)[:, :Nn]
    Kk = A.shape[1]
    sl = slab.reshape(-1)
    for sp in range(nsplit):
        lo, hi = sp * tiles_per_split * 32, min(M, (sp + 1) * tiles_per_split * 32)
        sl[sp * Nn * Kk:(sp + 1) * Nn * Kk] = (g[lo:hi].t() @ A[lo:hi]).reshape(-1) if hi > lo else 0.0
        if bslab is not None:
            bslab.reshape(-1)[sp * Nn:(sp + 1) * Nn] = g[lo:hi].sum(0) if hi > lo else 0.0


def astp_fwd(x, logits, R, T, Cc, out, aux):
    xx, ll = x.reshape(R, T, Cc), logits.reshape(R, T, Cc)
    m = ll.max(1).values
    e = torch.exp(ll - m[:, None])
    z = e.sum(1)
    mean, ex2 = (e * xx).sum(1) / z, (e * xx * xx).sum(1) / z
    out.reshape(R, 2 * Cc)[:, :Cc] = mean
    out.reshape(R, 2 * Cc)[:, Cc:] = torch.sqrt((ex2 - mean * mean).clamp(min=1e-7))
    a = aux.reshape(R, 4, Cc)
    a[:, 0], a[:, 1], a[:, 2], a[:, 3] = m, z, mean, ex2


def astp_bwd(x, logits, out, aux, dout, R, T, Cc, dx, dlogits):
    xx, ll = x.reshape(R, T, Cc), logits.reshape(R, T, Cc)
    a = aux.reshape(R, 4, Cc)
    m, z, mean, ex2 = a[:, 0], a[:, 1], a[:, 2], a[:, 3]
    sd = out.reshape(R, 2 * Cc)[:, Cc:]
    gmean, gstd = dout.reshape(R, 2 * Cc)[:, :Cc], dout.reshape(R, 2 * Cc)[:, Cc:]
    gv = torch.where(ex2 - mean * mean > 1e-7, gstd / (2 * sd), torch.zeros_like(sd))
    gm = gmean - 2 * mean * gv
    al = torch.exp(ll - m[:, None]) / z[:, None]
    da = gm[:, None] * xx + gv[:, None] * xx * xx
    dbar = gm * mean + gv * ex2
    dx.reshape(R, T, Cc)[:] = al * (gm[:, None] + 2 * gv[:, None] * xx)
    dlogits.reshape(R, T, Cc)[:] = al * (da - dbar[:, None])


def rowbias_act_fwd(x, rb, rows, Cc, rows_per_r, act, y):
    v = x.reshape(rows, Cc)
    if rb is not None:
        v = v + rb.reshape(-1, Cc)[torch.arange(rows) // rows_per_r]
    y.reshape(rows, Cc)[:] = torch.tanh(v) if act == 1 else torch.sigmoid(v)


def act_bwd(y, dy, act, dx):
    dx.reshape(-1)[:] = (dy * ((1 - y * y) if act == 1 else y * (1 - y))).reshape(-1)


def _seg_index(T, seg_len):
    return torch.arange(T) // seg_len


def seg_sums(a, b, R, T, Cc, seg_len, out):
    v = a.reshape(R, T, Cc) * (b.reshape(R, T, Cc) if b is not None else 1.0)
    nseg = -(-T // seg_len)
    o = torch.zeros(R, nseg, Cc)
    o.index_add_(1, _seg_index(T, seg_len), v)
    out.reshape(R, nseg, Cc)[:] = o


def seg_scale(x, m, R, T, Cc, seg_len, out):
    nseg = -(-T // seg_len)
    v = m.reshape(R, nseg, Cc)[:, _seg_index(T, seg_len)]
    if x is not None:
        v = v * x.reshape(R, T, Cc)
    out.reshape(R, T, Cc)[:] = v


EMULATED = [seg_sums, seg_scale, astp_fwd, astp_bwd, rowbias_act_fwd, act_bwd, conv_wgrad, gemm_nt, gemm_tn, reduce_slabs, transpose, affine_fwd, chan_sums, im2col_hw, col2im_hw, elu_fwd, elu_bwd,
            inorm_fwd, inorm_bwd, dwconv_fwd, dwconv_bwd, avgpool_fwd, avgpool_bwd, bilinear_fwd, bilinear_bwd,
            scale_bf_fwd, scale_bf_bwd, preemph_pad, ola_fwd, ola_bwd, total_sum, lstm_pack, lstm_fwd, lstm_bwd,
            group_stats, flat_stats, gn_bwd_reduce, norm_ab, norm_bwd_apply_cl, prelu_fwd, prelu_bwd, softmax_rows_fwd,
            softmax_rows_bwd, maskmul_fwd, maskmul_bwd, relu_mask, bn_stats, bn_prelu_fwd, bn_bwd, maxpool3_fwd, maxpool3_bwd,
            bcast_rows, cross_entropy, im2col, col2im, tstp_fwd, tstp_bwd, power_spec, log_eps, rowln_ok, rowln_fwd, rowln_bwd, in_act_fwd, in_act_bwd, conv3x3, conv3x3_pack, conv3x3_pack_srcs, conv3x3_wgrad, conv3x3_wgrad_tiles, heads_ok, heads_fwd, heads_bwd]


def install(monkeypatch):
    """Route the dev entry points above to the emulation and lift the CUDA-only guards (tests only)."""
    import wesep_amd.dev as dev
    import wesep_amd.functional as f0
    import wesep_amd.functional_dpccn as fd
    import wesep_amd.functional_tasnet as ft
    import wesep_amd.functional_resnet as fr
    import wesep_amd.functional_tfgridnet as fg
    import wesep_amd.functional_ecapa as fe
    import wesep_amd.functional_campplus as fc
    monkeypatch.setattr(fe, "_need_cuda", lambda t, who: None)
    monkeypatch.setattr(fc, "_need_cuda", lambda t, who: None)
    for fn in EMULATED:
        monkeypatch.setattr(dev, fn.__name__, fn)
    for mod in (f0, fd, ft, fg, fr):
        monkeypatch.setattr(mod, "_need_cuda", lambda t, who: None)
