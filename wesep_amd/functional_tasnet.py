"""Autograd shims of the Conv-TasNet / SpEx+ path over the C ABI.

Activations are CHANNELS-LAST: a reference tensor [R, C, T'] lives here as [R*T', C] (row m = r*T' + t),
so every 1x1 convolution is one row-major GEMM (`ws_gemm_nt` / `ws_gemm_tn`, split-bf16 MFMA) and the
framing convolutions of the encoder / decoder are GEMMs over overlapping row views of the waveform.
Everything else (PReLU, gLN / cLN, depthwise dilated convolution, concatConv broadcast, mask product,
overlap-add) is in csrc/tasnet.hip.  Three autograd nodes per model: encoder, one per conv block, decoder.

Reference lines: `wesep/modules/tasnet/encoder.py:66-114`, `convs.py:41-160`, `decoder.py:66-114`,
`wesep/modules/common/norm.py:7-76`.
"""
import torch

from . import _lib as L
from . import dev
from .dev import Geom, Rows, StatMap, flat
from .functional import _empty, _need_cuda, _reduce_new

LN_EPS = dev.LN_EPS


# ---------------------------------------------------------------------------------------------
# GEMM helpers on big-M channels-last operands
# ---------------------------------------------------------------------------------------------
def _vec(*dims):
    return 3 if all(d % 4 == 0 for d in dims) else 0


def _stat_map(norm, Tp):
    """row -> statistics index: gLN one pair per utterance row, cLN one pair per frame."""
    return StatMap(Tp, 1, 1, 0, 0) if norm == "gLN" else StatMap(1, 1, 1, 0, 0)


def _gemm(x, M, K, W, Nout, *, ldw=None, w_off=0, bias=None, act=0, R=None, T=None, norm=None,
          a_rows=None, a_off=0, out=None, c_ld=None, c_off=0, vec=None, mode=None, conv=None):
    """out[m, c_off + n] = epi(sum_k pro(x[m, k]) W[n, k]); norm = (stats, gamma, beta, stat_map); conv: x is an image
    and the operand its implicit patch matrix (dev.ConvView)."""
    ldw = ldw if ldw is not None else K
    c_ld = c_ld if c_ld is not None else Nout
    if out is None:
        out = _empty(x.device, M, c_ld)
    st, gm, bt, sm = norm if norm is not None else (None, None, None, None)
    if vec is None:
        vec = _vec(K, ldw, w_off, a_off) if a_rows is None else 0
    dev.gemm_nt(A=x, a_rows=a_rows or flat(K), M=M, N=Nout, K=K, W=W, ldw=ldw, w_off=w_off, bias=bias,
                C_out=out, c_rows=flat(c_ld), c_off=c_off, a_off=a_off, act=act, R=R, T=T, stats=st,
                gamma=gm, beta=bt, stat_map=sm, vec=vec, mode=mode, conv=conv)
    return out


def _wgrad(G, M, Nn, A, Kk, *, g_ld=None, g_off=0, a_rows=None, norm=None, with_bias=True, vec=None, mode=None,
           conv=None):
    """dW [Nn, Kk] = G^T pro(A), db [Nn] = colsum(G): split slabs + deterministic reduce."""
    # enough (split x tile) workgroups to fill 256 CUs; a split keeps >= 256 rows
    # (64 slabs, or up to 512 while all slabs stay below 32 MB: a single-tile gradient over ~1 M rows -- DPCCN's
    # full-resolution 1 x 1 convolutions, TF-GridNet's value / output projections -- ran on 64 workgroups at 0.9 TB/s)
    tiles = -(-Nn // 128) * -(-Kk // 128)
    cap = max(64, min(512, (32 << 20) // (4 * Nn * Kk)))
    nsplit = max(1, min(cap, M // 256, -(-512 // tiles)))
    rps = -(-(-(-M // nsplit)) // 32) * 32
    nsplit = -(-M // rps)
    d = G.device
    slab = _empty(d, nsplit, Nn * Kk)
    bslab = _empty(d, nsplit, Nn) if with_bias else None
    st, gm, bt, sm = norm if norm is not None else (None, None, None, None)
    if vec is None:
        vec = 1 if (a_rows is None and Kk % 4 == 0) else 0
    dev.gemm_tn(G=G, g_rows=flat(g_ld if g_ld is not None else Nn), g_off=g_off, A=A, a_rows=a_rows or flat(Kk),
                M=M, Nn=Nn, Kk=Kk, slab=slab, slab_stride=Nn * Kk, bslab=bslab, bslab_stride=Nn, nsplit=nsplit,
                rows_per_split=rps, stats=st, gamma=gm, beta=bt, stat_map=sm, vec=vec, mode=mode, conv=conv)
    dW = _reduce_new(slab, nsplit, Nn * Kk, (Nn, Kk))
    db = _reduce_new(bslab, nsplit, Nn, (Nn,)) if with_bias else None
    return dW, db


def _transposed(W, rows, cols, lds=None, src_off=0):
    WT = _empty(W.device, cols, rows)
    dev.transpose(W, rows, cols, lds if lds is not None else cols, WT, src_off=src_off)
    return WT


# ---------------------------------------------------------------------------------------------
# channel-affine norms on channels-last tensors
# ---------------------------------------------------------------------------------------------
def _cln_geom(M, Cc):
    return Geom(M, 1, Cc, 0, Cc, 1, Cc, 1)


def norm_stats(x, norm, R, Tp, Cc):
    """(mean, rstd) pairs: gLN [R, 2] over (T', C) (norm.py:39-40), cLN [R*T', 2] over C (norm.py:51-59)."""
    if norm == "gLN":
        st = _empty(x.device, R, 2)
        dev.flat_stats(x, R, Tp * Cc, st, LN_EPS)
    elif norm == "cLN":
        st = _empty(x.device, R * Tp, 2)
        dev.group_stats(x, _cln_geom(R * Tp, Cc), st, LN_EPS)
    else:
        raise NotImplementedError(f"norm {norm!r}: only gLN and cLN are built")
    return st


def norm_backward(x, dxn, stats, gamma, norm, R, Tp, Cc, res=None):
    """dx (written over dxn), dgamma [C], dbeta [C] of y = gamma * xhat + beta."""
    M = R * Tp
    d = x.device
    if norm == "gLN":
        sums = dev.chan_sums(dxn, x, stats, Tp, Tp, R, Cc)            # [R, 2, C]
        ab = _empty(d, R, 2)
        dev.norm_ab(sums, gamma, R, Cc, Tp * Cc, ab)
        tot = _empty(d, 2, Cc)
        dev.reduce_slabs(sums, R, 2 * Cc, 2 * Cc, tot)
        st_div = Tp
    else:
        ab = _empty(d, M, 2)
        dev.gn_bwd_reduce(x, dxn, stats, _cln_geom(M, Cc), ab, gamma=gamma)
        tot = dev.chan_sums(dxn, x, stats, 1, M, 1, Cc).view(2, Cc)
        st_div = 1
    dev.norm_bwd_apply_cl(x, dxn, stats, ab, gamma, res, M, Cc, st_div, dxn)
    return dxn, tot[1].contiguous(), tot[0].contiguous()


# ---------------------------------------------------------------------------------------------
# MultiEncoder (encoder.py:66-114)
# ---------------------------------------------------------------------------------------------
class MultiEncoderFn(torch.autograd.Function):
    """wav [R, T] -> (e [R*T', B], cat [R*T', 3N]); cat = ReLU outputs w1 | w2 | w3, kept for the decoder."""

    @staticmethod
    def forward(ctx, wav, stride, w1, b1, w2, b2, w3, b3, ln_w, ln_b, pw, pb):
        _need_cuda(wav, "ConvTasNet")
        R, T = wav.shape
        N = w1.shape[0]
        Ls = (w1.shape[-1], w2.shape[-1], w3.shape[-1])
        if T < Ls[0]:
            raise RuntimeError(f"ConvTasNet: input of {T} samples is shorter than the encoder window {Ls[0]}")
        Tp = (T - Ls[0]) // stride + 1
        Tpad = max(T, (Tp - 1) * stride + max(Ls))          # zero extension of encoder.py:106-111
        Tpad = -(-Tpad // 4) * 4
        d = wav.device
        xp = torch.zeros(R, Tpad, device=d, dtype=torch.float32)
        xp[:, :T] = wav
        M = R * Tp
        frames = Rows(Tp, Tpad, stride)                      # row m -> xp[r, t*stride : t*stride + L]
        cat = _empty(d, M, 3 * N)
        for i, (w, b) in enumerate(((w1, b1), (w2, b2), (w3, b3))):
            Lk = Ls[i]
            _gemm(xp, M, Lk, w.reshape(N, Lk).contiguous(), N, bias=b, act=2, a_rows=frames, out=cat,
                  c_ld=3 * N, c_off=i * N, vec=2 if Lk % 4 == 0 else 0)
        st = _empty(d, M, 2)
        dev.group_stats(cat, _cln_geom(M, 3 * N), st, LN_EPS)
        B = pw.shape[0]
        W2 = pw.reshape(B, 3 * N).contiguous()
        e = _gemm(cat, M, 3 * N, W2, B, bias=pb, norm=(st, ln_w, ln_b, StatMap(1, 1, 1, 0, 0)))
        ctx.save_for_backward(xp, cat, st, w1, w2, w3, ln_w, ln_b, W2)
        ctx.geo = (R, Tp, Tpad, stride, N, B, Ls)
        ctx.set_materialize_grads(False)     # the enrollment pass uses `cat` only: de arrives as None
        return e, cat

    @staticmethod
    def backward(ctx, de, dcat_dec):
        xp, cat, st, w1, w2, w3, ln_w, ln_b, W2 = ctx.saved_tensors
        R, Tp, Tpad, stride, N, B, Ls = ctx.geo
        M = R * Tp
        res = dcat_dec.contiguous() if dcat_dec is not None else None
        if de is not None:
            de = de.contiguous()
            sm = StatMap(1, 1, 1, 0, 0)
            dpw, dpb = _wgrad(de, M, B, cat, 3 * N, norm=(st, ln_w, ln_b, sm))
            dxn = _gemm(de, M, B, _transposed(W2, B, 3 * N), 3 * N)
            dcat, dlnw, dlnb = norm_backward(cat, dxn, st, ln_w, "cLN", R, Tp, 3 * N, res=res)
            dpw = dpw.view(B, 3 * N, 1)
        else:                                  # only the ReLU outputs were used (speaker-encoder input)
            if res is None:
                return (None,) * 12
            dcat, dlnw, dlnb, dpw, dpb = res.clone(), None, None, None, None
        dev.relu_mask(dcat, cat)
        frames = Rows(Tp, Tpad, stride)
        grads = []
        for i, w in enumerate((w1, w2, w3)):
            Lk = Ls[i]
            dW, db = _wgrad(dcat, M, N, xp, Lk, g_ld=3 * N, g_off=i * N, a_rows=frames, vec=0)
            grads += [dW.view(N, 1, Lk), db]
        return (None, None, *grads, dlnw, dlnb, dpw, dpb)


# ---------------------------------------------------------------------------------------------
# Conv1DBlock / Conv1DBlock4Fuse (convs.py:41-160): gLN / cLN / BN, non-causal or causal, optional skip branch
# ---------------------------------------------------------------------------------------------
def _block_norm(y, norm, R, Tp, Cc, gamma, beta, bn):
    """Statistics of one in-block norm over y [R*T', C] -> (kernel-facing (stats, gamma, beta, stat_map, st_div),
    backward-facing (stats, gamma)).  gLN / cLN: (mean, rstd) per utterance / per frame, applied on load by the consumer
    kernels.  BN (select_norm 'BN' = nn.BatchNorm1d, norm.py:62-76): per-CHANNEL statistics, i.e. a per-column affine
    map -- folded into the consumer's gamma / beta with identity row statistics, so the same kernels serve it."""
    if norm != "BN":
        st = norm_stats(y, norm, R, Tp, Cc)
        return (st, gamma, beta, _stat_map(norm, Tp), Tp if norm == "gLN" else 1), (st, gamma)
    rm, rv, training = bn
    M = R * Tp
    st = _empty(y.device, 2, Cc)
    if training:
        dev.bn_stats(y, M, Cc, rm, rv, st)
    else:
        st[0].copy_(rm)
        st[1].copy_(torch.rsqrt(rv + dev.BN_EPS))
    gk = (st[1] * gamma).contiguous()
    bk = (beta - st[0] * gk).contiguous()
    ident = _ident(y.device)
    return (ident, gk, bk, StatMap(1, 0, 1, 0, 0), M), (st, gamma)


_IDENT = {}


def _ident(device):
    """(mean, rstd) = (0, 1) on the device, made once: torch.tensor(..., device=...) is a pageable host-to-device copy and
    synchronises the stream -- once per BatchNorm layer and step here (round 6)."""
    key = (device.type, device.index)
    if key not in _IDENT:
        _IDENT[key] = torch.tensor([[0.0, 1.0]], device=device, dtype=torch.float32)
    return _IDENT[key]


def _block_norm_backward(y, dyn, st, gamma, norm, R, Tp, Cc):
    """dy (written over dyn), dgamma, dbeta."""
    if norm != "BN":
        return norm_backward(y, dyn, st, gamma, norm, R, Tp, Cc)
    sums = dev.bn_bwd(y, dyn, st, gamma, R * Tp, Cc, dyn)
    return dyn, sums[1].contiguous(), sums[0].contiguous()


class ConvBlockFn(torch.autograd.Function):
    """x [R*T', B] -> x + sconv(norm2(prelu2(dconv(norm1(prelu1(conv1x1(x) + rb))))));
    rb [R, H] = W_e e + b (concatConv fusion, convs.py:143-148) or None (then the conv1x1 bias is used).
    geo = (R, T', norm, dilation[, causal[, bn]]): causal puts every depthwise tap at or before t (convs.py:61-62,91-92);
    bn = (rm1, rv1, rm2, rv2, training) are the BatchNorm1d buffers of norm = 'BN'.  With the skip branch's (wsc, bsc)
    (skip_con, convs.py:73-75,98-101) the result is the pair (x + Output(c), Sc_conv(c))."""

    @staticmethod
    def forward(ctx, x, rb, geo, w1, b1, a1, g1, be1, wd, bd, a2, g2, be2, w3, b3, wsc=None, bsc=None):
        _need_cuda(x, "ConvTasNet")
        R, Tp, norm, dil = geo[:4]
        causal = bool(geo[4]) if len(geo) > 4 else False
        bn = geo[5] if len(geo) > 5 else None
        M, B = x.shape
        H, P = wd.shape[0], wd.shape[-1]
        x = x.contiguous()
        ldw = w1.shape[1]
        W1 = w1.reshape(H, ldw).contiguous()
        c = _gemm(x, M, B, W1, H, ldw=ldw, bias=None if rb is not None else b1)
        y1 = _empty(x.device, M, H)
        dev.prelu_fwd(c, rb.contiguous() if rb is not None else None, a1, M, H, Tp, y1)
        g1f, be1f, g2f, be2f = (t.reshape(H).contiguous() for t in (g1, be1, g2, be2))
        (st1, g1k, be1k, _, st_div), (st1n, _) = _block_norm(y1, norm, R, Tp, H, g1f, be1f, bn and (bn[0], bn[1], bn[4]))
        wdf = wd.reshape(H, P).contiguous()
        z = _empty(x.device, M, H)
        dev.dwconv_fwd(y1, st1, g1k, be1k, wdf, bd, R, Tp, H, P, dil, st_div, z, causal=causal)
        y2 = _empty(x.device, M, H)
        dev.prelu_fwd(z, None, a2, M, H, Tp, y2)
        (st2, g2k, be2k, sm, _), (st2n, _) = _block_norm(y2, norm, R, Tp, H, g2f, be2f, bn and (bn[2], bn[3], bn[4]))
        W3 = w3.reshape(B, H).contiguous()
        out = _gemm(y2, M, H, W3, B, bias=b3, R=x, norm=(st2, g2k, be2k, sm))
        skip = wsc is not None
        Wsc = wsc.reshape(B, H).contiguous() if skip else W3
        sc = _gemm(y2, M, H, Wsc, B, bias=bsc, norm=(st2, g2k, be2k, sm)) if skip else None
        ctx.save_for_backward(x, c, y1, st1, z, y2, st2, W1, a1, g1k, be1k, wdf, a2, g2k, be2k, W3, st1n, st2n, g1f, g2f,
                              Wsc)
        ctx.geo = (R, Tp, norm, dil, B, H, P, ldw, rb is not None, causal, skip, st_div, sm)
        ctx.shapes = (w1.shape, g1.shape, wd.shape, w3.shape)
        if bn is not None and not bn[4]:
            ctx.eval_bn = True
        if skip:
            ctx.set_materialize_grads(False)     # the last block's `out` is dropped by Separation: its gradient is None
            return out, sc
        return out

    @staticmethod
    def backward(ctx, dout, dsc=None):
        (x, c, y1, st1, z, y2, st2, W1, a1, g1k, be1k, wdf, a2, g2k, be2k, W3, st1n, st2n, g1f, g2f,
         Wsc) = ctx.saved_tensors
        R, Tp, norm, dil, B, H, P, ldw, has_rb, causal, skip, st_div, sm = ctx.geo
        if getattr(ctx, "eval_bn", False):
            raise L.WesepHipError("ConvTasNet norm='BN': backward in eval mode (running statistics) is not built")
        M = R * Tp
        d = x.device
        dout = dout.contiguous() if dout is not None else None
        dsc = dsc.contiguous() if dsc is not None else None
        if dout is None and dsc is None:
            return (None,) * (17 if skip else 15)
        nk = (st2, g2k, be2k, sm)
        dW3 = db3 = dWsc = dbsc = dyn2 = None
        if dout is not None:
            dW3, db3 = _wgrad(dout, M, B, y2, H, norm=nk)
            dyn2 = _gemm(dout, M, B, _transposed(W3, B, H), H)
        if dsc is not None:
            dWsc, dbsc = _wgrad(dsc, M, B, y2, H, norm=nk)
            dyn2 = _gemm(dsc, M, B, _transposed(Wsc, B, H), H, R=dyn2)
        dy2, dg2, dbe2 = _block_norm_backward(y2, dyn2, st2n, g2f, norm, R, Tp, H)
        da2 = dev.prelu_bwd(z, dy2, a2, dy2)                                   # dy2 -> dz in place
        dyn1 = _empty(d, M, H)
        dwd, dbd = dev.dwconv_bwd(dy2, y1, st1, g1k, be1k, wdf, R, Tp, H, P, dil, st_div, dyn1, causal=causal)
        dy1, dg1, dbe1 = _block_norm_backward(y1, dyn1, st1n, g1f, norm, R, Tp, H)
        da1 = dev.prelu_bwd(c, dy1, a1, dy1)                                   # dy1 -> dc in place
        drb = None
        if has_rb:
            drb = dev.chan_sums(dy1, None, None, 1, Tp, R, H)[:, 0, :].contiguous()   # [R, H]
        dW1x, db1 = _wgrad(dy1, M, H, x, B, with_bias=not has_rb)
        if ldw != B:                                                          # speaker columns: via rb's producer
            dW1 = torch.zeros(H, ldw, device=d, dtype=torch.float32)
            dW1[:, :B] = dW1x
        else:
            dW1 = dW1x
        W1xT = _transposed(W1, H, B, lds=ldw)
        dx = _gemm(dy1, M, H, W1xT, B, R=dout)
        s1, sg, sd, s3 = ctx.shapes
        grads = (dx, drb, None, dW1.view(s1), db1, da1, dg1.view(sg), dbe1.view(sg), dwd.reshape(sd), dbd, da2,
                 dg2.view(sg), dbe2.view(sg), dW3.view(s3) if dW3 is not None else None, db3)
        if skip:
            grads += (dWsc.view(s3) if dWsc is not None else None, dbsc)
        return grads


# ---------------------------------------------------------------------------------------------
# MultiDecoder (decoder.py:66-114), actLayer = ReLU
# ---------------------------------------------------------------------------------------------
class MultiDecoderFn(torch.autograd.Function):
    """(e [R*T', B], cat [R*T', 3N]) -> est1, est2, est3 [R, (T'-1)*stride + L1]."""

    @staticmethod
    def forward(ctx, e, cat, geo, *params):
        _need_cuda(e, "ConvTasNet")
        R, Tp, stride = geo
        M, B = e.shape
        N = cat.shape[1] // 3
        e, cat = e.contiguous(), cat.contiguous()
        d = e.device
        L1 = params[6].shape[-1]
        xlen = (Tp - 1) * stride + L1
        ests, masks = [], []
        for i in range(3):
            wm, bm = params[2 * i], params[2 * i + 1]
            wdec, bdec = params[6 + 2 * i], params[7 + 2 * i]
            Lk = wdec.shape[-1]
            m = _gemm(e, M, B, wm.reshape(N, B).contiguous(), N, bias=bm, act=2)
            s = _empty(d, M, N)
            dev.maskmul_fwd(cat, i * N, 3 * N, m, M, N, s)
            WT = _transposed(wdec.reshape(N, Lk).contiguous(), N, Lk)            # [Lk, N]
            fr = _gemm(s, M, N, WT, Lk)
            est = _empty(d, R, xlen)
            dev.ola_fwd(fr, bdec, R, Tp, Lk, stride, xlen, est)
            ests.append(est)
            masks.append(m)
        ctx.save_for_backward(e, cat, *masks, *params)
        ctx.geo = (R, Tp, stride, B, N, xlen)
        return tuple(ests)

    @staticmethod
    def backward(ctx, *dests):
        e, cat = ctx.saved_tensors[:2]
        masks = ctx.saved_tensors[2:5]
        params = ctx.saved_tensors[5:]
        R, Tp, stride, B, N, xlen = ctx.geo
        M = R * Tp
        d = e.device
        dcat = torch.zeros(M, 3 * N, device=d, dtype=torch.float32)
        de = None
        gm = [None] * 6
        gd = [None] * 6
        for i in range(3):
            if dests[i] is None:
                continue
            dest = dests[i].contiguous()
            wm = params[2 * i]
            wdec = params[6 + 2 * i]
            Lk = wdec.shape[-1]
            W2 = wdec.reshape(N, Lk).contiguous()
            dfr = _empty(d, M, Lk)
            dev.ola_bwd(dest, R, Tp, Lk, stride, xlen, dfr)
            s = _empty(d, M, N)
            dev.maskmul_fwd(cat, i * N, 3 * N, masks[i], M, N, s)                # recomputed, not saved
            dWd, _ = _wgrad(s, M, N, dfr, Lk, with_bias=False)
            ds = _gemm(dfr, M, Lk, W2, N)
            dm = s                                                               # reuse the buffer
            dev.maskmul_bwd(ds, cat, i * N, 3 * N, masks[i], M, N, dcat, i * N, 3 * N, dm)
            Wm = wm.reshape(N, B).contiguous()
            dWm, dbm = _wgrad(dm, M, N, e, B)
            de = _gemm(dm, M, N, _transposed(Wm, N, B), B, R=de)
            gm[2 * i], gm[2 * i + 1] = dWm.view(N, B, 1), dbm
            gd[2 * i], gd[2 * i + 1] = dWd.view(N, 1, Lk), dev.total_sum(dest)
        return (de, dcat, None, *gm, *gd)


# ---------------------------------------------------------------------------------------------
# The classic Conv-TasNet ends (convtasnet.py:79-85,147-153: encoder_type / decoder_type other than 'Multi'):
# one strided Conv1d encoder, mask * encoder output, one ConvTranspose1d decoder
# ---------------------------------------------------------------------------------------------
class PlainEncoderFn(torch.autograd.Function):
    """wav [R, T] -> conv1d(wav, w [N, 1, L], b; stride) (-> ReLU) as channels-last frames [R*T', N], T' = (T - L) // stride
    + 1: one GEMM on the overlapping frame view of the waveform (no frames are materialised)."""

    @staticmethod
    def forward(ctx, wav, stride, relu, w, b):
        _need_cuda(wav, "ConvTasNet")
        R, T = wav.shape
        N, _, Lk = w.shape
        if T < Lk:
            raise RuntimeError(f"ConvTasNet: input of {T} samples is shorter than the encoder window {Lk}")
        Tp = (T - Lk) // stride + 1
        Tpad = -(-T // 4) * 4
        d = wav.device
        xp = torch.zeros(R, Tpad, device=d, dtype=torch.float32)
        xp[:, :T] = wav
        M = R * Tp
        frames = Rows(Tp, Tpad, stride)                      # row m -> xp[r, t*stride : t*stride + L]
        y = _gemm(xp, M, Lk, w.reshape(N, Lk).contiguous(), N, bias=b, act=2 if relu else 0, a_rows=frames,
                  vec=2 if Lk % 4 == 0 else 0)
        ctx.save_for_backward(xp, y if relu else None)
        ctx.geo = (R, Tp, Tpad, stride, N, Lk, relu)
        return y

    @staticmethod
    def backward(ctx, dy):
        xp, y = ctx.saved_tensors
        R, Tp, Tpad, stride, N, Lk, relu = ctx.geo
        dy = dy.contiguous()
        if relu:
            dy = dy.clone()
            dev.relu_mask(dy, y)
        dW, db = _wgrad(dy, R * Tp, N, xp, Lk, a_rows=Rows(Tp, Tpad, stride), vec=0)
        return None, None, None, dW.view(N, 1, Lk), db


class TransDecoderFn(torch.autograd.Function):
    """x [R*T', N] -> conv_transpose1d(x, w [N, 1, L], b [1]; stride) [R, (T' - 1) * stride + L]: frame synthesis GEMM +
    overlap-add (decoder.py / convs.py:27-41 ConvTrans1D)."""

    @staticmethod
    def forward(ctx, x, geo, w, b):
        _need_cuda(x, "ConvTasNet")
        R, Tp, stride = geo
        x = x.contiguous()
        M, N = x.shape
        Lk = w.shape[-1]
        W2 = w.reshape(N, Lk).contiguous()
        fr = _gemm(x, M, N, _transposed(W2, N, Lk), Lk)
        xlen = (Tp - 1) * stride + Lk
        est = _empty(x.device, R, xlen)
        dev.ola_fwd(fr, b, R, Tp, Lk, stride, xlen, est)
        ctx.save_for_backward(x, W2)
        ctx.geo = (R, Tp, stride, N, Lk, xlen)
        return est

    @staticmethod
    def backward(ctx, dest):
        x, W2 = ctx.saved_tensors
        R, Tp, stride, N, Lk, xlen = ctx.geo
        M = R * Tp
        dest = dest.contiguous()
        dfr = _empty(x.device, M, Lk)
        dev.ola_bwd(dest, R, Tp, Lk, stride, xlen, dfr)
        dW, _ = _wgrad(x, M, N, dfr, Lk, with_bias=False)
        dx = _gemm(dfr, M, Lk, W2, N) if ctx.needs_input_grad[0] else None
        return dx, None, dW.view(N, 1, Lk), dev.total_sum(dest)


class MulFn(torch.autograd.Function):
    """a * b on [M, C] (mask times encoder output), both operands differentiable."""

    @staticmethod
    def forward(ctx, a, b):
        _need_cuda(a, "ConvTasNet")
        a, b = a.contiguous(), b.contiguous()
        M, Cc = a.shape
        y = torch.empty_like(a)
        dev.maskmul_fwd(a, 0, Cc, b, M, Cc, y)
        ctx.save_for_backward(a, b)
        return y

    @staticmethod
    def backward(ctx, dy):
        a, b = ctx.saved_tensors
        M, Cc = a.shape
        dy = dy.contiguous()
        da, db = torch.empty_like(a), torch.empty_like(a)
        dev.maskmul_fwd(dy, 0, Cc, b, M, Cc, da)
        dev.maskmul_fwd(dy, 0, Cc, a, M, Cc, db)
        return da, db


# ---------------------------------------------------------------------------------------------
# ResNet4SpExplus (tasnet/speaker.py:49-64) on the shared encoder's [w1 | w2 | w3] of the enrollment
# ---------------------------------------------------------------------------------------------
# GEMM mode of the speaker-encoder branch.  Exact-fp32 MFMA: three BatchNorm stages over T0/3, T0/9, T0/27 frames
# amplify operand rounding in the backward (measured: split-bf16 products give 5e-3 relative gradient error on the
# first ResBlock at a 2.4 k-sample enrollment), and the branch is < 1 % of the step's FLOPs.
SPK_MODE = "f32"


class SpkEncoderFn(torch.autograd.Function):
    """cat_aux [R*T0, 768] -> speaker embedding [R, E].
    cLN - 1x1 conv - 3 x ResBlock(conv1x1, BN, PReLU, conv1x1, BN, + residual, PReLU, MaxPool1d(3)) - 1x1 conv - mean
    over time (the last convolution is applied after the mean: both are linear).  BatchNorm1d runs in training mode
    (batch statistics, running buffers updated in place) or, with training=False, on the running statistics.
    params: ln_w, ln_b, w1, b1, then per block (conv1.w, conv2.w, bn1.w, bn1.b, bn2.w, bn2.b, prelu1, prelu2,
    downsample.w or None), then w5, b5.  buffers: per block (rm1, rv1, rm2, rv2).
    """

    @staticmethod
    def forward(ctx, cat, geo, buffers, *params):
        _need_cuda(cat, "ConvTasNet speaker encoder")
        R, T0, training = geo
        cat = cat.contiguous()
        d = cat.device
        M0, C0 = cat.shape
        ln_w, ln_b, w1, b1 = params[:4]
        w5, b5 = params[-2:]
        st0 = _empty(d, M0, 2)
        dev.group_stats(cat, _cln_geom(M0, C0), st0, LN_EPS)
        W1 = w1.reshape(w1.shape[0], C0).contiguous()
        x = _gemm(cat, M0, C0, W1, W1.shape[0], bias=b1, norm=(st0, ln_w, ln_b, StatMap(1, 1, 1, 0, 0)))
        saved, meta = [cat, st0, W1, ln_w, ln_b], []
        T = T0
        for i in range(3):
            wc1, wc2, g1, be1, g2, be2, a1, a2, wds = params[4 + 9 * i: 13 + 9 * i]
            rm1, rv1, rm2, rv2 = buffers[4 * i: 4 * i + 4]
            co, ci = wc1.shape[0], wc1.shape[1]
            M = R * T
            Wc1, Wc2 = wc1.reshape(co, ci).contiguous(), wc2.reshape(co, co).contiguous()
            c1 = _gemm(x, M, ci, Wc1, co, mode=SPK_MODE)
            s1 = SpkEncoderFn._bn_stats(c1, M, co, rm1, rv1, training)
            u1, y1 = _empty(d, M, co), _empty(d, M, co)
            dev.bn_prelu_fwd(c1, s1, g1, be1, None, a1, M, co, u1, y1)
            c2 = _gemm(y1, M, co, Wc2, co, mode=SPK_MODE)
            s2 = SpkEncoderFn._bn_stats(c2, M, co, rm2, rv2, training)
            Wds = wds.reshape(co, ci).contiguous() if wds is not None else None
            res = _gemm(x, M, ci, Wds, co, mode=SPK_MODE) if Wds is not None else x
            u2, y2 = _empty(d, M, co), _empty(d, M, co)
            dev.bn_prelu_fwd(c2, s2, g2, be2, res, a2, M, co, u2, y2)
            To = T // 3
            if To < 1:
                raise RuntimeError("ConvTasNet speaker encoder: enrollment too short for three MaxPool1d(3) stages")
            p = _empty(d, R * To, co)
            dev.maxpool3_fwd(y2, R, T, co, p)
            saved += [x, c1, s1, u1, y1, c2, s2, u2, y2, Wc1, Wc2, g1, g2, a1, a2] + ([Wds] if Wds is not None else [])
            meta.append((T, ci, co, Wds is not None, wc1.shape, wc2.shape, wds.shape if wds is not None else None))
            x, T = p, To
        C3 = x.shape[1]
        mean = dev.chan_sums(x, None, None, 1, T, R, C3)[:, 0, :].contiguous()
        dev.affine_fwd(mean, None, None, 1.0 / T, R, 1, C3, mean)
        W5 = w5.reshape(w5.shape[0], C3).contiguous()
        emb = _gemm(mean, R, C3, W5, W5.shape[0], bias=b5, mode=SPK_MODE)
        ctx.save_for_backward(*saved, mean, W5)
        ctx.meta = (R, T0, T, C0, meta, (w1.shape, w5.shape), training)
        return emb

    @staticmethod
    def _bn_stats(c, M, Cc, rm, rv, training):
        st = _empty(c.device, 2, Cc)
        if training:
            dev.bn_stats(c, M, Cc, rm, rv, st)
        else:
            st[0].copy_(rm)
            st[1].copy_(torch.rsqrt(rv + dev.BN_EPS))
        return st

    @staticmethod
    def backward(ctx, demb):
        from .functional import _lin_bwd_w
        R, T0, T3, C0, meta, (s1shape, s5shape), training = ctx.meta
        if not training:
            raise L.WesepHipError("speaker encoder backward in eval mode (running statistics) is not built")
        sv = list(ctx.saved_tensors)
        cat, st0, W1, ln_w, ln_b = sv[:5]
        mean, W5 = sv[-2:]
        d = cat.device
        demb = demb.contiguous()
        dW5, db5 = _lin_bwd_w(demb, mean)
        C3 = W5.shape[1]
        dmean = _gemm(demb, R, W5.shape[0], _transposed(W5, W5.shape[0], C3), C3, mode=SPK_MODE)
        dp = _empty(d, R * T3, C3)
        dev.bcast_rows(dmean, 1.0 / T3, T3, R * T3, C3, dp)
        grads_blocks = []
        pos = len(sv) - 2
        for (T, ci, co, has_ds, sh1, sh2, shd) in reversed(meta):
            n = 15 + (1 if has_ds else 0)
            blk = sv[pos - n: pos]
            pos -= n
            x, c1, s1, u1, y1, c2, s2, u2, y2, Wc1, Wc2, g1, g2, a1, a2 = blk[:15]
            Wds = blk[15] if has_ds else None
            M = R * T
            dy2 = _empty(d, M, co)
            dev.maxpool3_bwd(y2, dp, R, T, co, dy2)
            da2 = dev.prelu_bwd(u2, dy2, a2, dy2)                      # -> du2 (also the residual gradient)
            dc2 = _empty(d, M, co)
            sums2 = dev.bn_bwd(c2, dy2, s2, g2, M, co, dc2)
            dWc2, _ = _wgrad(dc2, M, co, y1, co, with_bias=False, mode=SPK_MODE)
            dy1 = _gemm(dc2, M, co, _transposed(Wc2, co, co), co, mode=SPK_MODE)
            da1 = dev.prelu_bwd(u1, dy1, a1, dy1)                      # -> du1
            sums1 = dev.bn_bwd(c1, dy1, s1, g1, M, co, dy1)            # -> dc1 in place
            dWc1, _ = _wgrad(dy1, M, co, x, ci, with_bias=False, mode=SPK_MODE)
            if has_ds:
                dWds, _ = _wgrad(dy2, M, co, x, ci, with_bias=False, mode=SPK_MODE)
                dres = _gemm(dy2, M, co, _transposed(Wds, co, ci), ci, mode=SPK_MODE)
            else:
                dWds, dres = None, dy2
            dp = _gemm(dy1, M, co, _transposed(Wc1, co, ci), ci, R=dres, mode=SPK_MODE)
            grads_blocks.append([dWc1.view(sh1), dWc2.view(sh2), sums1[1].contiguous(), sums1[0].contiguous(),
                                 sums2[1].contiguous(), sums2[0].contiguous(), da1, da2,
                                 dWds.view(shd) if has_ds else None])
        M0 = R * T0
        sm = StatMap(1, 1, 1, 0, 0)
        dW1, db1 = _wgrad(dp, M0, W1.shape[0], cat, C0, norm=(st0, ln_w, ln_b, sm), mode=SPK_MODE)
        dxn = _gemm(dp, M0, W1.shape[0], _transposed(W1, W1.shape[0], C0), C0, mode=SPK_MODE)
        dcat, dlnw, dlnb = norm_backward(cat, dxn, st0, ln_w, "cLN", R, T0, C0)
        flat_blocks = [g for blk in reversed(grads_blocks) for g in blk]
        return (dcat, None, None, dlnw, dlnb, dW1.view(s1shape), db1, *flat_blocks, dW5.view(s5shape), db5)


class CrossEntropyFn(torch.autograd.Function):
    """nn.CrossEntropyLoss (mean) on [R, S] logits with int64 labels."""

    @staticmethod
    def forward(ctx, logits, label):
        _need_cuda(logits, "CrossEntropyLoss")
        logits = logits.contiguous().float()
        loss = _empty(logits.device, 1)
        dlogits = torch.empty_like(logits)
        dev.cross_entropy(logits, label.contiguous().long(), loss, dlogits)
        ctx.save_for_backward(dlogits)
        return loss.view(())

    @staticmethod
    def backward(ctx, gout):
        (dlogits,) = ctx.saved_tensors
        return dlogits * gout, None
