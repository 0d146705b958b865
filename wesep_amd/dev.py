"""Typed Python wrappers over the C ABI (include/wesep_hip.h): torch tensors in, raw device
pointers + the current HIP stream out.  No arithmetic happens here."""
import ctypes as C
import os
from typing import NamedTuple, Optional

import numpy as np
import torch

from . import _lib as L

BIG = 1 << 30  # divisor meaning "never wraps" (flat row addressing)


def gemm_mode() -> str:
    """'bf16x3' (default): GEMM products as 3 bf16 MFMAs on hi/lo splits, fp32 accumulate;
    'f32': exact fp32 MFMA.  Set WESEP_GEMM=f32 to force the exact kernels."""
    return os.environ.get("WESEP_GEMM", "bf16x3")


def _mode_bit(mode):
    return 4 if (mode or gemm_mode()) == "bf16x3" else 0
GN_EPS = float(np.finfo(np.float32).eps)  # bsrnn.py:23


class Rows(NamedTuple):
    """row m -> (m // div) * s1 + (m % div) * s2 elements."""
    div: int
    s1: int
    s2: int


def flat(ld: int) -> Rows:
    return Rows(BIG, 0, ld)


class StatMap(NamedTuple):
    """row m -> stat index (m // div1) * m1 + (m % div2) * m2 + base."""
    div1: int
    m1: int
    div2: int
    m2: int
    base: int = 0


def _chk(t: Optional[torch.Tensor], name: str, dtype=torch.float32):
    if t is None:
        return
    if not t.is_cuda or t.dtype != dtype or not t.is_contiguous():
        raise L.WesepHipError(f"{name}: expected a contiguous {dtype} CUDA tensor, got "
                              f"{t.dtype} {t.device} contiguous={t.is_contiguous()}")


def _p(t: Optional[torch.Tensor], off: int = 0):
    if t is None:
        return None
    return C.c_void_p(t.data_ptr() + 4 * off)


# Algorithmic work counters for the bench tools (None = off): kind -> [bytes, flops, launches] of what a launch must
# move / compute at least (operands once, outputs once; an implicit patch matrix counts as its image), next to the
# HIP-event time of the same kind (prof_collect) -> roofline fractions without a profiler.
ALG = None
ALG_LSTM_UNITS = None    # hidden units the recurrence launches are PRICED at (None: the kernels' 256)


def alg_reset(on=True, lstm_units=None):
    """lstm_units: the model's real LSTM hidden size when it is smaller than the 256 units the kernels are built for
    (TF-GridNet: 192, zero-padded) -- algorithmic bytes / FLOPs of the recurrence launches then count the real units; the
    padding is the kernel's cost, not the algorithm's (priced at 256, counter traffic came out BELOW the
    'algorithmic' bytes)."""
    global ALG, ALG_LSTM_UNITS
    ALG = {} if on else None
    ALG_LSTM_UNITS = lstm_units if on else None


def _alg(kind, nbytes, flops):
    if ALG is not None:
        e = ALG.setdefault(kind, [0, 0, 0])
        e[0] += int(nbytes)
        e[1] += int(flops)
        e[2] += 1


class ConvView(NamedTuple):
    """ws_conv_view (include/wesep_hip.h): the A operand of gemm_nt / gemm_tn is the never-materialised im2col matrix
    of the channels-last image [R][H][W][C] that A points to; (Ho, Wo) = rows per image; mode 0 = convolution view
    (input pixel = o*s + tap - p), mode 1 = transposed view (input pixel = (o + p - tap) / s, exact divisions only)."""
    mode: int
    H: int
    W: int
    C: int
    Ho: int
    Wo: int
    k: int
    sh: int
    sw: int
    p: int
    dil: int = 1
    ldp: int = 0          # floats between consecutive pixels (0 = C): the image is the first C columns of a wider tensor


def _set_conv(a, conv: Optional[ConvView]):
    if conv is not None:
        a.conv.on = 1
        (a.conv.mode, a.conv.H, a.conv.W, a.conv.C, a.conv.Ho, a.conv.Wo, a.conv.k, a.conv.sh, a.conv.sw,
         a.conv.p, a.conv.dil, a.conv.ldp) = conv


def gemm_nt(*, A, a_rows: Rows, M: int, C_out, c_rows: Rows, N=0, K=0, W=None, ldw=0, bias=None,
            R=None, T=None, stats=None, gamma=None, beta=None, stat_map: Optional[StatMap] = None,
            act=0, groups=None, ngroups=0, max_n=0, vec=3, a_off=0, c_off=0, w_off=0, mode=None,
            conv: Optional[ConvView] = None):
    for n, t in (("A", A), ("W", W), ("bias", bias), ("C", C_out), ("R", R), ("T", T),
                 ("stats", stats), ("gamma", gamma), ("beta", beta)):
        _chk(t, n)
    a = L.GemmNTArgs()
    a.A, a.W, a.bias, a.C = _p(A, a_off), _p(W, w_off), _p(bias), _p(C_out, c_off)
    a.R, a.T = _p(R, c_off), _p(T, c_off)
    a.stats, a.gamma, a.beta = _p(stats), _p(gamma), _p(beta)
    a.groups = C.c_void_p(groups.data_ptr()) if groups is not None else None
    a.a_div, a.a_s1, a.a_s2 = a_rows
    a.c_div, a.c_s1, a.c_s2 = c_rows
    sm = stat_map or StatMap(1, 0, 1, 0, 0)
    a.st_div1, a.st_m1, a.st_div2, a.st_m2, a.st_base = sm
    a.M, a.N, a.K, a.ldw = M, N, K, ldw
    a.act, a.ngroups, a.max_n, a.vec = act, ngroups, max_n, vec | _mode_bit(mode)
    _set_conv(a, conv)
    if ALG is not None and groups is None:
        a_elems = (M // (conv.Ho * conv.Wo)) * conv.H * conv.W * conv.C if conv is not None else M * K
        _alg("gemm_nt", 4 * (a_elems + M * N * (1 + (R is not None) + (T is not None)) + N * K), 2 * M * N * K)
    L.check(L.lib().ws_gemm_nt(C.byref(a), L.stream_ptr()), "ws_gemm_nt")


def tn_splits(M: int):
    nsplit = max(1, min(64, M // 2048))
    rows = -(-M // nsplit)
    rows = -(-rows // 32) * 32
    nsplit = -(-M // rows)
    return nsplit, rows


def gemm_tn(*, G, g_rows: Rows, A, a_rows: Rows, M: int, slab, slab_stride: int, nsplit: int,
            rows_per_split: int, Nn=0, Kk=0, bslab=None, bslab_stride=0, out_off=0, bout_off=0,
            stats=None, gamma=None, beta=None, stat_map: Optional[StatMap] = None,
            shift_rows=0, seq_div=1, seq_len=1, groups=None, ngroups=0, max_n=0, max_k=0, vec=1,
            g_off=0, a_off=0, mode=None, conv: Optional[ConvView] = None):
    for n, t in (("G", G), ("A", A), ("slab", slab), ("bslab", bslab), ("stats", stats),
                 ("gamma", gamma), ("beta", beta)):
        _chk(t, n)
    a = L.GemmTNArgs()
    a.G, a.A, a.slab, a.bslab = _p(G, g_off), _p(A, a_off), _p(slab), _p(bslab)
    a.stats, a.gamma, a.beta = _p(stats), _p(gamma), _p(beta)
    a.groups = C.c_void_p(groups.data_ptr()) if groups is not None else None
    a.g_div, a.g_s1, a.g_s2 = g_rows
    a.a_div, a.a_s1, a.a_s2 = a_rows
    sm = stat_map or StatMap(1, 0, 1, 0, 0)
    a.st_div1, a.st_m1, a.st_div2, a.st_m2, a.st_base = sm
    a.slab_stride, a.bslab_stride, a.out_off, a.bout_off = slab_stride, bslab_stride, out_off, bout_off
    a.M, a.Nn, a.Kk, a.rows_per_split, a.nsplit = M, Nn, Kk, rows_per_split, nsplit
    a.shift_rows, a.seq_div, a.seq_len = shift_rows, seq_div, seq_len
    a.ngroups, a.max_n, a.max_k, a.vec = ngroups, max_n, max_k, vec | _mode_bit(mode)
    _set_conv(a, conv)
    if ALG is not None and groups is None:
        a_elems = (M // (conv.Ho * conv.Wo)) * conv.H * conv.W * conv.C if conv is not None else M * Kk
        _alg("gemm_tn", 4 * (a_elems + M * Nn + nsplit * Nn * Kk), 2 * M * Nn * Kk)
    L.check(L.lib().ws_gemm_tn(C.byref(a), L.stream_ptr()), "ws_gemm_tn")


CONV_WGRAD_MAXK = 768


def conv_wgrad_ok(Nn: int, conv: ConvView) -> bool:
    """The one-pass convolution weight-gradient kernel (conv_wgrad.hip) takes this shape."""
    return conv.mode == 0 and 4 <= Nn <= 32 and Nn % 4 == 0 and conv.C % 4 == 0 and conv.k <= 5


def conv_wgrad(*, G, ldg: int, X, M: int, Nn: int, conv: ConvView, slab, nsplit: int, tiles_per_split: int, bslab=None):
    for n, t in (("G", G), ("X", X), ("slab", slab), ("bslab", bslab)):
        _chk(t, n)
    a = L.ConvWgradArgs()
    a.G, a.X, a.slab, a.bslab = _p(G), _p(X), _p(slab), _p(bslab)
    Kk = conv.k * conv.k * conv.C
    a.ldg, a.slab_stride, a.bslab_stride = ldg, Nn * Kk, Nn
    a.M, a.Nn, a.nsplit, a.tiles_per_split = M, Nn, nsplit, tiles_per_split
    _set_conv(a, conv)
    _alg("gemm_tn", 4 * ((M // (conv.Ho * conv.Wo)) * conv.H * conv.W * conv.C + M * Nn + nsplit * Nn * Kk), 2 * M * Nn * Kk)
    L.check(L.lib().ws_conv_wgrad(C.byref(a), L.stream_ptr()), "ws_conv_wgrad")


def reduce_slabs(slab, nsplit: int, stride: int, count: int, out, w=0, ldo=0, out_off=0):
    _chk(slab, "slab")
    _chk(out, "out")
    L.check(L.lib().ws_reduce_slabs(_p(slab), nsplit, stride, count, _p(out, out_off), w, ldo,
                                    L.stream_ptr()), "ws_reduce_slabs")


def transpose(src, rows: int, cols: int, lds: int, dst, src_off=0, dst_off=0):
    _chk(src, "src")
    _chk(dst, "dst")
    L.check(L.lib().ws_transpose(_p(src, src_off), rows, cols, lds, _p(dst, dst_off), L.stream_ptr()),
            "ws_transpose")


class Geom(NamedTuple):
    ngroups: int
    gdiv: int
    gs1: int
    gs2: int
    rs: int
    L: int
    W: int
    nbands: int = 1
    band_w: Optional[torch.Tensor] = None
    band_off: Optional[torch.Tensor] = None

    def c(self):
        g = L.GroupsGeom()
        g.band_w = C.c_void_p(self.band_w.data_ptr()) if self.band_w is not None else None
        g.band_off = C.c_void_p(self.band_off.data_ptr()) if self.band_off is not None else None
        g.gs1, g.gs2, g.rs = self.gs1, self.gs2, self.rs
        g.ngroups, g.gdiv, g.L, g.W, g.nbands = self.ngroups, self.gdiv, self.L, self.W, self.nbands
        return g


def group_stats(x, geo: Geom, stats, eps=GN_EPS):
    _chk(x, "x")
    _chk(stats, "stats")
    g = geo.c()
    L.check(L.lib().ws_group_stats(_p(x), C.byref(g), eps, _p(stats), L.stream_ptr()), "ws_group_stats")


def _tab(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _word(t):
    """device pointer of an int32 word tensor (status / guard / counter / scale words), or NULL."""
    return C.c_void_p(t.data_ptr()) if t is not None else None


def gn_bwd_reduce(x, dxn, stats, geo: Geom, ab, gamma=None, gamma_tab=None):
    for n, t in (("x", x), ("dxn", dxn), ("stats", stats), ("ab", ab), ("gamma", gamma)):
        _chk(t, n)
    g = geo.c()
    L.check(L.lib().ws_gn_bwd_reduce(_p(x), _p(dxn), _p(stats), _p(gamma), _tab(gamma_tab),
                                     C.byref(g), _p(ab), L.stream_ptr()), "ws_gn_bwd_reduce")


def gn_bwd_apply(x, dxn, stats, ab, geo: Geom, dx, gamma=None, gamma_tab=None, res=None):
    for n, t in (("x", x), ("dxn", dxn), ("stats", stats), ("ab", ab), ("gamma", gamma),
                 ("res", res), ("dx", dx)):
        _chk(t, n)
    g = geo.c()
    L.check(L.lib().ws_gn_bwd_apply(_p(x), _p(dxn), _p(stats), _p(ab), _p(gamma), _tab(gamma_tab),
                                    _p(res), C.byref(g), _p(dx), L.stream_ptr()), "ws_gn_bwd_apply")


def gn_bwd_fused_ok(geo: Geom) -> bool:
    """Small single-band groups of 128-float rows (pBSRNN's band view): the one-pass GroupNorm backward applies.
    WESEP_GN_FUSED=0 keeps the three-kernel form."""
    return (os.environ.get("WESEP_GN_FUSED", "1") != "0" and geo.nbands == 1 and geo.W == 128 and geo.band_w is None
            and geo.band_off is None and 2 <= geo.L <= 32 and geo.L % 2 == 0 and geo.rs % 4 == 0
            and geo.gs1 % 4 == 0 and geo.gs2 % 4 == 0)


def gn_bwd_fused(x, dxn, stats, geo: Geom, gamma, dx, nwg: int, pslab, res=None, pout=None, counter=None, dxn2=None):
    """reduce + apply + parameter sums of the GroupNorm backward in one pass (norm.hip gn_bwd_fused_kernel);
    pslab [nwg, 2, 128]: per-workgroup shares of (dgamma, dbeta); pout [2, 128] (optional, with a zeroed int32 `counter`
    word): their sum, taken by the last workgroup of the launch.  dxn2 (ABI v19): a second addend of d(xn) -- the other LSTM
    direction's share when the BPTT wrote d(xn) itself (lstm_bwd(..., dxn=))."""
    for n, t in (("x", x), ("dxn", dxn), ("dxn2", dxn2), ("stats", stats), ("gamma", gamma), ("res", res), ("dx", dx),
                 ("pslab", pslab), ("pout", pout)):
        _chk(t, n)
    g = geo.c()
    L.check(L.lib().ws_gn_bwd_fused2(_p(x), _p(dxn), _p(dxn2), _p(stats), _p(gamma), _p(res), C.byref(g), nwg, _p(dx),
                                     _p(pslab), _p(pout), _word(counter), L.stream_ptr()), "ws_gn_bwd_fused")


def tree_groups(nblocks: int) -> int:
    """Extra partial rows / counter words of the two-level in-kernel sum (common.h ws_tree_sum256): one per 32 workgroups."""
    return -(-nblocks // 32)


def gn_bwd_apply_pg_ok(geo: Geom) -> bool:
    """Single-band groups of 128-float rows (the time view of ResRNN.norm): apply + parameter sums in one pass."""
    return (os.environ.get("WESEP_GN_FUSED", "1") != "0" and geo.nbands == 1 and geo.W == 128 and geo.band_w is None
            and geo.band_off is None and geo.rs % 4 == 0 and geo.gs1 % 4 == 0 and geo.gs2 % 4 == 0)


def gn_bwd_apply_pg(x, dxn, stats, ab, geo: Geom, dx, gamma, pslab, pout, counter, res=None):
    """GroupNorm backward pass 2 with the parameter sums (norm.hip gn_bwd_apply_pg_kernel): dx, and (dgamma, dbeta) in
    pout [2, 128] via pslab [ngroups, 2, 128] and the launch's last workgroup; counter: a zeroed int32 device word."""
    for n, t in (("x", x), ("dxn", dxn), ("stats", stats), ("ab", ab), ("gamma", gamma), ("res", res), ("dx", dx),
                 ("pslab", pslab), ("pout", pout)):
        _chk(t, n)
    g = geo.c()
    L.check(L.lib().ws_gn_bwd_apply_pg(_p(x), _p(dxn), _p(stats), _p(ab), _p(gamma), _p(res), C.byref(g), _p(dx),
                                       _p(pslab), _p(pout), _word(counter), L.stream_ptr()), "ws_gn_bwd_apply_pg")


def gn_param_grad(x, dxn, stats, geo: Geom, nsplit: int, slab):
    for n, t in (("x", x), ("dxn", dxn), ("stats", stats), ("slab", slab)):
        _chk(t, n)
    g = geo.c()
    L.check(L.lib().ws_gn_param_grad(_p(x), _p(dxn), _p(stats), C.byref(g), nsplit, _p(slab),
                                     L.stream_ptr()), "ws_gn_param_grad")


class SeqMap(NamedTuple):
    """sequence s, step t -> row (s // div) * s1 + (s % div) * s2 + t * step_rows.  nvalid (0 = nseq): sequences >= nvalid
    are padding -- zeros on the way into the blocked layout, dropped on the way out (wesep_hip.h ws_seqmap)."""
    nseq: int
    div: int
    s1: int
    s2: int
    step_rows: int
    L: int
    nvalid: int = 0


def lstm_mode(nseq: int) -> int:
    """'bf16x3' (default): split-bf16 recurrence, 32 sequences per workgroup; WESEP_LSTM=f32 selects
    the exact-fp32 MFMA kernels (32-sequence workgroups once they still cover the chip)."""
    if os.environ.get("WESEP_LSTM", "bf16x3") == "bf16x3":
        return L.LSTM_BF16X3
    return L.LSTM_F32_MT2 if nseq >= 8192 else L.LSTM_F32_MT1


def lstm_blk_mode(nseq: int) -> int:
    """Blocked-layout recurrence: 16-sequence workgroups when 32-sequence ones would cover well under
    the 256 CUs (pBSRNN's time view: 1024 sequences), else 32.  WESEP_LSTM_SEQS=16|32 overrides."""
    env = os.environ.get("WESEP_LSTM_SEQS")
    if env:
        return L.LSTM_BF16X3_BLK16 if env == "16" else L.LSTM_BF16X3_BLK
    return L.LSTM_BF16X3_BLK16 if 2 * (-(-nseq // 32)) <= 128 else L.LSTM_BF16X3_BLK


def lstm_pack(whh_f, whh_r, pack_fwd, pack_bwd, mode=L.LSTM_BF16X3):
    for n, t in (("whh_f", whh_f), ("whh_r", whh_r), ("pack_fwd", pack_fwd), ("pack_bwd", pack_bwd)):
        _chk(t, n)
    L.check(L.lib().ws_lstm_pack(_p(whh_f), _p(whh_r), _p(pack_fwd), _p(pack_bwd), mode,
                                 L.stream_ptr()), "ws_lstm_pack")


def lstm_pack_bwd_f8(whh_f, whh_r, pack_bwd):
    """BPTT pack of lstm_bwd(..., rfmt=2): fp16 hi + scaled-FP8 lo of 256 w (ws_lstm_pack_bwd_f8)."""
    for n, t in (("whh_f", whh_f), ("whh_r", whh_r), ("pack_bwd", pack_bwd)):
        _chk(t, n)
    L.check(L.lib().ws_lstm_pack_bwd_f8(_p(whh_f), _p(whh_r), _p(pack_bwd), L.stream_ptr()), "ws_lstm_pack_bwd_f8")


def lstm_pack_dx_f8(wcat, pack):
    """W_ih^T stream of lstm_bwd(..., dxn=): fp16 hi + scaled-FP8 lo of 256 w in 16x16x32 fragment order (ws_lstm_pack_dx_f8);
    wcat [2, 4H, 128] from lstm_cat_ih, pack: L.LSTM_DX_PACK_FLOATS floats."""
    for n, t in (("wcat", wcat), ("pack", pack)):
        _chk(t, n)
    L.check(L.lib().ws_lstm_pack_dx_f8(_p(wcat), _p(pack), L.stream_ptr()), "ws_lstm_pack_dx_f8")


def lstm_cat_ih(wih_f, wih_r, bih_f, bhh_f, bih_r, bhh_r, n_in, wcat, bcat):
    for t in (wih_f, wih_r, bih_f, bhh_f, bih_r, bhh_r, wcat, bcat):
        _chk(t, "lstm_cat_ih arg")
    L.check(L.lib().ws_lstm_cat_ih(_p(wih_f), _p(wih_r), _p(bih_f), _p(bhh_f), _p(bih_r), _p(bhh_r),
                                   n_in, _p(wcat), _p(bcat), L.stream_ptr()), "ws_lstm_cat_ih")


def _bptt_bytes(gfmt: int) -> int:
    """Algorithmic bytes of one BPTT cell (position, direction, unit): 4 saved gates in, c and d(h) in (fp32), 4 d(gates)
    out -- 40 B with fp32 gates / split-pair d(gates), 24 B with unorm16 gates and bf16 d(gates), 32 B for H2S."""
    return {L.GATES_F32: 40, L.GATES_H2: 24, L.GATES_H2S: 32, L.GATES_H2F: 24}[gfmt]


def gates_fmt() -> int:
    """Storage of the saved activated gates / d(gates) of the blocked-layout ResRNN (WS_GATES_*, wesep_hip.h):
    'h2' (default; WS_GATES_H2F): unorm16 gates, d(gates) as fp16 scaled by a power of two taken from the launch's
    max |d(hcat)| -- half the bytes of the step's largest buffer and of every pass over it, 11-bit d(gates);
    'h2b' (WS_GATES_H2): the same with bf16 d(gates) (8 bits: no scale word, two MFMAs per product in the consumers);
    'h2s': unorm16 gates, d(gates) as full split pairs in a separate buffer; 'f32': the ABI <= 14 format (fp32 gates,
    split-pair d(gates) in place).  WESEP_GATES selects."""
    return {"h2": L.GATES_H2F, "h2b": L.GATES_H2, "h2s": L.GATES_H2S, "f32": L.GATES_F32}[os.environ.get("WESEP_GATES", "h2")]


def blh_floats(nblocks: int, C_: int) -> int:
    """float32 elements of storage behind a BLH(C_) buffer of `nblocks` blocks (2-byte elements; the wrappers take
    float32 tensors as opaque storage)."""
    return nblocks * 32 * C_ // 2


def _lstm_args(gates, cbuf, hcat, wpack, sm: SeqMap, mode, dhcat=None, run_if=None, gfmt=0, gates_in=None, dgates=None,
               amax=None, rfmt=0, dxn=None, wxpack=None):
    for n, t in (("gates", gates), ("cbuf", cbuf), ("hcat", hcat), ("wpack", wpack), ("dhcat", dhcat),
                 ("gates_in", gates_in), ("dgates", dgates), ("dxn", dxn), ("wxpack", wxpack)):
        _chk(t, n)
    a = L.LstmArgs()
    a.gates, a.cbuf, a.hcat, a.dhcat, a.wpack = _p(gates), _p(cbuf), _p(hcat), _p(dhcat), _p(wpack)
    a.sq_s1, a.sq_s2, a.step_rows = sm.s1, sm.s2, sm.step_rows
    a.nseq, a.sq_div, a.L, a.mode = sm.nseq, sm.div, sm.L, mode
    a.run_if = C.c_void_p(run_if.data_ptr()) if run_if is not None else None
    a.gates_in, a.dgates, a.gfmt, a.rfmt = _p(gates_in), _p(dgates), gfmt, rfmt
    a.amax = C.c_void_p(amax.data_ptr()) if amax is not None else None
    if dxn is not None:     # ABI v19: [2, P, 128], one plain-row buffer per direction
        if dxn.dim() != 3 or dxn.shape[0] != 2 or dxn.shape[2] != 128 or getattr(sm, "nvalid", 0):
            raise L.WesepHipError(f"lstm_bwd: dxn must be [2, P, 128] on a sequence map without padding sequences, got {tuple(dxn.shape)}")
        a.dxn, a.dxn_dir_stride, a.wxpack = _p(dxn), dxn.shape[1] * 128, _p(wxpack)
    return a


def lstm_fwd(gates, cbuf, hcat, wpack, sm: SeqMap, mode=L.LSTM_BF16X3, run_if=None, gfmt=0, gates_in=None):
    """run_if: optional 1-element int32 device tensor; the launch is a no-op unless it is non-zero at kernel start.
    gfmt != 0 (blocked-layout modes): pre-activations from `gates_in` (fp32 BL), `gates` receives unorm16 (BLH)."""
    a = _lstm_args(gates, cbuf, hcat, wpack, sm, mode, run_if=run_if, gfmt=gfmt, gates_in=gates_in)
    if run_if is None:     # fp32 pre-activations in, gates (fp32 in place / unorm16) + c + h out
        u = ALG_LSTM_UNITS or L.LSTM_H
        _alg("lstm_fwd", sm.nseq * sm.L * 2 * u * (16 + (2 if gfmt else 4) * 4 + 8), 2 * sm.nseq * sm.L * 2 * 4 * u * u)
    L.check(L.lib().ws_lstm_fwd(C.byref(a), L.stream_ptr()), "ws_lstm_fwd")


def lstm_bwd(gates, cbuf, hcat, dhcat, wpack, sm: SeqMap, mode=L.LSTM_BF16X3, gfmt=0, dgates=None, run_if=None, amax=None,
             rfmt=0, dxn=None, wxpack=None):
    """run_if (blocked-layout modes): 1-element int32 device tensor; the launch is a no-op unless it is non-zero at
    kernel start -- the predicated fall-back behind lstm_bwd_pair.  dgates: out-of-place d(gates) (required for
    GATES_H2S; optional BLH buffer for GATES_H2, which then leaves the saved gates intact).  rfmt = 2 (LSTM_BF16X3_BLK with
    GATES_H2F only): fp16 recurrence on fp16 + FP8 weights, wpack from lstm_pack_bwd_f8.  dxn [2, P, 128] + wxpack
    (lstm_pack_dx_f8; rfmt 2 only, ABI v19): the kernel also writes d(normalised input) per direction -- plain rows at the
    sequence map's positions -- so that gemm_b2p over d(gates) is not needed (gn_bwd_fused(..., dxn2=) adds the two)."""
    a = _lstm_args(gates, cbuf, hcat, wpack, sm, mode, dhcat, gfmt=gfmt, dgates=dgates, run_if=run_if, amax=amax, rfmt=rfmt,
                   dxn=dxn, wxpack=wxpack)
    # per (position, direction, unit): read 4 gates + c + dh, write 4 d(gates); 2 * 4H * H MACs per position
    if run_if is None:
        _alg("lstm_bwd", _bptt_bytes(gfmt) * sm.nseq * sm.L * 2 * (ALG_LSTM_UNITS or L.LSTM_H),
             2 * sm.nseq * sm.L * 2 * 4 * (ALG_LSTM_UNITS or L.LSTM_H) ** 2)
    L.check(L.lib().ws_lstm_bwd(C.byref(a), L.stream_ptr()), "ws_lstm_bwd")


_CU_COUNT = {}


def cu_count(device) -> int:
    key = (device.type, device.index)
    if key not in _CU_COUNT:
        _CU_COUNT[key] = torch.cuda.get_device_properties(device).multi_processor_count
    return _CU_COUNT[key]


def lstm_cluster_ok(sm: SeqMap, device) -> bool:
    """The weight-stationary cluster recurrence needs 64-sequence clusters that are all co-resident
    (8 workgroups per 64 sequences per direction <= CUs) and pays off for long sequences only.
    WESEP_LSTM_CLUSTER=0 disables it."""
    if os.environ.get("WESEP_LSTM_CLUSTER", "1") == "0":
        return False
    return sm.nseq % 64 == 0 and (sm.nseq // 32) * 8 <= cu_count(device) and sm.L >= 64


class _ClusterScratch:
    """Per (device, stream) scratch of the cluster recurrences: exchange buffer + flag words (grow-only; launches on
    one stream are ordered, so they share it), and the sticky status words [forward, BPTT] with their pinned host
    mirror for the asynchronous check."""

    def __init__(self, device):
        self.device = device
        self.xchg = None
        self.flags = None
        self.status = torch.zeros(2, device=device, dtype=torch.int32)
        self.host = torch.zeros(2, dtype=torch.int32)
        if device.type == "cuda" and torch.cuda.is_available():   # (dry-run harnesses fake is_cuda on CPU tensors)
            self.host = self.host.pin_memory()
        self.event = None
        self.fallbacks = 0

    def get(self, xchg_floats: int, nflags: int):
        if self.xchg is None or self.xchg.numel() < xchg_floats:
            self.xchg = torch.empty(xchg_floats, device=self.device, dtype=torch.float32)
        if self.flags is None or self.flags.numel() < nflags:
            self.flags = torch.empty(nflags, device=self.device, dtype=torch.int32)
        return self.xchg, self.flags


_CLUSTER_SCRATCH = {}


def _cluster_scratch(device) -> _ClusterScratch:
    key = (device.type, device.index, L.stream_ptr().value)
    if key not in _CLUSTER_SCRATCH:
        _CLUSTER_SCRATCH[key] = _ClusterScratch(device)
    return _CLUSTER_SCRATCH[key]


class StepFence:
    """Bounds how far the host may run ahead of the GPU, in optimizer steps.

    The headline step takes the host 15-25 ms to enqueue and the GPU 93 ms to run, and nothing in it waits for the device.
    Left alone the host is soon several steps ahead -- and every block that was handed to the side stream
    (`record_stream`: d(gates), xn, hcat, d(out) of twelve ResRNNs, 49 GB per step) cannot be reused by the caching allocator
    before the GPU has actually passed it, so each step of run-ahead costs another 49 GB of hipMalloc (seen: 75 -> 265 GB
    reserved within five steps, single hipMalloc calls of 2.8-3.7 s, bench lines of 344-850 ms per step with every kernel at
    its usual duration: profiles/r06_c52_diag.txt).  `fence()` records an event behind the optimizer's last launch and waits
    for the event of `depth` steps earlier: with depth 1 the host enqueues step n + 1 while the GPU runs step n -- the queue is
    never empty -- and reserved memory settles after two steps.  WESEP_RUN_AHEAD=<depth> (0: synchronise every step; a
    negative value disables the fence)."""

    def __init__(self, depth=None, make_event=None):
        self.depth = int(os.environ.get("WESEP_RUN_AHEAD", "1")) if depth is None else int(depth)
        self._make = make_event
        self._ring = []

    def fence(self):
        if self.depth < 0:
            return
        ev = self._make() if self._make is not None else torch.cuda.Event(blocking=True)
        ev.record()
        self._ring.append(ev)
        while len(self._ring) > self.depth:
            self._ring.pop(0).synchronize()


_STEP_FENCE = {}


def step_fence(device):
    """One optimizer step of `device`'s current stream is enqueued: wait for the step before the previous one (StepFence)."""
    if device.type != "cuda" or not torch.cuda.is_available():
        return
    key = (device.index, L.stream_ptr().value)
    if key not in _STEP_FENCE:
        _STEP_FENCE[key] = StepFence()
    with torch.cuda.device(device):
        _STEP_FENCE[key].fence()


def poll_cluster_status(device, block=False):
    """Asynchronous check of the cluster recurrences' sticky status words on the current stream: evaluates the
    8-byte device->pinned-host copy started by the previous call once its event has completed and starts the next
    one (`block`: copy now and wait).  A forward timeout was already repaired on the device by the predicated
    streaming kernels (wesep_hip.h, ws_lstm_fwd_cluster): it is counted and reported once.  A BPTT timeout has no
    device-side repair (the kernel works in place): WesepHipError.  Returns the number of repaired forward timeouts."""
    key = (device.type, device.index, L.stream_ptr().value)
    if key not in _CLUSTER_SCRATCH or not torch.cuda.is_available():
        return 0          # no cluster launch on this stream so far (or a GPU-less dry-run harness)
    sc = _CLUSTER_SCRATCH[key]

    def start():
        sc.host.copy_(sc.status, non_blocking=True)
        sc.event = torch.cuda.Event()
        sc.event.record()

    def evaluate():
        sc.event.synchronize()
        sc.event = None
        fwd_to, bwd_to = int(sc.host[0]), int(sc.host[1])
        if fwd_to or bwd_to:
            sc.status.zero_()
        if fwd_to:
            if sc.fallbacks == 0:
                import warnings
                warnings.warn("lstm_fwd_cluster / lstm_bwd_pair: a bounded wait timed out (workgroups not co-resident: "
                              "another stream or process holds CUs); the layer was recomputed by the streaming kernels.  "
                              "WESEP_LSTM_CLUSTER=0 / WESEP_LSTM_PAIR_BWD=0 avoid these kernels altogether", RuntimeWarning)
            sc.fallbacks += 1
        if bwd_to:
            raise L.WesepHipError(
                "lstm_bwd_pair / lstm_bwd_cluster: a bounded wait timed out (the workgroups were not co-resident: another "
                "stream or process holds CUs); d(gates) of that launch are NaN-poisoned.  WESEP_LSTM_PAIR_BWD=0 (and "
                "WESEP_LSTM_CLUSTER_BWD unset) selects the streaming BPTT kernel")

    if sc.event is not None and (block or sc.event.query()):
        evaluate()
    if block:
        start()
        evaluate()
    elif sc.event is None:
        start()
    return sc.fallbacks


def lstm_fwd_cluster(gates, cbuf, hcat, whh_f, whh_r, sm: SeqMap, status=None, dbg=0, gfmt=0, gates_in=None):
    """Forward recurrence on the blocked layout with W_hh resident in registers across clusters of 8
    workgroups (lstm_cluster.hip).  Returns the launch's timeout word (a 1-element int32 view of the flag scratch):
    pass it as `run_if` to gemm_p2b + lstm_fwd behind this call -- the predicated fall-back."""
    for n, t in (("gates", gates), ("cbuf", cbuf), ("hcat", hcat), ("whh_f", whh_f), ("whh_r", whh_r), ("gates_in", gates_in)):
        _chk(t, n)
    ncl = sm.nseq // 32
    sc = _cluster_scratch(gates.device)
    xchg, flags = sc.get(ncl * 2 * 8 * 8192 // 4, ncl * 8 + 8)
    a = L.LstmClusterArgs()
    a.gates, a.cbuf, a.hcat, a.whh_f, a.whh_r = _p(gates), _p(cbuf), _p(hcat), _p(whh_f), _p(whh_r)
    a.gfmt, a.gates_in = gfmt, _p(gates_in)
    a.xchg, a.flags = C.c_void_p(xchg.data_ptr()), C.c_void_p(flags.data_ptr())
    a.status = C.c_void_p((status if status is not None else sc.status).data_ptr())
    a.nseq, a.L, a.dbg = sm.nseq, sm.L, dbg
    u = ALG_LSTM_UNITS or L.LSTM_H
    _alg("lstm_fwd", sm.nseq * sm.L * 2 * u * (16 + (2 if gfmt else 4) * 4 + 8), 2 * sm.nseq * sm.L * 2 * 4 * u * u)
    L.check(L.lib().ws_lstm_fwd_cluster(C.byref(a), L.stream_ptr()), "ws_lstm_fwd_cluster")
    return flags[ncl * 8:ncl * 8 + 1]


_FALLBACK_SCRATCH = {}


def fallback_scratch(device, nfloats: int) -> torch.Tensor:
    """The fp32 pre-activation buffer of the predicated streaming fall-back behind ws_lstm_fwd_cluster2 (per device and
    stream; grow-only).  After a clean cluster launch nobody touches it, so the six time-view layers of a step share one
    buffer that is allocated once -- as a fresh 4.2 GB torch allocation per layer it made the caching allocator hunt for a
    block that size twelve times per SSA step."""
    key = (device.type, device.index, L.stream_ptr().value if device.type == "cuda" and torch.cuda.is_available() else 0)
    buf = _FALLBACK_SCRATCH.get(key)
    if buf is None or buf.numel() < nfloats:
        buf = _FALLBACK_SCRATCH[key] = torch.empty(nfloats, device=device, dtype=torch.float32)
    return buf[:nfloats]


def lstm_cluster2_on() -> bool:
    """Second-generation cluster forward (lstm_cluster2.hip, ABI v17: fp16 h, x-projection fused in from the split-pair
    normalised input, data-tagged hand-off) for the time view of the 2-byte gate formats; WESEP_LSTM_CLUSTER2=0 keeps the
    round-1..4 kernel behind ws_gemm_p2b's fp32 pre-activations."""
    return os.environ.get("WESEP_LSTM_CLUSTER2", "1") != "0"


def cluster2_rfmt() -> int:
    """ws_lstm_cluster2_args.rfmt (ABI v20): 1 (WESEP_CLUSTER2_F8=1) = the lo term of the time view's recurrent product on the
    block-scaled FP8 matrix instruction; 0 = fp16 hi / lo (round 5)."""
    return 1 if os.environ.get("WESEP_CLUSTER2_F8", "0") == "1" else 0


def lstm_fwd_cluster2(gates, cbuf, hcat, xn, wcat, bcat, whh_f, whh_r, sm: SeqMap, status=None, dbg=0, dbg_buf=None, rfmt=None):
    """ws_lstm_fwd_cluster2: gates (unorm16 BLH), cbuf, hcat (BLS) <- the BLSTM forward of the blocked-layout sequences from
    the normalised input xn (BL(128) of BLS pairs: gemm_p2b's A_bl), W_ih / biases as ws_lstm_cat_ih leaves them and the fp32
    W_hh.  Returns the launch's time-out word: pass it as `run_if` to gemm_p2b + lstm_fwd behind this call -- the predicated
    fall-back."""
    for n, t in (("gates", gates), ("cbuf", cbuf), ("hcat", hcat), ("xn", xn), ("wcat", wcat), ("bcat", bcat),
                 ("whh_f", whh_f), ("whh_r", whh_r)):
        _chk(t, n)
    ncl = sm.nseq // 32
    sc = _cluster_scratch(gates.device)
    xchg, flags = sc.get(ncl * 2 * 8 * 4096 // 4, ncl * 8 + 8)
    a = L.LstmCluster2Args()
    a.gates, a.cbuf, a.hcat, a.xn, a.wcat, a.bcat = _p(gates), _p(cbuf), _p(hcat), _p(xn), _p(wcat), _p(bcat)
    a.whh_f, a.whh_r = _p(whh_f), _p(whh_r)
    tw = flags[ncl * 8:ncl * 8 + 1]
    a.xchg, a.tword = C.c_void_p(xchg.data_ptr()), C.c_void_p(tw.data_ptr())
    a.status = C.c_void_p((status if status is not None else sc.status).data_ptr())
    a.nseq, a.L, a.dbg = sm.nseq, sm.L, dbg
    a.rfmt = cluster2_rfmt() if rfmt is None else rfmt
    a.dbg_buf = C.c_void_p(dbg_buf.data_ptr()) if dbg_buf is not None else None
    u = ALG_LSTM_UNITS or L.LSTM_H     # (as lstm_fwd_fused: unorm16 gates + c + h out, the split-pair input in)
    _alg("lstm_fwd", sm.nseq * sm.L * (2 * u * 16 + 4 * 128), 2 * sm.nseq * sm.L * 2 * 4 * u * (u + 128))
    L.check(L.lib().ws_lstm_fwd_cluster2(C.byref(a), L.stream_ptr()), "ws_lstm_fwd_cluster2")
    return tw


def lstm_bwd_cluster(gates, cbuf, dhcat, whh_f, whh_r, sm: SeqMap, status=None, dbg=0):
    """BPTT on the blocked layout over clusters of 8 workgroups (lstm_cluster.hip); gates: activated
    gates in, d(pre-activation gates) out.  Returns the launch's timeout word (see lstm_fwd_cluster); there is no
    device-side fall-back (in place), poll_cluster_status raises."""
    for n, t in (("gates", gates), ("cbuf", cbuf), ("dhcat", dhcat), ("whh_f", whh_f), ("whh_r", whh_r)):
        _chk(t, n)
    ncl = sm.nseq // 32
    sc = _cluster_scratch(gates.device)
    xchg, flags = sc.get(ncl * 2 * 64 * 8192 // 4, ncl * 8 + 8)
    a = L.LstmClusterArgs()
    a.gates, a.cbuf, a.dhcat, a.whh_f, a.whh_r = _p(gates), _p(cbuf), _p(dhcat), _p(whh_f), _p(whh_r)
    a.xchg, a.flags = C.c_void_p(xchg.data_ptr()), C.c_void_p(flags.data_ptr())
    a.status = C.c_void_p(status.data_ptr() if status is not None else sc.status.data_ptr() + 4)
    a.nseq, a.L, a.dbg = sm.nseq, sm.L, dbg
    L.check(L.lib().ws_lstm_bwd_cluster(C.byref(a), L.stream_ptr()), "ws_lstm_bwd_cluster")
    return flags[ncl * 8:ncl * 8 + 1]


def lstm_pair_ok(sm: SeqMap, device) -> bool:
    """The pair BPTT (lstm_pair.hip: two workgroups per (32-sequence tile, direction), W_hh split by gate rows, hi plane
    resident) wants views with few, long sequences: both members of every pair co-resident on at most HALF of the CUs
    (the other half is what the side stream's weight-gradient GEMMs run on) and enough steps to pay for loading the
    resident plane.  WESEP_LSTM_PAIR_BWD=0 disables it (the 16-sequence streaming kernel then runs the time view)."""
    if os.environ.get("WESEP_LSTM_PAIR_BWD", "1") == "0":
        return False
    return 4 * (-(-sm.nseq // 32)) <= cu_count(device) // 2 and sm.L >= 64


def lstm_pack_pair(whh_f, whh_r, pack, f16=False):
    for n, t in (("whh_f", whh_f), ("whh_r", whh_r), ("pack", pack)):
        _chk(t, n)
    # f16: the pair BPTT's rfmt -- 0 bf16 hi / lo, 1 fp16 hi / lo of 256 w, 2 fp16 hi + FP8 lo (block-scaled) of 256 w, 3 the same
    # codes as operand fragments of the FP8 matrix instruction (ABI v20)
    fn = (L.lib().ws_lstm_pack_pair, L.lib().ws_lstm_pack_pair_f16, L.lib().ws_lstm_pack_pair_f8,
          L.lib().ws_lstm_pack_pair_f8mx)[int(f16)]
    L.check(fn(_p(whh_f), _p(whh_r), _p(pack), L.stream_ptr()), "ws_lstm_pack_pair")


def lstm_bwd_pair(gates, cbuf, dhcat, wpack, sm: SeqMap, status=None, dbg=0, dbg_buf=None, gfmt=0, dgates=None,
                  repairable=False, amax=None, rfmt=0):
    """BPTT on the blocked layout over pairs of workgroups (lstm_pair.hip); gates: activated gates in,
    d(pre-activation gates) (BLS) out.  Returns the launch's timeout word; in place, so there is no device-side
    fall-back: poll_cluster_status raises (one step late, without a host sync) when a bounded wait timed out."""
    for n, t in (("gates", gates), ("cbuf", cbuf), ("dhcat", dhcat), ("wpack", wpack), ("dgates", dgates)):
        _chk(t, n)
    npair = 2 * (-(-sm.nseq // 32))
    sc = _cluster_scratch(gates.device)
    xchg, flags = sc.get(npair * 65536 // 4, npair * 8 + 8)
    a = L.LstmPairArgs()
    a.gates, a.cbuf, a.dhcat, a.wpack = _p(gates), _p(cbuf), _p(dhcat), _p(wpack)
    a.xchg, a.flags = C.c_void_p(xchg.data_ptr()), C.c_void_p(flags.data_ptr())
    # sticky status words: [0] time-outs a predicated fall-back repairs on the device (counted, reported once),
    # [1] time-outs of an in-place BPTT (fatal: poll_cluster_status raises, FusedClipAdam skips the update on the device)
    a.status = C.c_void_p(status.data_ptr() if status is not None else sc.status.data_ptr() + (0 if repairable else 4))
    a.nseq, a.L, a.dbg = sm.nseq, sm.L, dbg
    a.dbg_buf = C.c_void_p(dbg_buf.data_ptr()) if dbg_buf is not None else None
    a.gfmt, a.dgates = gfmt, _p(dgates)
    a.amax = C.c_void_p(amax.data_ptr()) if amax is not None else None
    a.rfmt = rfmt           # 1 / 2: fp16 recurrence (wpack from lstm_pack_pair(..., f16=rfmt); WS_GATES_H2F only)
    _alg("lstm_bwd", _bptt_bytes(gfmt) * sm.nseq * sm.L * 2 * (ALG_LSTM_UNITS or L.LSTM_H),
             2 * sm.nseq * sm.L * 2 * 4 * (ALG_LSTM_UNITS or L.LSTM_H) ** 2)
    L.check(L.lib().ws_lstm_bwd_pair(C.byref(a), L.stream_ptr()), "ws_lstm_bwd_pair")
    return flags[npair * 8:npair * 8 + 1]


class BandTables:
    """Device-resident band tables shared by the STFT / norm / GEMM launches."""

    def __init__(self, band_width, device):
        bw = np.asarray(band_width, dtype=np.int32)
        f0 = np.concatenate([[0], np.cumsum(bw)[:-1]]).astype(np.int32)
        self.nband = len(bw)
        self.nbins = int(bw.sum())
        self.bw_host, self.f0_host = bw, f0
        bob = np.repeat(np.arange(self.nband, dtype=np.int32), bw)
        to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
        self.band_of_bin, self.f0, self.bw = to(bob), to(f0), to(bw)
        self.bw2, self.off2 = to(2 * bw), to(2 * f0)

    def c(self):
        b = L.Bands()
        b.band_of_bin = C.c_void_p(self.band_of_bin.data_ptr())
        b.band_f0 = C.c_void_p(self.f0.data_ptr())
        b.band_bw = C.c_void_p(self.bw.data_ptr())
        b.nband, b.nbins = self.nband, self.nbins
        return b


def stft_bandsplit(wav, bands: BandTables, xbs):
    _chk(wav, "wav")
    _chk(xbs, "xbs")
    R, T = wav.shape
    b = bands.c()
    L.check(L.lib().ws_stft_bandsplit(_p(wav), R, T, C.byref(b), _p(xbs), L.stream_ptr()),
            "ws_stft_bandsplit")


def mask_istft_frames(xbs, mask3, R, Tf, bands: BandTables, frames):
    for n, t in (("xbs", xbs), ("mask3", mask3), ("frames", frames)):
        _chk(t, n)
    b = bands.c()
    L.check(L.lib().ws_mask_istft_frames(_p(xbs), _p(mask3), R, Tf, C.byref(b), _p(frames),
                                         L.stream_ptr()), "ws_mask_istft_frames")


def istft_ola(frames, R, Tf, T, wav):
    _chk(frames, "frames")
    _chk(wav, "wav")
    L.check(L.lib().ws_istft_ola(_p(frames), R, Tf, T, _p(wav), L.stream_ptr()), "ws_istft_ola")


def mask_istft_bwd(dwav, xbs, mask3, R, Tf, T, bands: BandTables, dmask3):
    for n, t in (("dwav", dwav), ("xbs", xbs), ("mask3", mask3), ("dmask3", dmask3)):
        _chk(t, n)
    b = bands.c()
    L.check(L.lib().ws_mask_istft_bwd(_p(dwav), _p(xbs), _p(mask3), R, Tf, T, C.byref(b), _p(dmask3),
                                      L.stream_ptr()), "ws_mask_istft_bwd")


def affine_fwd(z, a, b, a0, rows, rows_per_r, N, out):
    for n, t in (("z", z), ("a", a), ("b", b), ("out", out)):
        _chk(t, n)
    L.check(L.lib().ws_affine_fwd(_p(z), _p(a), _p(b), a0, rows, rows_per_r, N, _p(out),
                                  L.stream_ptr()), "ws_affine_fwd")


def affine_bwd(dz, z_in, a, a0, rows, rows_per_r, N, nsplit, dz_in, da_slab, db_slab, da=None, db=None, counter=None):
    """da / db [R, N] (optional, with a zeroed int32 `counter` word): the slabs summed over the splits by the last workgroup
    of the launch instead of by ws_reduce_slabs launches."""
    for n, t in (("dz", dz), ("z_in", z_in), ("a", a), ("dz_in", dz_in), ("da_slab", da_slab), ("db_slab", db_slab),
                 ("da", da), ("db", db)):
        _chk(t, n)
    L.check(L.lib().ws_affine_bwd(_p(dz), _p(z_in), _p(a), a0, rows, rows_per_r, N, nsplit, _p(dz_in),
                                  _p(da_slab), _p(db_slab), _p(da), _p(db), _word(counter), L.stream_ptr()), "ws_affine_bwd")


def sisdr_fwd(est, tgt, rowstat, loss, eps=1e-8):
    for n, t in (("est", est), ("tgt", tgt), ("rowstat", rowstat), ("loss", loss)):
        _chk(t, n)
    R, T = est.shape
    L.check(L.lib().ws_sisdr_fwd(_p(est), _p(tgt), R, T, eps, _p(rowstat), _p(loss), L.stream_ptr()),
            "ws_sisdr_fwd")


def sisdr_bwd(est, tgt, rowstat, gout, dest):
    for n, t in (("est", est), ("tgt", tgt), ("rowstat", rowstat), ("gout", gout), ("dest", dest)):
        _chk(t, n)
    R, T = est.shape
    L.check(L.lib().ws_sisdr_bwd(_p(est), _p(tgt), _p(rowstat), _p(gout), R, T, _p(dest),
                                 L.stream_ptr()), "ws_sisdr_bwd")


def grad_norms(tab, ntensors, norms, guard=None):
    """guard: optional int32 device tensor: [0] (zeroed by the caller before the step's first launch) is set to 1 when a
    norm is NaN / Inf."""
    L.check(L.lib().ws_grad_norms(C.c_void_p(tab.data_ptr()), ntensors, _p(norms), _word(guard), L.stream_ptr()),
            "ws_grad_norms")


def bptt_status_word(device):
    """The sticky status word of the in-place BPTT launches (pair / cluster kernels without a device-side fall-back) of
    the current stream as a 1-element int32 view, or None when no such launch has happened on it."""
    key = (device.type, device.index, L.stream_ptr().value)
    sc = _CLUSTER_SCRATCH.get(key)
    return sc.status[1:2] if sc is not None else None


_WEIGHT_EPOCH = [0]


def weight_epoch() -> int:
    """Counter of raw-pointer parameter updates (ws_clip_adam_step writes parameters without touching torch's version
    counters): part of the signature of every cached weight pack (functional.PackCache)."""
    return _WEIGHT_EPOCH[0]


def bump_weight_epoch():
    _WEIGHT_EPOCH[0] += 1


def clip_adam_step(tab, ntensors, norms, clip, lr, beta1, beta2, eps, weight_decay, step, clip_only=False, skip=(None, None),
                   step_lag=None):
    """skip: up to two 1-element int32 device tensors; the launch does nothing when one of them is non-zero at kernel
    start (the guard of grad_norms, the BPTT status word: wesep_hip.h).  step_lag (1-element int32 device tensor, ABI v17):
    skipped steps the host's `step` still counts -- the bias corrections use step - lag."""
    if not clip_only:
        bump_weight_epoch()
    L.check(L.lib().ws_clip_adam_step(C.c_void_p(tab.data_ptr()), ntensors, _p(norms), clip, lr, beta1,
                                      beta2, eps, weight_decay, step, int(clip_only), _word(skip[0]), _word(skip[1]),
                                      _word(step_lag), L.stream_ptr()),
            "ws_clip_adam_step")


def guard_commit(guard, skip1=None):
    """ws_guard_commit: closes the step's book-keeping on the device (guard: 4-element int32 device tensor -- [0] this
    step's skip word, [1] skipped steps so far, [2] consecutive skipped steps, [3] bias-correction lag)."""
    L.check(L.lib().ws_guard_commit(_word(guard), _word(skip1), L.stream_ptr()), "ws_guard_commit")


def debug_occupy(nblocks: int, usec: int, stop=None):
    """Test support: holds `nblocks` CUs (112 KB of LDS each) for `usec` microseconds on the current stream, or until the
    1-element int32 device tensor `stop` becomes non-zero (ws_debug_occupy)."""
    L.check(L.lib().ws_debug_occupy(nblocks, usec, _word(stop), None, L.stream_ptr()), "ws_debug_occupy")


def prof_enable(on: bool):
    L.check(L.lib().ws_prof_enable(int(on)), "ws_prof_enable")


def prof_collect(kind: int):
    ms = C.c_double(0.0)
    n = C.c_longlong(0)
    L.check(L.lib().ws_prof_collect(kind, C.byref(ms), C.byref(n)), "ws_prof_collect")
    return ms.value, n.value


# ---------------------------------------------------------------------------------------------
# Blocked layout BL (include/wesep_hip.h): helpers for tests / probes.  The product path never
# converts: the GEMM kernels read and write BL directly.
# ---------------------------------------------------------------------------------------------
def bl_num_blocks(sm: SeqMap) -> int:
    return -(-sm.nseq // 32) * sm.L


def bl_positions(sm: SeqMap, device):
    """(pos, valid): position (row of the Z layout) of every BL slot, block-major; padded slots
    (sequence index >= nseq) are invalid and reported as position 0."""
    ntile = -(-sm.nseq // 32)
    seq = torch.arange(ntile * 32, device=device).view(ntile, 1, 32)
    step = torch.arange(sm.L, device=device).view(1, sm.L, 1)
    nv = sm.nvalid or sm.nseq
    valid = (seq < nv).expand(ntile, sm.L, 32)
    sq = seq.clamp(max=nv - 1)
    pos = torch.div(sq, sm.div, rounding_mode="floor") * sm.s1 + (sq % sm.div) * sm.s2 + step * sm.step_rows
    return (pos * valid).reshape(-1), valid.reshape(-1)


def bls_pack(x: torch.Tensor) -> torch.Tensor:
    """fp32 values -> split-bf16 storage BLS (include/wesep_hip.h): same shape / dtype, each element's BITS are
    bf16 hi << 16 | bf16 lo with hi = bf16(x), lo = bf16(x - hi) (round to nearest even, like the kernels)."""
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    bits = (hi.view(torch.int16).to(torch.int32) << 16) | (lo.view(torch.int16).to(torch.int32) & 0xffff)
    return bits.view(torch.float32)


def bls_unpack(x: torch.Tensor) -> torch.Tensor:
    """BLS -> fp32 values hi + lo (exact: the sum of the two terms is representable)."""
    bits = x.contiguous().view(torch.int32)
    return (bits & -65536).view(torch.float32) + (bits << 16).view(torch.float32)


def to_blocked(x: torch.Tensor, sm: SeqMap, split=False) -> torch.Tensor:
    """[P][C] plain rows -> BL(C) [nblk][C/4][32][4]; padded slots are zero.  split: elements in BLS form (h,
    d(gates), the normalised input -- whatever the blocked GEMMs / the fused recurrence consume)."""
    pos, valid = bl_positions(sm, x.device)
    C_ = x.shape[1]
    rows = x[pos] * valid[:, None].to(x.dtype)
    out = rows.view(-1, 32, C_ // 4, 4).permute(0, 2, 1, 3).contiguous()
    return bls_pack(out) if split else out


def from_blocked(xb: torch.Tensor, sm: SeqMap, P: int, split=False) -> torch.Tensor:
    """BL(C) -> [P][C] plain rows (padded slots dropped).  split: the buffer holds BLS elements."""
    if split:
        xb = bls_unpack(xb)
    pos, valid = bl_positions(sm, xb.device)
    nblk, Cq = xb.shape[0], xb.shape[1]
    rows = xb.permute(0, 2, 1, 3).reshape(nblk * 32, Cq * 4)
    out = torch.zeros(P, Cq * 4, device=xb.device, dtype=xb.dtype)
    out[pos[valid]] = rows[valid]
    return out


# ---- BLH: the BL index formula on 2-byte elements (WS_GATES_H2 / H2S, wesep_hip.h): test / tool side views ----------
def blh_bf16_pack(xb: torch.Tensor) -> torch.Tensor:
    """BL-shaped fp32 [nblk][C/4][32][4] -> BLH buffer of bf16 elements (round to nearest even = the hi term of the split
    pair), returned as float32 storage [nblk][C/4][32][2]."""
    return xb.to(torch.bfloat16).contiguous().view(torch.float32)


def blh_f16_pack(xb: torch.Tensor) -> torch.Tensor:
    """BL-shaped fp32 [nblk][C/4][32][4] -> BLH buffer of fp16 elements (round to nearest even), as float32 storage."""
    return xb.to(torch.float16).contiguous().view(torch.float32)


def blh_f16_unpack(buf: torch.Tensor, nblk: int, C_: int) -> torch.Tensor:
    """BLH buffer of fp16 elements (float32 storage) -> BL-shaped fp32 [nblk][C/4][32][4]."""
    return buf.reshape(-1)[: nblk * 32 * C_ // 2].view(torch.float16).view(nblk, C_ // 4, 32, 4).float()


def blh_bf16_unpack(buf: torch.Tensor, nblk: int, C_: int) -> torch.Tensor:
    """BLH buffer of bf16 elements (float32 storage) -> BL-shaped fp32 [nblk][C/4][32][4]."""
    return buf.reshape(-1)[: nblk * 32 * C_ // 2].view(torch.bfloat16).view(nblk, C_ // 4, 32, 4).float()


def blh_gates_unpack(buf: torch.Tensor, nblk: int) -> torch.Tensor:
    """BLH(2 * 4H) buffer of unorm16 gate codes (float32 storage) -> BL-shaped fp32 activated gates
    [nblk][2 * 4H / 4][32][4], decoded exactly like the kernels (one fp32 fma per element): i, f, o = u / 65535,
    g = u / 32767.5 - 1 (column = dir * 4H + gate * H + unit -> the g gate's quads are (q >> 6) & 3 == 2)."""
    Cq = 2 * 4 * L.LSTM_H // 4
    code = (buf.reshape(-1)[: nblk * 32 * Cq * 2].view(torch.int16).to(torch.int32) & 0xFFFF).view(nblk, Cq, 32, 4).double()
    s_sig = float(torch.tensor(1.0 / 65535.0, dtype=torch.float32))
    s_tanh = float(torch.tensor(1.0 / 32767.5, dtype=torch.float32))
    is_g = ((torch.arange(Cq, device=buf.device) >> 6) & 3 == 2).view(1, Cq, 1, 1)
    return torch.where(is_g, code * s_tanh - 1.0, code * s_sig).float()


# ---------------------------------------------------------------------------------------------
# GEMMs between the plain Z layout and BL (gemm_blk.hip)
# ---------------------------------------------------------------------------------------------
def _smc(sm: SeqMap):
    c = L.SeqMapC()
    c.sq_s1, c.sq_s2, c.step_rows, c.nseq, c.sq_div, c.L = sm.s1, sm.s2, sm.step_rows, sm.nseq, sm.div, sm.L
    c.nvalid = sm.nvalid
    return c


def pack_w(W, N: int, K: int, ldw: int, out, trans=False, order=0, w_off=0, f16=False):
    """f16: fp16 hi / lo of 256 W (ws_pack_w_f16): the weight operand of gemm_b2p with a_fmt = 2; f16 = 2: fp16 hi + e4m3 lo
    fragments (ws_pack_w_f16f8): a_fmt = 3."""
    _chk(W, "W")
    _chk(out, "out")
    if out.numel() < N * K:
        raise L.WesepHipError("pack_w: output too small")
    if int(f16) == 2:     # fp16 hi + FP8 lo fragments (ws_pack_w_f16f8, ABI v20): gemm_b2p with a_fmt = 3; b2p order only
        if order != 1:
            raise L.WesepHipError("pack_w: the fp16 + FP8 pack exists in the b2p order only")
        L.check(L.lib().ws_pack_w_f16f8(_p(W, w_off), N, K, ldw, int(trans), _p(out), L.stream_ptr()), "ws_pack_w_f16f8")
        return
    fn = L.lib().ws_pack_w_f16 if f16 else L.lib().ws_pack_w
    L.check(fn(_p(W, w_off), N, K, ldw, int(trans), order, _p(out), L.stream_ptr()), "ws_pack_w_f16" if f16 else "ws_pack_w")


def gemm_p2b(*, A, lda: int, sm: SeqMap, Wpack, N: int, C_out, K=128, bias=None, A_bl=None, stats=None,
             gamma=None, beta=None, stat_map: Optional[StatMap] = None, run_if=None, amax=None, A_bl16=None):
    """A_bl16 (ABI v16): the (normalised) operand once more as fp16 in BLH(K) (float32 storage of half the element count:
    blh_floats) -- the 2-byte A operand of gemm_tnb (a_fmt = 1)."""
    for n, t in (("A", A), ("Wpack", Wpack), ("bias", bias), ("C", C_out), ("A_bl", A_bl), ("stats", stats),
                 ("gamma", gamma), ("beta", beta), ("A_bl16", A_bl16)):
        _chk(t, n)
    a = L.GemmP2BArgs()
    a.A, a.Wpack, a.bias, a.C, a.A_bl = _p(A), _p(Wpack), _p(bias), _p(C_out), _p(A_bl)
    a.stats, a.gamma, a.beta = _p(stats), _p(gamma), _p(beta)
    a.sm = _smc(sm)
    st = stat_map or StatMap(1, 0, 1, 0, 0)
    a.st_div1, a.st_m1, a.st_div2, a.st_m2, a.st_base = st
    a.lda, a.N, a.K = lda, N, K
    a.run_if = C.c_void_p(run_if.data_ptr()) if run_if is not None else None
    a.amax = C.c_void_p(amax.data_ptr()) if amax is not None else None     # 1-element int32, zeroed by the caller
    a.A_bl16 = _p(A_bl16)
    if ALG is not None and run_if is None:
        # rows x (plain operand in, BL result out, the operand's BL copies: split pairs 4 B, fp16 2 B) + the weights
        rows = (getattr(sm, "nvalid", 0) or sm.nseq) * sm.L
        _alg("gemm_nt", rows * (4 * K + 4 * N + (4 * K if A_bl is not None else 0) + (2 * K if A_bl16 is not None else 0)) + 4 * N * K,
             2 * rows * N * K)
    L.check(L.lib().ws_gemm_p2b(C.byref(a), L.stream_ptr()), "ws_gemm_p2b")


def gemm_b2p(*, A, K: int, sm: SeqMap, Wpack, C_out, ldc: int, N=128, bias=None, R=None, a_fmt=0, amax=None, a16_out=None):
    """a_fmt = 1: A holds bf16 elements in BLH(K) (d(gates) of WS_GATES_H2) instead of split pairs in BL(K); 2: fp16 elements
    scaled by the power of two the word `amax` defines (WS_GATES_H2F).  a16_out (a_fmt 0, ABI v16): the A operand once more as
    fp16 in BLH(K) -- hcat for gemm_tnb (a_fmt = 1)."""
    for n, t in (("A", A), ("Wpack", Wpack), ("bias", bias), ("C", C_out), ("R", R), ("a16_out", a16_out)):
        _chk(t, n)
    if a16_out is not None and a_fmt != 0:
        raise L.WesepHipError("gemm_b2p: a16_out goes with a_fmt = 0 (split-pair A)")
    a = L.GemmB2PArgs()
    a.A, a.Wpack, a.bias, a.R, a.C = _p(A), _p(Wpack), _p(bias), _p(R), _p(C_out)
    a.sm = _smc(sm)
    a.ldc, a.N, a.K, a.a_fmt = ldc, N, K, a_fmt
    a.amax = C.c_void_p(amax.data_ptr()) if amax is not None else None
    a.a16_out = _p(a16_out)
    if ALG is not None:
        rows = (getattr(sm, "nvalid", 0) or sm.nseq) * sm.L
        _alg("gemm_nt", rows * ((2 if a_fmt else 4) * K + 4 * N * (1 + (R is not None)) + (2 * K if a16_out is not None else 0))
             + 4 * N * K, 2 * rows * N * K)
    L.check(L.lib().ws_gemm_b2p(C.byref(a), L.stream_ptr()), "ws_gemm_b2p")


def tnb_splits(nblk: int, gtiles: int):
    """Splits of the block range: gtiles * nsplit workgroups ~ WESEP_TNB_WGS (default: one per CU, 256), at most
    WESEP_TNB_MAXSPLIT (64) slabs."""
    cap = int(os.environ.get("WESEP_TNB_MAXSPLIT", "64"))
    nsplit = max(1, min(cap, int(os.environ.get("WESEP_TNB_WGS", "256")) // gtiles, nblk))
    if nsplit >= 8:
        nsplit -= nsplit % 8          # multiple of 8: same-split workgroups share an XCD (and its L2)
    bps = -(-nblk // nsplit)
    return nsplit, bps                # trailing splits may be empty (they write zero slabs)


def gemm_tnb(*, G, g_width: int, g_off: int, g_cols: int, A0, a0_width: int, a0_off: int, a0_cols: int,
             nblk: int, L_: int, slab, nsplit: int, blocks_per_split: int, a0_shift=0, A1=None, a1_width=0,
             a1_off=0, a1_cols=0, a1_shift=0, bslab=None, aslab=None, g_fmt=0, amax=None, a_fmt=0):
    """g_fmt = 1: G holds bf16 elements in BLH(g_width) (d(gates) of WS_GATES_H2); 2: fp16 elements scaled by the power of
    two the word `amax` defines (WS_GATES_H2F); both need 384 A columns.  a_fmt = 1 (ABI v16, g_fmt 2 only): A0 / A1 hold
    fp16 elements in BLH (gemm_p2b's A_bl16, gemm_b2p's a16_out): one MFMA per product, 32 instead of 56 KB loaded per block."""
    for n, t in (("G", G), ("A0", A0), ("A1", A1), ("slab", slab), ("bslab", bslab), ("aslab", aslab)):
        _chk(t, n)
    a = L.GemmTNBArgs()
    a.G, a.A0, a.A1, a.slab, a.bslab, a.aslab = _p(G), _p(A0), _p(A1), _p(slab), _p(bslab), _p(aslab)
    a.slab_stride = g_cols * (a0_cols + a1_cols)
    a.bslab_stride = g_cols
    a.aslab_stride = a0_cols + a1_cols
    a.g_width, a.g_off, a.g_cols = g_width, g_off, g_cols
    a.a0_width, a.a0_off, a.a0_cols, a.a0_shift = a0_width, a0_off, a0_cols, a0_shift
    a.a1_width, a.a1_off, a.a1_cols, a.a1_shift = a1_width, a1_off, a1_cols, a1_shift
    a.nblk, a.L, a.nsplit, a.blocks_per_split, a.g_fmt, a.a_fmt = nblk, L_, nsplit, blocks_per_split, g_fmt, a_fmt
    a.amax = C.c_void_p(amax.data_ptr()) if amax is not None else None
    if ALG is not None:
        rows, acols = nblk * 32, a0_cols + a1_cols
        _alg("gemm_tn", rows * ((2 if g_fmt else 4) * g_cols + (2 if a_fmt else 4) * acols) + 4 * nsplit * g_cols * acols,
             2 * rows * g_cols * acols)
    L.check(L.lib().ws_gemm_tnb(C.byref(a), L.stream_ptr()), "ws_gemm_tnb")


# ---- Conv-TasNet / SpEx+ pieces (tasnet.hip), channels-last [R*T'][C] -----------------------------
LN_EPS = 1e-5  # norm.py:18 (gLN), nn.LayerNorm default (cLN)


def _call(name, *args):
    L.check(getattr(L.lib(), name)(*args, L.stream_ptr()), name)


def flat_stats(x, ngroups: int, n_per_group: int, stats, eps=LN_EPS):
    """(mean, rstd) of contiguous groups, chunked over the chip."""
    nchunk = max(1, min(max(1, 512 // ngroups), n_per_group // 16384))
    scratch = torch.empty(ngroups, nchunk, 4, device=x.device, dtype=torch.float32)
    for n, t in (("x", x), ("stats", stats)):
        _chk(t, n)
    _call("ws_flat_stats", _p(x), ngroups, n_per_group, eps, nchunk, _p(scratch), _p(stats))


def prelu_fwd(x, rb, a, rows: int, Cc: int, rows_per_r: int, y):
    for n, t in (("x", x), ("rb", rb), ("a", a), ("y", y)):
        _chk(t, n)
    _call("ws_prelu_fwd", _p(x), _p(rb), _p(a), rows, Cc, rows_per_r, _p(y))


def prelu_bwd(pre, dy, a, dx):
    """returns d(slope) as a [1] tensor"""
    for n, t in (("pre", pre), ("dy", dy), ("a", a), ("dx", dx)):
        _chk(t, n)
    n = pre.numel()
    nslab = max(1, min(1024, n // 4096))
    slab = torch.empty(nslab, device=pre.device, dtype=torch.float32)
    _call("ws_prelu_bwd", _p(pre), _p(dy), _p(a), n, _p(dx), _p(slab), nslab)
    da = torch.empty(1, device=pre.device, dtype=torch.float32)
    reduce_slabs(slab, nslab, 1, 1, da)
    return da


def dwconv_fwd(x, stats, gamma, beta, w, b, R: int, Tp: int, Cc: int, P: int, dil: int, st_div: int, y, causal=False):
    for n, t in (("x", x), ("stats", stats), ("gamma", gamma), ("beta", beta), ("w", w), ("b", b), ("y", y)):
        _chk(t, n)
    _call("ws_dwconv_ex_fwd", _p(x), _p(stats), _p(gamma), _p(beta), _p(w), _p(b), R, Tp, Cc, P, dil, st_div, int(causal),
          _p(y))


def row_splits(M: int, target=1024):
    nsplit = max(1, min(target, M // 64))
    rows = -(-M // nsplit)
    return -(-M // rows), rows


def dwconv_bwd(dy, x, stats, gamma, beta, w, R: int, Tp: int, Cc: int, P: int, dil: int, st_div: int, dxn, causal=False):
    """returns (dw [C, P], db [C])"""
    for n, t in (("dy", dy), ("x", x), ("stats", stats), ("gamma", gamma), ("beta", beta), ("w", w), ("dxn", dxn)):
        _chk(t, n)
    nsplit, rows = row_splits(R * Tp)
    slab = torch.empty(nsplit, P + 1, Cc, device=dy.device, dtype=torch.float32)
    _call("ws_dwconv_ex_bwd", _p(dy), _p(x), _p(stats), _p(gamma), _p(beta), _p(w), R, Tp, Cc, P, dil, st_div,
          int(causal), _p(dxn), nsplit, rows, _p(slab))
    out = torch.empty(P + 1, Cc, device=dy.device, dtype=torch.float32)
    reduce_slabs(slab, nsplit, (P + 1) * Cc, (P + 1) * Cc, out)
    return out[:P].t().contiguous(), out[P].contiguous()


def chan_sums(g, x, stats, st_div: int, rows_per_group: int, ngroups: int, Cc: int):
    """[ngroups, 2, C]: per-channel sums of g and g * xhat over each group of rows_per_group rows."""
    for n, t in (("g", g), ("x", x), ("stats", stats)):
        _chk(t, n)
    nsplit = max(1, min(max(1, 1024 // ngroups), rows_per_group // 32))
    slab = torch.empty(nsplit, ngroups, 2, Cc, device=g.device, dtype=torch.float32)
    _call("ws_chan_sums", _p(g), _p(x), _p(stats), st_div, rows_per_group, ngroups, nsplit, Cc, _p(slab))
    out = torch.empty(ngroups, 2, Cc, device=g.device, dtype=torch.float32)
    reduce_slabs(slab, nsplit, ngroups * 2 * Cc, ngroups * 2 * Cc, out)
    return out


def norm_ab(sums, gamma, ngroups: int, Cc: int, n_per_group: int, ab):
    for n, t in (("sums", sums), ("gamma", gamma), ("ab", ab)):
        _chk(t, n)
    _call("ws_norm_ab", _p(sums), _p(gamma), ngroups, Cc, n_per_group, _p(ab))


def norm_bwd_apply_cl(x, dxn, stats, ab, gamma, res, rows: int, Cc: int, st_div: int, dx):
    for n, t in (("x", x), ("dxn", dxn), ("stats", stats), ("ab", ab), ("gamma", gamma), ("res", res), ("dx", dx)):
        _chk(t, n)
    _call("ws_norm_bwd_apply_cl", _p(x), _p(dxn), _p(stats), _p(ab), _p(gamma), _p(res), rows, Cc, st_div, _p(dx))


def maskmul_fwd(w, w_off: int, ldw: int, m, rows: int, N: int, s):
    for n, t in (("w", w), ("m", m), ("s", s)):
        _chk(t, n)
    _call("ws_maskmul_fwd", _p(w, w_off), ldw, _p(m), rows, N, _p(s))


def maskmul_bwd(ds, w, w_off: int, ldw: int, m, rows: int, N: int, dw, dw_off: int, ld_dw: int, dm):
    for n, t in (("ds", ds), ("w", w), ("m", m), ("dw", dw), ("dm", dm)):
        _chk(t, n)
    _call("ws_maskmul_bwd", _p(ds), _p(w, w_off), ldw, _p(m), rows, N, _p(dw, dw_off), ld_dw, _p(dm))


def relu_mask(d, y):
    for n, t in (("d", d), ("y", y)):
        _chk(t, n)
    _call("ws_relu_mask", _p(d), _p(y), d.numel())


def ola_fwd(frames, bias, R: int, Tp: int, Lk: int, hop: int, Tout: int, est):
    for n, t in (("frames", frames), ("bias", bias), ("est", est)):
        _chk(t, n)
    _call("ws_ola_fwd", _p(frames), _p(bias), R, Tp, Lk, hop, Tout, _p(est))


def ola_bwd(dest, R: int, Tp: int, Lk: int, hop: int, Tout: int, dframes):
    for n, t in (("dest", dest), ("dframes", dframes)):
        _chk(t, n)
    _call("ws_ola_bwd", _p(dest), R, Tp, Lk, hop, Tout, _p(dframes))


def total_sum(x):
    """sum of all elements as a [1] tensor (deterministic two-stage reduction)"""
    _chk(x, "x")
    n = x.numel()
    nslab = max(1, min(1024, n // 4096))
    slab = torch.empty(nslab, device=x.device, dtype=torch.float32)
    _call("ws_sum_partial", _p(x), n, _p(slab), nslab)
    out = torch.empty(1, device=x.device, dtype=torch.float32)
    reduce_slabs(slab, nslab, 1, 1, out)
    return out


# ---- SpEx+ speaker encoder pieces -------------------------------------------------------------------
BN_EPS, BN_MOMENTUM = 1e-5, 0.1  # nn.BatchNorm1d defaults


def _bn_splits(M: int):
    return max(1, min(512, M // 256))


def bn_stats(x, M: int, Cc: int, running_mean, running_var, stats, eps=BN_EPS, momentum=BN_MOMENTUM):
    if M <= 1:      # torch.nn.BatchNorm*: "Expected more than 1 value per channel when training"
        raise ValueError(f"BatchNorm in training mode needs more than 1 value per channel (got {M} row)")
    for n, t in (("x", x), ("running_mean", running_mean), ("running_var", running_var), ("stats", stats)):
        _chk(t, n)
    ns = _bn_splits(M)
    scratch = torch.empty(ns, Cc, device=x.device, dtype=torch.float32)
    _call("ws_bn_stats", _p(x), M, Cc, eps, momentum, _p(running_mean), _p(running_var), ns, _p(scratch), _p(stats))


def bn_prelu_fwd(x, stats, gamma, beta, res, a, M: int, Cc: int, u, y):
    for n, t in (("x", x), ("stats", stats), ("gamma", gamma), ("beta", beta), ("res", res), ("a", a), ("u", u),
                 ("y", y)):
        _chk(t, n)
    _call("ws_bn_prelu_fwd", _p(x), _p(stats), _p(gamma), _p(beta), _p(res), _p(a), M, Cc, _p(u), _p(y))


def bn_bwd(x, du, stats, gamma, M: int, Cc: int, dx):
    """returns sums [2, C] = (dbeta, dgamma)"""
    for n, t in (("x", x), ("du", du), ("stats", stats), ("gamma", gamma), ("dx", dx)):
        _chk(t, n)
    ns = _bn_splits(M)
    slab = torch.empty(ns, 2, Cc, device=x.device, dtype=torch.float32)
    sums = torch.empty(2, Cc, device=x.device, dtype=torch.float32)
    _call("ws_bn_bwd", _p(x), _p(du), _p(stats), _p(gamma), M, Cc, ns, _p(slab), _p(sums), _p(dx))
    return sums


def bn_bwd_any(x, du, stats, gamma, M: int, Cc: int, dx, training: bool):
    """BatchNorm backward on [M, C] rows in either mode; returns sums [2, C] = (dbeta, dgamma) like bn_bwd.
    Training mode: ws_bn_bwd (batch statistics: the mean / variance terms are part of dx).  Eval mode (running
    statistics are constants -- fine-tuning with frozen BatchNorm, torch's `module.eval()` + backward):
        dx = du * gamma * rstd,   dbeta = sum du,   dgamma = sum du * xhat
    composed from existing entry points: the two sums are what ws_bn_bwd computes in any mode (its dx goes to a
    scratch buffer), dx is the BatchNorm kernel applied to du with a zero mean, zero shift and identity activation."""
    if training:
        return bn_bwd(x, du, stats, gamma, M, Cc, dx)
    scratch = torch.empty_like(dx)
    sums = bn_bwd(x, du, stats, gamma, M, Cc, scratch)
    st0 = stats.clone()
    st0[0].zero_()
    one = torch.ones(1, device=x.device, dtype=torch.float32)
    bn_prelu_fwd(du, st0, gamma, torch.zeros_like(gamma), None, one, M, Cc, scratch, dx)
    return sums


def maxpool3_fwd(x, R: int, T: int, Cc: int, y):
    _chk(x, "x")
    _chk(y, "y")
    _call("ws_maxpool3_fwd", _p(x), R, T, Cc, _p(y))


def maxpool3_bwd(x, dy, R: int, T: int, Cc: int, dx):
    for n, t in (("x", x), ("dy", dy), ("dx", dx)):
        _chk(t, n)
    _call("ws_maxpool3_bwd", _p(x), _p(dy), R, T, Cc, _p(dx))


def bcast_rows(src, scale: float, rows_per_r: int, M: int, Cc: int, out):
    _chk(src, "src")
    _chk(out, "out")
    _call("ws_bcast_rows", _p(src), scale, rows_per_r, M, Cc, _p(out))


def cross_entropy(logits, label, loss, dlogits):
    _chk(logits, "logits")
    _chk(label, "label", torch.int64)
    _chk(loss, "loss")
    _chk(dlogits, "dlogits")
    R, S = logits.shape
    _call("ws_cross_entropy", _p(logits), C.c_void_p(label.data_ptr()), R, S, _p(loss), _p(dlogits))


# ---- forward recurrence with the input projection fused in (lstm_fused.hip) --------------------------
def lstm_fuse_ok(nseq: int, cluster: bool) -> bool:
    """Fused x-projection + recurrence for views with enough sequences to fill the chip with 32-sequence
    workgroups (pBSRNN's band view); WESEP_LSTM_FUSE=0 keeps the two-kernel path."""
    if os.environ.get("WESEP_LSTM_FUSE", "1") == "0" or cluster:
        return False
    return lstm_blk_mode(nseq) == L.LSTM_BF16X3_BLK


def lstm_fused_hfmt(gfmt) -> int:
    """Arithmetic of the fused band-view forward's recurrent part (ws_lstm_fused_args.hfmt, ABI v19): 1 (default with the 2-byte
    gate formats) = h as ONE fp16 operand against W_hh as fp16 hi / lo of 256 w on v_mfma_f32_32x32x16_f16, two MFMAs per product
    -- what ws_lstm_fwd_cluster2 runs in the time view since round 5; the x part keeps the three-term split product.
    WESEP_FUSED_H16=0: the three-term product of rounds 1-5 for both parts.
    Bit 2 (ABI v20, default on; with bit 0: 5): the lo term of that product on the block-scaled FP8 matrix instruction (one
    K = 64 MFMA at twice the fp16 rate for four K = 16 ones; 2.12 -> 1.92 ms per launch alone, 1.3 ms per step:
    profiles/r06_c21_band_probe.txt, r06_ab/r06_c22_*) -- both the 64- and the 32-sequence kernel.
    WESEP_FUSED_F8=0: both terms on the fp16 MFMA."""
    if gfmt == L.GATES_F32 or os.environ.get("WESEP_FUSED_H16", "1") == "0":
        return 0
    return 5 if os.environ.get("WESEP_FUSED_F8", "1") != "0" else 1


def lstm_pack_fused(wih_f, wih_r, whh_f, whh_r, pack, hfmt=0):
    for n, t in (("wih_f", wih_f), ("wih_r", wih_r), ("whh_f", whh_f), ("whh_r", whh_r), ("pack", pack)):
        _chk(t, n)
    _call("ws_lstm_pack_fused_h8" if hfmt & 4 else "ws_lstm_pack_fused_h16" if hfmt & 1 else "ws_lstm_pack_fused",
          _p(wih_f), _p(wih_r), _p(whh_f), _p(whh_r), _p(pack))


def lstm_fwd_fused(gates, cbuf, hcat, xn, wpack, bias, sm: SeqMap, gfmt=0, hfmt=0):
    for n, t in (("gates", gates), ("cbuf", cbuf), ("hcat", hcat), ("xn", xn), ("wpack", wpack), ("bias", bias)):
        _chk(t, n)
    a = L.LstmFusedArgs()
    a.gates, a.cbuf, a.hcat, a.xn, a.wpack, a.bias = _p(gates), _p(cbuf), _p(hcat), _p(xn), _p(wpack), _p(bias)
    a.nseq, a.L, a.gfmt = sm.nseq, sm.L, gfmt
    a.hfmt = hfmt | (2 if os.environ.get("WESEP_FUSED_DRAIN", "0") == "1" else 0)      # (bit 1: A/B of the end-of-step wait)
    # per (position, direction): 4H gates (2 B / 4 B) + c + h out, the 128-wide split-pair input in (shared by both directions)
    u = ALG_LSTM_UNITS or L.LSTM_H
    _alg("lstm_fwd", sm.nseq * sm.L * (2 * u * ((2 if gfmt else 4) * 4 + 8) + 4 * 128), 2 * sm.nseq * sm.L * 2 * 4 * u * (u + 128))
    L.check(L.lib().ws_lstm_fwd_fused(C.byref(a), L.stream_ptr()), "ws_lstm_fwd_fused")


# ---- 2-D convolution pieces (conv2d.hip) -------------------------------------------------------------
def conv_out(n: int, k: int, s: int, p: int) -> int:
    return (n + 2 * p - k) // s + 1


def im2col(x, R: int, H: int, W: int, Cc: int, k: int, s: int, p: int, patches, ldp: int):
    _chk(x, "x")
    _chk(patches, "patches")
    _call("ws_im2col", _p(x), R, H, W, Cc, k, s, p, ldp, _p(patches))


def col2im(dpatches, R: int, H: int, W: int, Cc: int, k: int, s: int, p: int, dx):
    _chk(dpatches, "dpatches")
    _chk(dx, "dx")
    _call("ws_col2im", _p(dpatches), R, H, W, Cc, k, s, p, _p(dx))


def tstp_fwd(x, R: int, F: int, T: int, Cc: int, stats, eps=1e-7):
    _chk(x, "x")
    _chk(stats, "stats")
    _call("ws_tstp_fwd", _p(x), R, F, T, Cc, eps, _p(stats))


def tstp_bwd(x, stats, dstats, R: int, F: int, T: int, Cc: int, dx):
    for n, t in (("x", x), ("stats", stats), ("dstats", dstats), ("dx", dx)):
        _chk(t, n)
    _call("ws_tstp_bwd", _p(x), _p(stats), _p(dstats), R, F, T, Cc, _p(dx))


ASTP_FLOOR = 1e-7


def astp_fwd(x, logits, R: int, T: int, Cc: int, out, aux):
    for n, t in (("x", x), ("logits", logits), ("out", out), ("aux", aux)):
        _chk(t, n)
    _call("ws_astp_fwd", _p(x), _p(logits), R, T, Cc, ASTP_FLOOR, _p(out), _p(aux))


def astp_bwd(x, logits, out, aux, dout, R: int, T: int, Cc: int, dx, dlogits):
    for n, t in (("x", x), ("logits", logits), ("out", out), ("aux", aux), ("dout", dout), ("dx", dx),
                 ("dlogits", dlogits)):
        _chk(t, n)
    _call("ws_astp_bwd", _p(x), _p(logits), _p(out), _p(aux), _p(dout), R, T, Cc, ASTP_FLOOR, _p(dx), _p(dlogits))


def rowbias_act_fwd(x, rb, rows: int, Cc: int, rows_per_r: int, act: int, y):
    for n, t in (("x", x), ("rb", rb), ("y", y)):
        _chk(t, n)
    _call("ws_rowbias_act_fwd", _p(x), _p(rb), rows, Cc, rows_per_r, act, _p(y))


def act_bwd(y, dy, act: int, dx):
    for n, t in (("y", y), ("dy", dy), ("dx", dx)):
        _chk(t, n)
    _call("ws_act_bwd", _p(y), _p(dy), y.numel(), act, _p(dx))


def seg_sums(a, b, R: int, T: int, Cc: int, seg_len: int, out):
    """out [R, ceil(T / seg_len), C] = per-segment sums of a (* b) over the frames of channels-last [R*T, C]."""
    for n, t in (("a", a), ("b", b), ("out", out)):
        _chk(t, n)
    _call("ws_seg_sums", _p(a), _p(b), R, T, Cc, seg_len, _p(out))


def seg_scale(x, m, R: int, T: int, Cc: int, seg_len: int, out):
    """out[r, t] = (x[r, t] if x is not None else 1) * m[r, t // seg_len]."""
    for n, t in (("x", x), ("m", m), ("out", out)):
        _chk(t, n)
    _call("ws_seg_scale", _p(x), _p(m), R, T, Cc, seg_len, _p(out))


# ---- in-model enrollment front-end (conv2d.hip) ---------------------------------------------------------
def preemph_pad(x, R: int, T: int, pad: int, ldo: int, coef: float, out):
    _chk(x, "x")
    _chk(out, "out")
    _call("ws_preemph_pad", _p(x), R, T, pad, ldo, coef, _p(out))


def power_spec(spec, M: int, nf: int, lds: int, ldp: int, p):
    _chk(spec, "spec")
    _chk(p, "p")
    _call("ws_power_spec", _p(spec), M, nf, lds, ldp, _p(p))


def log_eps(x, eps: float):
    _chk(x, "x")
    _call("ws_log_eps", _p(x), x.numel(), eps)


# ---- DPCCN pieces (conv2d.hip) --------------------------------------------------------------------------
IN_EPS = 1e-5  # nn.InstanceNorm{1,2}d default


def im2col_hw(x, R: int, H: int, W: int, Cc: int, k: int, sh: int, sw: int, p: int, patches, ldp: int):
    _chk(x, "x")
    _chk(patches, "patches")
    _call("ws_im2col_hw", _p(x), R, H, W, Cc, k, sh, sw, p, ldp, _p(patches))


def col2im_hw(dpatches, R: int, H: int, W: int, Cc: int, k: int, sh: int, sw: int, p: int, dx):
    _chk(dpatches, "dpatches")
    _chk(dx, "dx")
    _call("ws_col2im_hw", _p(dpatches), R, H, W, Cc, k, sh, sw, p, _p(dx))


def elu_fwd(x, y):
    _chk(x, "x")
    _chk(y, "y")
    _call("ws_elu_fwd", _p(x), x.numel(), _p(y))


def elu_bwd(x, dy, dx):
    for n, t in (("x", x), ("dy", dy), ("dx", dx)):
        _chk(t, n)
    _call("ws_elu_bwd", _p(x), _p(dy), x.numel(), _p(dx))


def inorm_fwd(x, G: int, P: int, Cc: int, y, eps=IN_EPS):
    """y = InstanceNorm(x) over the P positions of each of G rows; returns stats [G, 2, C] = (mean, rstd)."""
    _chk(x, "x")
    _chk(y, "y")
    sums = chan_sums(x, x, None, 1, P, G, Cc)
    stats = torch.empty(G, 2, Cc, device=x.device, dtype=torch.float32)
    _call("ws_inorm_finalize", _p(sums), G, Cc, P, eps, _p(stats))
    _call("ws_inorm_apply", _p(x), _p(stats), G * P, P, Cc, _p(y))
    return stats


def inorm_bwd(y, dy, stats, G: int, P: int, Cc: int, dx):
    for n, t in (("y", y), ("dy", dy), ("stats", stats), ("dx", dx)):
        _chk(t, n)
    sums = chan_sums(dy, y, None, 1, P, G, Cc)
    _call("ws_inorm_bwd_apply", _p(y), _p(dy), _p(stats), _p(sums), G * P, P, Cc, _p(dx))


def conv3x3_pack(W2, Cin: int, Cout: int):
    """W2 [Cout, 9 * Cin] (tap-major rows: (ky*3 + kx)*Cin + ci) -> the bf16 hi / lo MFMA-fragment order ws_conv3x3 reads
    (include/wesep_hip.h), as a float32-typed tensor of packed pairs.  A handful of torch ops on a tiny tensor."""
    ntt, nch = -(-Cout // 32), -(-Cin // 16)
    ntp = ntt if ntt <= 2 else ntt + (ntt & 1)
    Wp = torch.zeros(ntp * 32, 9, nch * 16, device=W2.device, dtype=torch.float32)
    Wp[:Cout, :, :Cin] = W2.reshape(Cout, 9, Cin)
    hi = Wp.to(torch.bfloat16)
    lo = (Wp - hi.to(torch.float32)).to(torch.bfloat16)
    both = torch.stack([hi, lo], 0).view(2, ntp, 32, 9, nch, 2, 8)             # part, t, l31, tap, chunk, half, j
    return both.permute(4, 3, 1, 0, 5, 2, 6).contiguous().view(torch.float32).reshape(-1)


def conv3x3_pack_floats(Cin: int, Cout: int) -> int:
    ntt, nch = -(-Cout // 32), -(-Cin // 16)
    ntp = ntt if ntt <= 2 else ntt + (ntt & 1)
    return nch * 9 * ntp * 2 * 64 * 8 // 2


def conv3x3_pack_srcs(srcs, Cin: int, Cout: int, flip: bool = False):
    """The packed weights of conv3x3 in ONE launch (ws_conv3x3_pack, ABI v19).  srcs: up to five (w, elem_off, s_row, s_col,
    s_tap, col_off, cols) -- column c of the logical W[n][tap][c] comes from the source whose [col_off, col_off + cols) holds it:
    w.flat[elem_off + n * s_row + (c - col_off) * s_col + (8 - tap if flip else tap) * s_tap].  Replaces conv3x3_pack's ATen
    composition (zeros, slice copy, casts, stack, permute: ~9 launches per pack)."""
    if not 0 < len(srcs) <= 5:
        raise L.WesepHipError("conv3x3_pack_srcs: 1 .. 5 sources")
    out = torch.empty(conv3x3_pack_floats(Cin, Cout), device=srcs[0][0].device, dtype=torch.float32)
    a = L.Conv3x3PackArgs()
    for k, (w, off, s_row, s_col, s_tap, col_off, cols) in enumerate(srcs):
        _chk(w, "conv3x3_pack_srcs w")
        a.src[k].w = _p(w, off)
        a.src[k].s_row, a.src[k].s_col, a.src[k].s_tap, a.src[k].col_off, a.src[k].cols = s_row, s_col, s_tap, col_off, cols
    a.out, a.Cin, a.Cout, a.nsrc, a.flip = _p(out), Cin, Cout, len(srcs), int(flip)
    L.check(L.lib().ws_conv3x3_pack(C.byref(a), L.stream_ptr()), "ws_conv3x3_pack")
    return out


def conv3x3(*, X, ldx: int, W, ldw: int, B: int, H: int, Wd: int, Cin: int, Cout: int, Y, ldy: int, bias=None, R=None,
            x_off: int = 0, y_off: int = 0):
    """3 x 3 / stride 1 / padding 1 convolution through an LDS halo tile (conv3x3.hip); Y = bias + R + conv(X); W = the
    packed weights of conv3x3_pack.  x_off / y_off: the image is columns [x_off, x_off + Cin) of rows of stride ldx, the
    output (and R) columns [y_off, y_off + Cout) of rows of stride ldy."""
    for n, t in (("X", X), ("W", W), ("bias", bias), ("R", R), ("Y", Y)):
        _chk(t, n)
    _cols_ok(X, B * H * Wd, ldx, x_off, Cin, "conv3x3 X")
    _cols_ok(Y, B * H * Wd, ldy, y_off, Cout, "conv3x3 Y")
    a = L.Conv3x3Args()
    a.X, a.W, a.bias, a.R, a.Y = _p(X, x_off), _p(W), _p(bias), _p(R, y_off), _p(Y, y_off)
    a.ldx, a.ldw, a.ldy = ldx, ldw, ldy
    a.B, a.H, a.Wd, a.Cin, a.Cout = B, H, Wd, Cin, Cout
    _alg("gemm_nt", 4 * (B * H * Wd * (Cin + Cout * (1 + (R is not None))) + 9 * Cin * Cout), 2 * B * H * Wd * 9 * Cin * Cout)
    L.check(L.lib().ws_conv3x3(C.byref(a), L.stream_ptr()), "ws_conv3x3")


def conv3x3_wgrad_tiles(B: int, H: int, Wd: int) -> int:
    """Tiles (30 rows x 4 columns) ws_conv3x3_wgrad cuts B images of H x Wd into."""
    return B * (-(-H // 30)) * (-(-Wd // 4))


def conv3x3_wgrad(*, G, ldg: int, X, ldx: int, B: int, H: int, Wd: int, Cin: int, Nn: int, slab, nsplit: int,
                  tiles_per_split: int, bslab=None, sw: int = 1, Wx: int = 0, g_off: int = 0):
    """Weight (+ bias) gradient slabs of a 3 x 3 / padding 1 convolution with stride (1, sw), one pass over the image
    (conv3x3.hip): slab [nsplit, Nn * 9 * Cin], bslab [nsplit, Nn].  Wd = width of the gradient grid, Wx = of the image."""
    for n, t in (("G", G), ("X", X), ("slab", slab), ("bslab", bslab)):
        _chk(t, n)
    _cols_ok(G, B * H * Wd, ldg, g_off, Nn, "conv3x3_wgrad G")
    a = L.Conv3x3WgradArgs()
    a.G, a.X, a.slab, a.bslab = _p(G, g_off), _p(X), _p(slab), _p(bslab)
    a.ldg, a.ldx, a.slab_stride, a.bslab_stride = ldg, ldx, Nn * 9 * Cin, Nn
    a.B, a.H, a.Wd, a.Wx, a.sw, a.Cin, a.Nn, a.nsplit, a.tiles_per_split = B, H, Wd, Wx or Wd, sw, Cin, Nn, nsplit, tiles_per_split
    _alg("gemm_tn", 4 * (B * H * (Wx or Wd) * Cin + B * H * Wd * Nn * (-(-Cin // 32)) + nsplit * Nn * 9 * Cin),
         2 * B * H * Wd * Nn * 9 * Cin)
    L.check(L.lib().ws_conv3x3_wgrad(C.byref(a), L.stream_ptr()), "ws_conv3x3_wgrad")


IN_ELU_PRE, IN_ELU_POST = 1, 2     # ws_in_act_* flags: y = IN(ELU(x)) / y = ELU(IN(x))


def _in_act_sums(x, dy, stats, G: int, P: int, Cc: int, flags: int, dy_ld: int = 0, dy_off: int = 0):
    nsplit = max(1, min(max(1, 1024 // G), P // 32))
    slab = torch.empty(nsplit, G, 2, Cc, device=x.device, dtype=torch.float32)
    _call("ws_in_act_sums", _p(x), _p(dy, dy_off), dy_ld, _p(stats), P, G, nsplit, Cc, flags, _p(slab))
    out = torch.empty(G, 2, Cc, device=x.device, dtype=torch.float32)
    reduce_slabs(slab, nsplit, G * 2 * Cc, G * 2 * Cc, out)
    return out


def _cols_ok(t, rows: int, ld: int, off: int, Cc: int, name: str):
    """Columns [off, off + Cc) of the dense [rows, ld] tensor t (ld = 0: t is [rows, Cc] itself)."""
    if ld and (ld % 4 or off % 4 or off < 0 or off + Cc > ld or t.numel() < rows * ld):
        raise L.WesepHipError(f"{name}: columns [{off}, {off + Cc}) of rows of stride {ld} (both % 4) do not fit the tensor")
    if not ld and (off or t.numel() < rows * Cc):
        raise L.WesepHipError(f"{name}: expected {rows} x {Cc} elements")


def in_act_fwd(x, G: int, P: int, Cc: int, flags: int, y, eps=IN_EPS, y_ld: int = 0, y_off: int = 0):
    """y = IN(ELU(x)) (flags IN_ELU_PRE) or ELU(IN(x)) (IN_ELU_POST) over the P positions of each of G rows, three
    passes; returns the statistics [G, 2, C] the backward needs (with x).  y_ld / y_off: write columns [y_off, y_off + Cc)
    of the dense [G*P, y_ld] tensor y instead of a dense [G*P, Cc] one."""
    _chk(x, "x")
    _chk(y, "y")
    _cols_ok(y, G * P, y_ld, y_off, Cc, "in_act_fwd y")
    sums = _in_act_sums(x, None, None, G, P, Cc, flags)
    stats = torch.empty(G, 2, Cc, device=x.device, dtype=torch.float32)
    _call("ws_inorm_finalize", _p(sums), G, Cc, P, eps, _p(stats))
    _call("ws_in_act_apply", _p(x), _p(stats), G * P, P, Cc, flags, _p(y, y_off), y_ld)
    return stats


def in_act_bwd(x, dy, stats, G: int, P: int, Cc: int, flags: int, dx, dy_ld: int = 0, dy_off: int = 0, dx_ld: int = 0,
               dx_off: int = 0):
    """dy_ld / dy_off: dy is columns [dy_off, dy_off + Cc) of a dense [G*P, dy_ld] tensor; dx_ld / dx_off likewise for dx."""
    for n, t in (("x", x), ("dy", dy), ("stats", stats), ("dx", dx)):
        _chk(t, n)
    _cols_ok(dy, G * P, dy_ld, dy_off, Cc, "in_act_bwd dy")
    _cols_ok(dx, G * P, dx_ld, dx_off, Cc, "in_act_bwd dx")
    sums = _in_act_sums(x, dy, stats, G, P, Cc, flags, dy_ld, dy_off)
    _call("ws_in_act_bwd_apply", _p(x), _p(dy, dy_off), dy_ld, _p(stats), _p(sums), G * P, P, Cc, flags, _p(dx, dx_off), dx_ld)


def avgpool_fwd(x, B: int, H: int, W: int, Cc: int, sz: int, y):
    _chk(x, "x")
    _chk(y, "y")
    _call("ws_avgpool_fwd", _p(x), B, H, W, Cc, sz, _p(y))


def avgpool_bwd(dy, B: int, H: int, W: int, Cc: int, sz: int, dx):
    _chk(dy, "dy")
    _chk(dx, "dx")
    _call("ws_avgpool_bwd", _p(dy), B, H, W, Cc, sz, _p(dx))


def bilinear_fwd(x, B: int, h: int, w: int, H: int, W: int, Cc: int, y):
    _chk(x, "x")
    _chk(y, "y")
    _call("ws_bilinear_fwd", _p(x), B, h, w, H, W, Cc, _p(y))


def bilinear_bwd(dy, B: int, h: int, w: int, H: int, W: int, Cc: int, dx):
    _chk(dy, "dy")
    _chk(dx, "dx")
    tmp = torch.empty(B * H * w * Cc, device=dy.device, dtype=torch.float32)      # row pass of the separable adjoint
    _call("ws_bilinear_bwd", _p(dy), B, h, w, H, W, Cc, _p(tmp), _p(dx))


def scale_bf_fwd(x, s, B: int, T: int, Fq: int, Cc: int, mode: int, y):
    for n, t in (("x", x), ("s", s), ("y", y)):
        _chk(t, n)
    _call("ws_scale_bf_fwd", _p(x), _p(s), B, T, Fq, Cc, mode, _p(y))


def freq_linear_fwd(x, W, ldw: int, rb, B: int, T: int, Fq: int, Cc: int, y):
    """ws_freq_linear_fwd: y[b, t, f', c] = sum_f W[f', f] x[b, t, f, c] + rb[b, f'] (the 'concat' speaker fusion's Linear over
    the frequency axis; the native runtime's forward form)."""
    for n, t in (("x", x), ("W", W), ("rb", rb), ("y", y)):
        _chk(t, n)
    _call("ws_freq_linear_fwd", _p(x), _p(W), ldw, _p(rb), B, T, Fq, Cc, _p(y))


def scale_bf_bwd(x, dy, s, B: int, T: int, Fq: int, Cc: int, mode: int, dx, ds):
    for n, t in (("x", x), ("dy", dy), ("s", s), ("dx", dx), ("ds", ds)):
        _chk(t, n)
    _call("ws_scale_bf_bwd", _p(x), _p(dy), _p(s), B, T, Fq, Cc, mode, _p(dx), _p(ds))


def softmax_rows_fwd(x, rows: int, n: int, scale: float, y):
    _chk(x, "x")
    _chk(y, "y")
    _call("ws_softmax_rows_fwd", _p(x), rows, n, scale, _p(y))


def rowln_ok(W: int) -> bool:
    """Widths the one-pass row LayerNorm kernels take (norm.hip); wider rows use group_stats + dwconv_fwd."""
    return 0 < W <= 256 and W % 4 == 0


def rowln_fwd(x, gamma, beta, M: int, W: int, y, stats, eps=LN_EPS):
    for nm, t in (("x", x), ("gamma", gamma), ("beta", beta), ("y", y), ("stats", stats)):
        _chk(t, nm)
    _call("ws_rowln_fwd", _p(x), _p(gamma), _p(beta), M, W, eps, _p(y), _p(stats))


def rowln_bwd(x, dy, stats, gamma, M: int, W: int, dx, res=None):
    """dx (may alias dy) and the [2, W] sums (d(beta), d(gamma)), reduced from the kernel's per-workgroup slabs."""
    for nm, t in (("x", x), ("dy", dy), ("stats", stats), ("gamma", gamma), ("dx", dx), ("res", res)):
        _chk(t, nm)
    n = L.lib().ws_rowln_grid(M, W)
    slab = torch.empty(n, 2 * W, device=x.device, dtype=torch.float32)
    _call("ws_rowln_bwd", _p(x), _p(dy), _p(stats), _p(gamma), _p(res), M, W, _p(dx), _p(slab))
    tot = torch.empty(2, W, device=x.device, dtype=torch.float32)
    reduce_slabs(slab, n, 2 * W, 2 * W, tot)
    return tot


HEADS_SLAB_PAD = 8


def heads_ok(Q: int, nh: int, ch: int) -> bool:
    """The one-pass attention-head kernels (heads.hip) take this shape."""
    return ch % 4 == 0 and 0 < nh <= 8 and Q * nh * ch <= 9216


def _heads_args(x, x_ld, x_off, slope, gamma, beta, B, T, Tp, Q, nh, ch, stats):
    a = L.HeadsArgs()
    a.x, a.slope, a.gamma, a.beta, a.stats = _p(x, x_off), _p(slope), _p(gamma), _p(beta), _p(stats)
    a.ldx, a.B, a.T, a.Tp, a.Q, a.nh, a.ch, a.eps = x_ld, B, T, Tp, Q, nh, ch, LN_EPS
    return a


def heads_fwd(x, x_ld: int, x_off: int, slope, gamma, beta, B: int, T: int, Tp: int, Q: int, nh: int, ch: int, y, stats):
    """y [nh*B, Tp, Q*ch] (zero rows t >= T) and stats [nh, B*T, 2] from columns [x_off, x_off + nh*ch) of the
    projection output x [B*T*Q, x_ld]: PReLU(slope[h]) + LayerNorm over (Q, ch) with gamma / beta [nh, Q*ch]."""
    for nm, t in (("x", x), ("slope", slope), ("gamma", gamma), ("beta", beta), ("y", y), ("stats", stats)):
        _chk(t, nm)
    _cols_ok(x, B * T * Q, x_ld, x_off, nh * ch, "heads_fwd x")
    a = _heads_args(x, x_ld, x_off, slope, gamma, beta, B, T, Tp, Q, nh, ch, stats)
    a.y = _p(y)
    L.check(L.lib().ws_heads_fwd(C.byref(a), L.stream_ptr()), "ws_heads_fwd")


def heads_bwd(x, x_ld: int, x_off: int, dy, slope, gamma, stats, B: int, T: int, Tp: int, Q: int, nh: int, ch: int,
              dx, dx_ld: int, dx_off: int):
    """dx into columns [dx_off, dx_off + nh*ch) of [B*T*Q, dx_ld]; returns (dgamma [nh, Q*ch], dbeta [nh, Q*ch], dslope [nh])."""
    for nm, t in (("x", x), ("dy", dy), ("slope", slope), ("gamma", gamma), ("stats", stats), ("dx", dx)):
        _chk(t, nm)
    _cols_ok(x, B * T * Q, x_ld, x_off, nh * ch, "heads_bwd x")
    _cols_ok(dx, B * T * Q, dx_ld, dx_off, nh * ch, "heads_bwd dx")
    W = Q * nh * ch
    nwg = min(B * T, 512)
    stride = 2 * W + HEADS_SLAB_PAD
    slab = torch.empty(nwg, stride, device=x.device, dtype=torch.float32)
    a = _heads_args(x, x_ld, x_off, slope, gamma, None, B, T, Tp, Q, nh, ch, stats)
    a.dy, a.dx, a.lddx, a.slab, a.nwg = _p(dy), _p(dx, dx_off), dx_ld, _p(slab), nwg
    L.check(L.lib().ws_heads_bwd(C.byref(a), L.stream_ptr()), "ws_heads_bwd")
    tot = torch.empty(stride, device=x.device, dtype=torch.float32)
    reduce_slabs(slab, nwg, stride, stride, tot)
    dg = tot[:W].view(Q, nh, ch).permute(1, 0, 2).reshape(nh, Q * ch).contiguous()
    db = tot[W:2 * W].view(Q, nh, ch).permute(1, 0, 2).reshape(nh, Q * ch).contiguous()
    return dg, db, tot[2 * W:2 * W + nh].contiguous()


def softmax_rows_bwd(y, dy, rows: int, n: int, scale: float, dx):
    for nm, t in (("y", y), ("dy", dy), ("dx", dx)):
        _chk(t, nm)
    _call("ws_softmax_rows_bwd", _p(y), _p(dy), rows, n, scale, _p(dx))
