"""Autograd shims of the TF-GridNet path over the C ABI.

Channels-last grids [B*T*Q, C] (T frames, Q frequency bins).  Everything is composed from entry points that other rows
already use, plus one row-softmax kernel:
  * the BLSTMs (hidden 192, input emb_dim * emb_ks) run on the split-bf16 recurrence kernels built for hidden 256:
    the weights are ZERO-PADDED to 256 units -- a padded unit has pre-activations 0, so i = f = o = 1/2, g = 0, c = h = 0
    for ever, and its outgoing weights are zero: the padded network computes exactly the same function;
  * `F.unfold(..., (emb_ks, 1), stride (emb_hs, 1))` in front of a BLSTM is an overlapping ROW VIEW of the layer-normed
    sequence (window of emb_ks consecutive positions = emb_ks * C contiguous floats) fed to the input-projection GEMM; its
    adjoint and the ConvTranspose1d behind the BLSTM are GEMM + the generic overlap-add kernel (hop = emb_hs * C floats);
  * LayerNorm over C, LayerNormalization4DCF over (C, F) and AllHeadPReLULayerNormalization4DCF over (E, F) are all
    "one mean/variance per row, affine per column" on a suitably flattened [rows, width] view -> the cLN kernels;
  * the full-band self-attention is two GEMMs per (batch row, head) around the row softmax.
Reference lines: wesep/models/tfgridnet.py:197-302, wesep/modules/tfgridnet/gridnet_block.py:118-284."""
import numpy as np
import os

import torch

from . import _lib as L
from . import dev
from .dev import BIG, Rows, SeqMap, flat
from .functional import _empty, _need_cuda, _reduce_new
from .functional_tasnet import _cln_geom, _gemm, _transposed, _wgrad, norm_backward

HP = L.LSTM_H          # 256: hidden size the recurrence kernels are built for
G4P = 4 * HP


def pad_lstm(w_ih, w_hh, b_ih, b_hh, col_perm=None):
    """nn.LSTM tensors of hidden size h <= 256 -> (w_ih [1024, K], w_hh [1024, 256], b [1024]) zero-padded to 256 units
    (gate-major rows g*256 + u).  torch index ops on small weight tensors: gradients flow back through them."""
    h = w_hh.shape[1]
    if h > HP:
        raise NotImplementedError(f"TF-GridNet lstm_hidden_units={h} > {HP}")
    d = w_ih.device
    rows = (torch.arange(4, device=d).unsqueeze(1) * HP + torch.arange(h, device=d).unsqueeze(0)).reshape(-1)
    wi = w_ih if col_perm is None else w_ih[:, col_perm]
    wi_p = torch.zeros(G4P, wi.shape[1], device=d, dtype=torch.float32).index_copy(0, rows, wi)
    wh_c = torch.zeros(4 * h, HP, device=d, dtype=torch.float32)
    wh_c = torch.cat([w_hh, wh_c[:, h:]], 1)
    wh_p = torch.zeros(G4P, HP, device=d, dtype=torch.float32).index_copy(0, rows, wh_c)
    b_p = torch.zeros(G4P, device=d, dtype=torch.float32).index_copy(0, rows, b_ih + b_hh)
    return wi_p, wh_p, b_p


def pad_hidden_cols(w, h):
    """[N, 2h] (forward | reverse hidden) -> [N, 512] with each half zero-padded to 256 columns."""
    N = w.shape[0]
    z = torch.zeros(N, HP - h, device=w.device, dtype=torch.float32)
    return torch.cat([w[:, :h], z, w[:, h:], z], 1)


class BlstmFn(torch.autograd.Function):
    """src [nseq*Lr, C] (layer-normed) -> hcat [nseq*n, 512]: windows of ks positions (hop hs) -> input projection ->
    bidirectional recurrence.  Weights are the padded tensors of pad_lstm (both directions stacked)."""

    @staticmethod
    def forward(ctx, src, geo, wcat, bcat, whf, whr):
        _need_cuda(src, "TF-GridNet")
        nseq, Lr, C, ks, hs = geo
        n = (Lr - ks) // hs + 1
        P, K = nseq * n, ks * C
        src = src.contiguous()
        d = src.device
        rows = Rows(n, Lr * C, hs * C)
        gates = _empty(d, P, 2 * G4P)
        dev.gemm_nt(A=src, a_rows=rows, M=P, N=2 * G4P, K=K, W=wcat, ldw=K, bias=bcat, C_out=gates, c_rows=flat(2 * G4P),
                    vec=3 if (C % 4 == 0) else 0)
        pack_f, pack_b = _empty(d, L.LSTM_PACK_FLOATS), _empty(d, L.LSTM_PACK_FLOATS)
        mode = dev.lstm_mode(nseq)
        dev.lstm_pack(whf.contiguous(), whr.contiguous(), pack_f, pack_b, mode)
        cbuf, hcat = _empty(d, P, 2 * HP), _empty(d, P, 2 * HP)
        seq = SeqMap(nseq, BIG, 0, n, 1, n)
        dev.lstm_fwd(gates, cbuf, hcat, pack_f, seq, mode)
        ctx.save_for_backward(src, gates, cbuf, hcat, wcat, pack_b)
        ctx.geo = (nseq, Lr, C, ks, hs, n, mode)
        return hcat

    @staticmethod
    def backward(ctx, dhcat):
        src, gates, cbuf, hcat, wcat, pack_b = ctx.saved_tensors
        nseq, Lr, C, ks, hs, n, mode = ctx.geo
        P, K = nseq * n, ks * C
        d = src.device
        seq = SeqMap(nseq, BIG, 0, n, 1, n)
        gates = gates.clone()                                    # BPTT works in place; keep the saved tensor intact
        dev.lstm_bwd(gates, cbuf, hcat, dhcat.contiguous(), pack_b, seq, mode)
        nsplit, rps = dev.tn_splits(P)
        dwhh = []
        slab = _empty(d, nsplit, G4P * HP)
        for di in (0, 1):
            dev.gemm_tn(G=gates, g_rows=flat(2 * G4P), g_off=di * G4P, A=hcat, a_rows=flat(2 * HP), a_off=di * HP, M=P,
                        Nn=G4P, Kk=HP, slab=slab, slab_stride=G4P * HP, nsplit=nsplit, rows_per_split=rps,
                        shift_rows=(-1 if di == 0 else 1), seq_div=1, seq_len=n)
            dwhh.append(_reduce_new(slab, nsplit, G4P * HP, (G4P, HP)))
        rows = Rows(n, Lr * C, hs * C)
        dwcat, dbcat = _wgrad(gates, P, 2 * G4P, src, K, a_rows=rows, vec=1 if C % 4 == 0 else 0)
        dxu = _gemm(gates, P, 2 * G4P, _transposed(wcat, 2 * G4P, K), K)
        dsrc = _empty(d, nseq, Lr * C)
        dev.ola_fwd(dxu, None, nseq, n, K, hs * C, Lr * C, dsrc)          # adjoint of the window view: overlap-add
        return dsrc.view(nseq * Lr, C), None, dwcat, dbcat, dwhh[0], dwhh[1]


class Deconv1dFn(torch.autograd.Function):
    """hcat [nseq*n, 512] -> [nseq*Lr, C]: ConvTranspose1d(2h -> C, ks, stride hs) along the sequence
    (gridnet_block.py:47-51): GEMM to frames (column i*C + c) + overlap-add with hop hs*C floats."""

    @staticmethod
    def forward(ctx, hcat, geo, Wt):
        nseq, Lr, C, ks, hs, n = geo
        hcat = hcat.contiguous()
        fr = _gemm(hcat, nseq * n, 2 * HP, Wt, ks * C)
        y = _empty(hcat.device, nseq, Lr * C)
        dev.ola_fwd(fr, None, nseq, n, ks * C, hs * C, Lr * C, y)
        ctx.save_for_backward(hcat, Wt)
        ctx.geo = geo
        return y.view(nseq * Lr, C)

    @staticmethod
    def backward(ctx, dy):
        hcat, Wt = ctx.saved_tensors
        nseq, Lr, C, ks, hs, n = ctx.geo
        P = nseq * n
        dfr = _empty(dy.device, P, ks * C)
        dev.ola_bwd(dy.contiguous().view(nseq, Lr * C), nseq, n, ks * C, hs * C, Lr * C, dfr)
        dWt, _ = _wgrad(dfr, P, ks * C, hcat, 2 * HP, with_bias=False)
        dh = _gemm(dfr, P, ks * C, _transposed(Wt, ks * C, 2 * HP), 2 * HP)
        return dh, None, dWt


class AddRowVecFn(torch.autograd.Function):
    """x [M, C] + b [C]."""

    @staticmethod
    def forward(ctx, x, b):
        x = x.contiguous()
        y = torch.empty_like(x)
        M, C = x.shape
        dev.affine_fwd(x, None, b.contiguous().view(1, C), 1.0, M, M, C, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        M, C = dy.shape
        db = dev.chan_sums(dy, None, None, 1, M, 1, C)[0, 0].contiguous()
        return dy, db


class RowLNFn(torch.autograd.Function):
    """y = gamma * (x - mean_row) * rstd_row + beta over the width of each row (eps 1e-5).  Rows of up to 256 floats
    (the per-position LayerNorm over emb_dim in front of every BLSTM) take the one-pass kernels of norm.hip."""

    @staticmethod
    def forward(ctx, x, gamma, beta):
        _need_cuda(x, "TF-GridNet")
        x = x.contiguous()
        M, Wd = x.shape
        d = x.device
        st = _empty(d, M, 2)
        y = torch.empty_like(x)
        g, b = gamma.contiguous().view(-1), beta.contiguous().view(-1)
        ctx.short = dev.rowln_ok(Wd)
        if ctx.short:
            dev.rowln_fwd(x, g, b, M, Wd, y, st)
        else:
            dev.group_stats(x, _cln_geom(M, Wd), st, dev.LN_EPS)
            dev.dwconv_fwd(x, st, g, b, torch.ones(Wd, 1, device=d), torch.zeros(Wd, device=d), M, 1, Wd, 1, 1, 1, y)
        ctx.save_for_backward(x, st, g)
        ctx.shapes = (gamma.shape, beta.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, st, g = ctx.saved_tensors
        M, Wd = x.shape
        if ctx.short:
            dy = dy.contiguous()
            dx = torch.empty_like(dy)
            tot = dev.rowln_bwd(x, dy, st, g, M, Wd, dx)
            return dx, tot[1].contiguous().view(ctx.shapes[0]), tot[0].contiguous().view(ctx.shapes[1])
        dx, dg, db = norm_backward(x, dy.contiguous().clone(), st, g, "cLN", M, 1, Wd)
        return dx, dg.view(ctx.shapes[0]), db.view(ctx.shapes[1])


class QKVHeadsFn(torch.autograd.Function):
    """The attention heads of a block from the ONE projection GEMM's output qkv [B*T*Q, nh*(2E + cp)] (columns: the Q
    heads, the K heads, the V heads): per projection PReLU(head slope) + LayerNorm over (Q, ch) with the head's affine,
    written head-major as [nh*B, T', Q*ch] (T' = T for the queries, Tp >= T with zero rows for keys / values) by one
    kernel each (dev.heads_fwd / heads_bwd, heads.hip; gridnet_block.py:176-199).  Parameters per projection:
    slope [nh], gamma / beta [nh, Q*ch] (index q*ch + e)."""

    @staticmethod
    def forward(ctx, qkv, geo, sq, gq, bq, sk, gk, bk, sv, gv, bv):
        _need_cuda(qkv, "TF-GridNet")
        B, T, Tp, Q, nh, E, cp = geo
        qkv = qkv.contiguous()
        ld, d = qkv.shape[1], qkv.device
        if ld != nh * (2 * E + cp) or qkv.shape[0] != B * T * Q:
            raise dev.L.WesepHipError(f"QKVHeadsFn: projection output {tuple(qkv.shape)} is not [{B * T * Q}, {nh * (2 * E + cp)}]")
        outs, keep, off = [], [], 0
        for s, g, b, ch, tp in ((sq, gq, bq, E, T), (sk, gk, bk, E, Tp), (sv, gv, bv, cp, Tp)):
            s, g, b = s.contiguous(), g.contiguous(), b.contiguous()
            y = _empty(d, nh * B, tp, Q * ch)
            st = _empty(d, nh, B * T, 2)
            dev.heads_fwd(qkv, ld, off, s, g, b, B, T, tp, Q, nh, ch, y, st)
            outs.append(y)
            keep += [s, g, st]
            off += nh * ch
        ctx.save_for_backward(qkv, *keep)
        ctx.geo = geo
        return tuple(outs)

    @staticmethod
    def backward(ctx, dq, dk, dv):
        qkv = ctx.saved_tensors[0]
        keep = ctx.saved_tensors[1:]
        B, T, Tp, Q, nh, E, cp = ctx.geo
        ld = qkv.shape[1]
        dqkv = torch.empty_like(qkv)                      # the three launches cover every column
        grads, off = [], 0
        for i, (dy, ch, tp) in enumerate(((dq, E, T), (dk, E, Tp), (dv, cp, Tp))):
            s, g, st = keep[3 * i:3 * i + 3]
            dg, db, ds = dev.heads_bwd(qkv, ld, off, dy.contiguous(), s, g, st, B, T, tp, Q, nh, ch, dqkv, ld, off)
            grads += [ds, dg, db]
            off += nh * ch
        return (dqkv, None) + tuple(grads)


class GroupLNFn(torch.autograd.Function):
    """nn.GroupNorm(1, C) on [B*P, C]: one mean/variance per batch row over (P, C), affine per channel."""

    @staticmethod
    def forward(ctx, x, gamma, beta, geo):
        _need_cuda(x, "TF-GridNet")
        B, P = geo
        x = x.contiguous()
        C = x.shape[1]
        d = x.device
        st = _empty(d, B, 2)
        dev.flat_stats(x, B, P * C, st, dev.LN_EPS)
        y = torch.empty_like(x)
        dev.dwconv_fwd(x, st, gamma.contiguous(), beta.contiguous(), torch.ones(C, 1, device=d),
                       torch.zeros(C, device=d), B, P, C, 1, 1, P, y)
        ctx.save_for_backward(x, st, gamma.contiguous())
        ctx.geo = (B, P, C)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, st, g = ctx.saved_tensors
        B, P, C = ctx.geo
        dx, dg, db = norm_backward(x, dy.contiguous().clone(), st, g, "gLN", B, P, C)
        return dx, dg, db, None


class PReluFn(torch.autograd.Function):
    """nn.PReLU with one slope a [1]."""

    @staticmethod
    def forward(ctx, x, a):
        x = x.contiguous()
        if x.numel() % 4:
            raise L.WesepHipError("PReLU: the element count must be a multiple of 4")
        y = torch.empty_like(x)
        dev.prelu_fwd(x.view(-1, 4), None, a, x.numel() // 4, 4, x.numel() // 4, y.view(-1, 4))   # elementwise: any 16-byte view
        ctx.save_for_backward(x, a)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, a = ctx.saved_tensors
        dx = torch.empty_like(x)
        da = dev.prelu_bwd(x.view(-1), dy.contiguous().view(-1), a, dx.view(-1))
        return dx, da.view(a.shape)


class SoftmaxFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scale):
        x = x.contiguous()
        y = torch.empty_like(x)
        dev.softmax_rows_fwd(x, x.shape[0], x.shape[1], scale, y)
        ctx.save_for_backward(y)
        ctx.scale = scale
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dx = torch.empty_like(y)
        dev.softmax_rows_bwd(y, dy.contiguous(), y.shape[0], y.shape[1], ctx.scale, dx)
        return dx, None


class MatmulNTFn(torch.autograd.Function):
    """A [M, K] x B [N, K]^T -> [M, N] (both operands differentiable): the attention products."""

    @staticmethod
    def forward(ctx, A, B):
        A, B = A.contiguous(), B.contiguous()
        M, K = A.shape
        N = B.shape[0]
        ctx.save_for_backward(A, B)
        return _gemm(A, M, K, B, N)

    @staticmethod
    def backward(ctx, dC):
        A, B = ctx.saved_tensors
        dC = dC.contiguous()
        M, K = A.shape
        N = B.shape[0]
        dA = _gemm(dC, M, N, _transposed(B, N, K), K)
        if N % 4 == 0:
            dB, _ = _wgrad(dC, M, N, A, K, with_bias=False)                     # dC^T A
        else:                                                                  # (A^T dC)^T: the TN kernel wants Nn % 4 == 0
            dBt, _ = _wgrad(A, M, K, dC, N, with_bias=False, vec=0)
            dB = _transposed(dBt, K, N)
        return dA, dB


class BatchedMatmulNTFn(torch.autograd.Function):
    """A [G, M, K] x B [G, N, K]^T (+ bias [N]) -> C [G, M, N], both operands differentiable: ONE grouped launch over
    the G = heads x batch rows attention problems of a block (gridnet_block.py:176-206) instead of a host loop of G
    launches, forward and backward (ws_gemm_nt / ws_gemm_tn with per-group descriptors).  bias carries no gradient (it
    is the -inf mask of zero-padded key columns)."""

    @staticmethod
    def _nt(A, W, bias, G, M, K, N, w_kn=False):
        """C[g] = A[g] W'[g]^T, W'[n][k] = W[g][n][k] -- or, w_kn (ws_gemm_nt_args.vec bit 3, round 6), = W[g][k][n]: the
        operand as it lies in memory, where rounds 2-5 copied it into the other orientation first."""
        d = A.device
        C_ = _empty(d, G, M, N)
        tab = np.zeros(G, dtype=L.GROUP_NT_DTYPE)
        wp, bp = W.data_ptr(), (bias.data_ptr() if bias is not None else 0)
        for g in range(G):
            tab[g] = (wp + 4 * g * N * K, bp, 0, 0, g * M * K, g * M * N, 0, K, N, N if w_kn else K, 0)
        desc = L.upload_struct_array(tab, d)
        dev.gemm_nt(A=A, a_rows=flat(K), M=M, C_out=C_, c_rows=flat(N), groups=desc, ngroups=G, max_n=N,
                    vec=(3 if (K % 4 == 0 and (M * K) % 4 == 0) else 0) | (8 if w_kn else 0))
        return C_

    @staticmethod
    def nn_ok(K, N, M):
        """The transposed-W staging takes this product (16-byte pieces of A and of W's rows; the split-bf16 kernel)."""
        return (K % 4 == 0 and N % 4 == 0 and (M * K) % 4 == 0 and dev.gemm_mode() == "bf16x3"
                and os.environ.get("WESEP_GEMM_NN", "1") != "0")

    @staticmethod
    def forward(ctx, A, B, bias):
        _need_cuda(A, "TF-GridNet attention")
        A, B = A.contiguous(), B.contiguous()
        G, M, K = A.shape
        N = B.shape[1]
        ctx.save_for_backward(A, B)
        return BatchedMatmulNTFn._nt(A, B, bias, G, M, K, N)

    @staticmethod
    def backward(ctx, dC):
        A, B = ctx.saved_tensors
        dC = dC.contiguous()
        G, M, K = A.shape
        N = B.shape[1]
        d = A.device
        dA = dB = None
        if ctx.needs_input_grad[0]:
            if BatchedMatmulNTFn.nn_ok(N, K, M):                                # dA = dC B with B [N, K] as it lies
                dA = BatchedMatmulNTFn._nt(dC, B, None, G, M, N, K, w_kn=True)
            else:
                Bt = B.transpose(1, 2).contiguous()                             # [G, K, N]: W'[k][n] of dA = dC B
                dA = BatchedMatmulNTFn._nt(dC, Bt, None, G, M, N, K)
        if ctx.needs_input_grad[1]:
            # dB[g] = dC[g]^T A[g]: the TN kernel, one split (M rows), the slab IS the result
            dB = _empty(d, G, N, K)
            tab = np.zeros(G, dtype=L.GROUP_TN_DTYPE)
            for g in range(G):
                tab[g] = (0, 0, g * M * N, g * M * K, 0, g * N * K, 0, N, K, 0, 0)
            desc = L.upload_struct_array(tab, d)
            rps = -(-M // 32) * 32
            dev.gemm_tn(G=dC, g_rows=flat(N), A=A, a_rows=flat(K), M=M, slab=dB, slab_stride=G * N * K, nsplit=1,
                        rows_per_split=rps, groups=desc, ngroups=G, max_n=N, max_k=K,
                        vec=1 if (K % 4 == 0 and (M * K) % 4 == 0) else 0)
        return dA, dB, None


class BatchedMatmulNNFn(torch.autograd.Function):
    """A [G, M, K] x B [G, K, N] -> C [G, M, N], both operands differentiable: att x V of the attention (gridnet_block.py:203-206)
    with V as the head kernel wrote it -- ws_gemm_nt stages the transposed operand itself (vec bit 3).  Backward: dA = dC B^T is
    the NT product with B's rows as W' rows; dB = A^T dC is the TN kernel.  No operand is copied."""

    @staticmethod
    def forward(ctx, A, B):
        _need_cuda(A, "TF-GridNet attention")
        A, B = A.contiguous(), B.contiguous()
        G, M, K = A.shape
        N = B.shape[2]
        ctx.save_for_backward(A, B)
        return BatchedMatmulNTFn._nt(A, B, None, G, M, K, N, w_kn=True)

    @staticmethod
    def backward(ctx, dC):
        A, B = ctx.saved_tensors
        dC = dC.contiguous()
        G, M, K = A.shape
        N = B.shape[2]
        d = A.device
        dA = dB = None
        if ctx.needs_input_grad[0]:
            dA = BatchedMatmulNTFn._nt(dC, B, None, G, M, N, K)                 # W'[k][n] = B[k][n]: B's rows ARE the W' rows
        if ctx.needs_input_grad[1]:
            dB = _empty(d, G, K, N)                                             # dB[g] = A[g]^T dC[g]
            tab = np.zeros(G, dtype=L.GROUP_TN_DTYPE)
            for g in range(G):
                tab[g] = (0, 0, g * M * K, g * M * N, 0, g * K * N, 0, K, N, 0, 0)
            desc = L.upload_struct_array(tab, d)
            rps = -(-M // 32) * 32
            dev.gemm_tn(G=A, g_rows=flat(K), A=dC, a_rows=flat(N), M=M, slab=dB, slab_stride=G * K * N, nsplit=1,
                        rows_per_split=rps, groups=desc, ngroups=G, max_n=K, max_k=N,
                        vec=1 if (N % 4 == 0 and (M * N) % 4 == 0) else 0)
        return dA, dB


def blocked_path_ok(C, ks, hs):
    """The blocked-layout recurrence machinery of the pBSRNN path (functional.ResRNNBlkFn: plain -> BL input
    projection, 16-sequence / cluster / fused-projection recurrences, BL -> plain output projection with bias and
    residual, BL x BL weight gradients) is built for 128 input features and one position per step -- which is the
    shipped TF-GridNet recipe (emb_dim 128, emb_ks = emb_hs = 1; tfgridnet.yaml).  Default for that geometry since its
    first hardware run (round 2: every stage within 1e-5 of an fp64 statement, tools/diag_tfgrid_blk.py);
    WESEP_TFGRID_BLOCKED=0 selects the row-major path."""
    import os
    return os.environ.get("WESEP_TFGRID_BLOCKED", "1") != "0" and C == 128 and ks == 1 and hs == 1


class BlstmLinearBlkFn(torch.autograd.Function):
    """y [nseq*Lr, 128] (layer-normed), res [nseq*Lr, 128] -> res + Linear(BLSTM(y)) on the blocked layout.

    gridnet_block.py:139-160 for emb_ks = emb_hs = 1.  Same kernels and call sequence as functional.ResRNNBlkFn
    without its GroupNorm (the LayerNorm over C happens before, per row).  Sequences are the contiguous runs of Lr
    rows; when the cluster recurrence applies apart from the sequence count (long sequences, few of them: the
    inter-frame path), the sequences are zero-padded to a multiple of 64 -- padded sequences cost no latency and their
    rows are dropped.  Weights: pad_lstm / pad_hidden_cols outputs (hidden zero-padded to 256).

    geo = (nseq, Lr): sequences are contiguous runs of Lr rows; geo = (nseq, Lr, div, s1, s2, step_rows): a STRIDED
    sequence map over the rows of y / res (sequence s, step t -> row (s // div) * s1 + (s % div) * s2 + t * step_rows) --
    the inter-frame path run in place on the [B, T, Q, C] map (round 4; the runtime's plan has done so since round 3): no
    transposed copy of the map before and after the BLSTM, forward or backward; the padding to a multiple of 64 is then
    the map's `nvalid` (wesep_hip.h ws_seqmap), not appended rows."""

    @staticmethod
    def forward(ctx, y, res, geo, dummy, box, wih_f, wih_r, b_f, b_r, whf, whr, lin_w, lin_b):
        """dummy / box: None, or the output and the box of this BLSTM's functional.WGradCarrierFn (created over the eight
        weight tensors, in this order, before the first block of the step) -- then the weight gradients are computed on the
        side stream, released under the NEXT inter-frame BPTT (a latency-bound launch on a quarter of the CUs), and reach
        autograd through the carrier; this node returns None for them (round 4: the twelve BLSTMs' weight-gradient GEMMs,
        48 ms of a 336 ms step, ran in line on the main stream before)."""
        from . import functional as F0
        _need_cuda(y, "TF-GridNet")
        nseq, Lr = geo[:2]
        strided = geo[2:] if len(geo) > 2 else None
        N, H, G4 = 128, HP, G4P
        d = y.device
        pad = 0
        if Lr >= 64 and nseq % 64 and (-(-nseq // 64) * 64 // 32) * 8 <= dev.cu_count(d):
            pad = -(-nseq // 64) * 64 - nseq
        ns = nseq + pad
        if pad and strided is None:
            z = torch.zeros(pad * Lr, N, device=d, dtype=torch.float32)
            y, res = torch.cat([y, z], 0), torch.cat([res, z], 0)
        y, res = y.contiguous(), res.contiguous()
        seq = SeqMap(ns, BIG, 0, Lr, 1, Lr) if strided is None else SeqMap(ns, *strided, Lr, nvalid=nseq)
        nb = dev.bl_num_blocks(seq)
        zero = torch.zeros(G4, device=d, dtype=torch.float32)
        wcat, bcat = _empty(d, 2 * G4, N), _empty(d, 2 * G4)
        dev.lstm_cat_ih(wih_f.contiguous(), wih_r.contiguous(), b_f.contiguous(), zero, b_r.contiguous(), zero, N, wcat,
                        bcat)
        pack_f, pack_b = _empty(d, L.LSTM_PACK_FLOATS), _empty(d, L.LSTM_PACK_FLOATS)
        lmode = dev.lstm_blk_mode(ns)
        whf, whr = whf.contiguous(), whr.contiguous()
        dev.lstm_pack(whf, whr, pack_f, pack_b, lmode)
        xn = _empty(d, nb, 32 * N)
        cbuf, hcat = _empty(d, nb, 32 * 2 * H), _empty(d, nb, 32 * 2 * H)
        cluster = dev.lstm_cluster_ok(seq, d)
        kind = F0._bptt_kind(seq, d, cluster)
        # storage of the saved gates / d(gates): as in functional.ResRNNBlkFn (wesep_hip.h WS_GATES_*) -- unorm16 gates in a
        # buffer of half the bytes by default (the twelve BLSTMs' saved gates were 77 of the step's 155 GB in round 3)
        gfmt = L.GATES_F32 if kind == "cluster" else dev.gates_fmt()
        h2 = gfmt != L.GATES_F32
        gates = _empty(d, dev.blh_floats(nb, 2 * G4)) if h2 else _empty(d, nb, 32 * 2 * G4)
        # fp16 copies of [xn | h] for the weight-gradient GEMMs (functional.ResRNNBlkFn; ws_gemm_tnb a_fmt = 1, ABI v16): the default
        # since round 6 (WESEP_TFG_TNB_A16=0 turns it off).  Round 5 left it opt-in -- "these GEMMs already hide under the
        # inter-frame BPTTs": 304.3 vs 310.0 ms for 8 GB more saved state -- but hidden work is not free work on this chip
        # (profiles/r06_side_stream_tax.md: the clock follows the load): one MFMA per product instead of three on the side stream
        # is 263.6 -> 248.7 ms per step at config 5's per-GPU shape (two runs each, one box), 123 -> 131 GB
        a16 = gfmt == L.GATES_H2F and any(ctx.needs_input_grad) and F0.tnb_a16() and \
            os.environ.get("WESEP_TFG_TNB_A16", "1") != "0"
        xn16 = _empty(d, dev.blh_floats(nb, N)) if a16 else None
        hcat16 = _empty(d, dev.blh_floats(nb, 2 * H)) if a16 else None
        if dev.lstm_fuse_ok(ns, cluster):
            dev.gemm_p2b(A=y, lda=N, sm=seq, Wpack=None, N=0, C_out=None, A_bl=xn, A_bl16=xn16)
            fpack = _empty(d, L.LSTM_FUSED_PACK_FLOATS)
            hf = dev.lstm_fused_hfmt(gfmt)      # round 6: fp16 h in the recurrent part (two terms), as functional.ResRNNBlkFn
            dev.lstm_pack_fused(wih_f.contiguous(), wih_r.contiguous(), whf, whr, fpack, hfmt=hf)
            dev.lstm_fwd_fused(gates, cbuf, hcat, xn, fpack, bcat, seq, gfmt=gfmt, hfmt=hf)
        elif cluster and h2 and dev.lstm_cluster2_on() and os.environ.get("WESEP_TFG_CLUSTER2", "1") != "0":
            # inter-frame path, 2-byte formats (round 5): ws_lstm_fwd_cluster2 computes x W_ih^T itself from the split-pair
            # rows ws_gemm_p2b relays into BL(128) (functional.ResRNNBlkFn; lstm_cluster2.hip); the fp32 pre-activations exist
            # only inside the predicated fall-back behind the launch.  Measured at BASELINE config 5's geometry: 301.0 ->
            # 277.7 ms/step (same box, together with the fp16 pair BPTT: profiles/r05_ab/r05_c11_tfg_*.json), every parity figure
            # of the recipe-geometry test unchanged (waveform 1.27e-5 -> 1.33e-5, worst gradient 3.4e-3 -> 3.1e-3, median
            # 4.1e-4 -> 4.2e-4: profiles/r05_tfg_cfg5_cluster2_bls_input.txt).  The kernel's FIRST cut -- fp16 copy of the input,
            # two-term x-projection -- was not: median gradient error 7.0e-4 and two scalar PReLU slopes over this model's 5e-3
            # bound (profiles/r05_tfg_cfg5_precision_split.txt); the input keeps its split pair since.  WESEP_TFG_CLUSTER2=0
            # selects the round-4 cluster kernel on fp32 pre-activations
            dev.gemm_p2b(A=y, lda=N, sm=seq, Wpack=None, N=0, C_out=None, A_bl=xn, A_bl16=xn16)
            tw = dev.lstm_fwd_cluster2(gates, cbuf, hcat, xn, wcat, bcat, whf, whr, seq, dbg=F0._cluster_dbg())
            wih_pack = _empty(d, 2 * G4 * N)
            dev.pack_w(wcat, 2 * G4, N, N, wih_pack, order=0)
            pre = dev.fallback_scratch(d, nb * 32 * 2 * G4)      # (untouched after a clean launch; one buffer per stream)
            dev.gemm_p2b(A=y, lda=N, sm=seq, Wpack=wih_pack, N=2 * G4, C_out=pre, bias=bcat, run_if=tw)
            dev.lstm_fwd(gates, cbuf, hcat, pack_f, seq, lmode, run_if=tw, gfmt=gfmt, gates_in=pre)
            del pre
        else:
            wih_pack = _empty(d, 2 * G4 * N)
            dev.pack_w(wcat, 2 * G4, N, N, wih_pack, order=0)
            pre = _empty(d, nb, 32 * 2 * G4) if h2 else gates      # 2-byte formats: pre-activations in a scratch buffer
            xproj = dict(A=y, lda=N, sm=seq, Wpack=wih_pack, N=2 * G4, C_out=pre, bias=bcat, A_bl=xn, A_bl16=xn16)
            rec = dict(gfmt=gfmt, gates_in=pre) if h2 else {}
            dev.gemm_p2b(**xproj)
            if cluster:
                # the streaming pair behind the cluster launch is predicated on its timeout word (wesep_hip.h): empty
                # launches after a clean run, the whole layer again if the workgroups were not co-resident (the 2-byte
                # formats leave the pre-activations intact: only the recurrence is repeated)
                tw = dev.lstm_fwd_cluster(gates, cbuf, hcat, whf, whr, seq, dbg=F0._cluster_dbg(), **rec)
                if not h2:
                    dev.gemm_p2b(run_if=tw, **xproj)
                dev.lstm_fwd(gates, cbuf, hcat, pack_f, seq, lmode, run_if=tw, **rec)
            else:
                dev.lstm_fwd(gates, cbuf, hcat, pack_f, seq, lmode, **rec)
            del pre
        lw = lin_w.contiguous()
        lin_pack = _empty(d, N * 2 * H)
        dev.pack_w(lw, N, 2 * H, 2 * H, lin_pack, order=1)
        out = torch.empty_like(res)
        dev.gemm_b2p(A=hcat, K=2 * H, sm=seq, Wpack=lin_pack, C_out=out, ldc=N, bias=lin_b.contiguous(), R=res, a16_out=hcat16)
        # the backward's weight packs are built HERE, where the GPU serves one stream: built lazily in the backward, these
        # 5 us launches queue behind the side stream's chip-filling weight-gradient GEMMs for up to a millisecond each
        # (functional.ResRNNBlkFn does the same)
        bw_packs = None
        if any(ctx.needs_input_grad):
            g2 = gfmt == L.GATES_H2F
            wlt_pack = _empty(d, 2 * H * N)
            dev.pack_w(lw, 2 * H, N, 2 * H, wlt_pack, trans=True, order=0)
            wct_pack = _empty(d, N * 2 * G4)
            # (d(xn) from the scaled-fp16 d(gates): fp16 hi / lo, or fp16 hi + FP8 lo fragments -- functional.dxn_fmt, round 6)
            dev.pack_w(wcat, N, 2 * G4, N, wct_pack, trans=True, order=1, f16=(2 if F0.dxn_fmt(2) == 3 else 1) if g2 else 0)
            ppack = None
            if kind == "pair":
                ppack = _empty(d, L.LSTM_PACK_FLOATS)
                dev.lstm_pack_pair(whf, whr, ppack, f16=F0.pair_rfmt(gfmt))     # (rfmt 1 / 2: fp16 hi, fp16 / FP8 lo of 256 w)
            elif kind == "stream" and F0.band_rfmt(gfmt, lmode):                # the streaming BPTT's rfmt 2 pack (round 6 default)
                ppack = _empty(d, L.LSTM_PACK_FLOATS)
                dev.lstm_pack_bwd_f8(whf, whr, ppack)
            bw_packs = (wlt_pack, wct_pack, ppack)
        ctx.bw_packs = bw_packs
        ctx.save_for_backward(gates, cbuf, hcat, xn16 if a16 else xn, wcat, pack_b, lw, whf, whr, hcat16)
        ctx.geo = (nseq, Lr, ns, lmode, cluster)
        ctx.seq = seq
        ctx.gfmt, ctx.kind = gfmt, kind
        ctx.F0 = F0
        ctx.box = box
        ctx.consumed = False
        return out[:nseq * Lr] if (pad and strided is None) else out

    @staticmethod
    def backward(ctx, dout):
        import os
        gates, cbuf, hcat, xn, wcat, pack_b, lw, whf, whr, hcat16 = ctx.saved_tensors     # (xn: its fp16 copy if hcat16)
        if ctx.consumed:         # same contract as functional.ResRNNBlkFn: BPTT turns the saved gates into d(gates) in place
            raise L.WesepHipError("TF-GridNet BLSTM: second backward through the same graph (retain_graph / multi-loss "
                                  "loops): the blocked path consumes its saved gates in place; run the forward again")
        ctx.consumed = True
        nseq, Lr, ns, lmode, cluster = ctx.geo
        N, H, G4 = 128, HP, G4P
        d = dout.device
        dout = dout.contiguous()
        dres = dout
        seq = ctx.seq
        appended = ns != nseq and not seq.nvalid      # contiguous runs: the padding sequences are rows behind the data
        if appended:
            dout = torch.cat([dout, torch.zeros((ns - nseq) * Lr, N, device=d, dtype=torch.float32)], 0)
        nb = dev.bl_num_blocks(seq)
        wlt_pack, wct_pack, ppack = ctx.bw_packs
        dh, dout_bl = _empty(d, nb, 32 * 2 * H), _empty(d, nb, 32 * N)
        amax = ctx.F0.amax_word(d) if ctx.gfmt == L.GATES_H2F else None   # (functional.py)
        dev.gemm_p2b(A=dout, lda=N, sm=seq, Wpack=wlt_pack, N=2 * H, C_out=dh, A_bl=dout_bl, amax=amax)
        # BPTT works in place on the saved gates (6.4 GB per BLSTM at the recipe's 8 rows x 6 s: a clone here was 12 copies
        # = 38 ms of a 490 ms step and the largest transient allocation of the backward)
        kind, gfmt = ctx.kind, ctx.gfmt
        g_fmt = {L.GATES_H2: 1, L.GATES_H2F: 2}.get(gfmt, 0)
        F0, box = ctx.F0, ctx.box
        # the inter-frame BPTT (few long sequences: pair / cluster kernels on a fraction of the CUs for ~10 ms) is where the
        # weight-gradient jobs deferred by the BLSTMs before it are released (functional.flush_deferred_wgrads)
        ready = F0.mark_wgrads_ready(d) if kind in ("pair", "cluster") else None
        if kind == "cluster":
            dg = gates
            dev.lstm_bwd_cluster(gates, cbuf, dh, whf, whr, seq)
        elif kind == "pair":                                    # few long sequences (the inter-frame path): lstm_pair.hip
            if gfmt == L.GATES_F32:
                dg = gates
                dev.lstm_bwd_pair(gates, cbuf, dh, ppack, seq, dbg=ctx.F0._pair_dbg())
            else:
                # d(gates) out of place, the streaming BPTT predicated on the launch's time-out word behind it (functional.py)
                dg = _empty(d, nb, 32 * 2 * G4) if gfmt == L.GATES_H2S else _empty(d, dev.blh_floats(nb, 2 * G4))
                tw = dev.lstm_bwd_pair(gates, cbuf, dh, ppack, seq, gfmt=gfmt, dgates=dg, repairable=True,
                                       dbg=ctx.F0._pair_dbg(), amax=amax, rfmt=ctx.F0.pair_rfmt(gfmt))
                dev.lstm_bwd(gates, cbuf, hcat, dh, pack_b, seq, lmode, gfmt=gfmt, dgates=dg, run_if=tw, amax=amax)
        elif gfmt == L.GATES_F32:
            dg = gates
            dev.lstm_bwd(gates, cbuf, hcat, dh, pack_b, seq, lmode)
        else:
            dg = _empty(d, nb, 32 * 2 * G4) if gfmt == L.GATES_H2S else gates
            brf = F0.band_rfmt(gfmt, lmode) if ppack is not None else 0
            dev.lstm_bwd(gates, cbuf, hcat, dh, ppack if brf else pack_b, seq, lmode, gfmt=gfmt,
                         dgates=dg if gfmt == L.GATES_H2S else None, amax=amax, rfmt=brf)
        if ready is not None:
            F0.flush_deferred_wgrads(d, ready)
        order = (0, 4, 2, 6, 1, 5, 8, 9)       # _weight_grads' list -> this node's weight arguments
        if box is not None:
            def job(side, gates=dg, xn=xn, hcat=hcat, dout_bl=dout_bl, seq=seq, nb=nb, box=box, g_fmt=g_fmt, amax=amax,
                    hcat16=hcat16, prod=torch.cuda.current_stream()):
                wg_ = F0.ResRNNBlkFn._weight_grads(gates, xn, hcat, dout_bl, seq, nb, 128, g_fmt, amax, hcat16)
                box.grads = [wg_[i] for i in order]
                box.event = torch.cuda.Event()
                box.event.record(side)
                F0.keep_for_side(box, (gates, xn, hcat, dout_bl, amax, hcat16), side, prod)
            F0.defer_wgrad(d, job)
            wgo = [None] * 8
        else:
            wg = F0.ResRNNBlkFn._weight_grads(dg, xn, hcat, dout_bl, seq, nb, N, g_fmt, amax, hcat16)
            wgo = [wg[i] for i in order]
        dy = _empty(d, (ns if appended else nseq) * Lr, N)
        dev.gemm_b2p(A=dg, K=2 * G4, sm=seq, Wpack=wct_pack, C_out=dy, ldc=N, a_fmt=F0.dxn_fmt(g_fmt), amax=amax)
        if appended:
            dy = dy[:nseq * Lr]
        # _weight_grads: [dW_ih_f, dW_hh_f, db_f, db_f (clone), dW_ih_r, dW_hh_r, db_r, db_r (clone), dW_lin, db_lin]
        gd = torch.zeros((), device=d) if box is not None else None     # keeps the carrier node in the graph walk
        return (dy, dres, None, gd, None) + tuple(wgo)
