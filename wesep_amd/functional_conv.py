"""2-D convolutions on channels-last images [B*H*W, C] as ONE split-bf16 MFMA GEMM each, with the im2col patch matrix
implicit in the GEMM's operand loader (`dev.ConvView`, include/wesep_hip.h ws_conv_view): forward, input gradient and
weight gradient of Conv2d and ConvTranspose2d never write or read the k*k-times larger patch matrix (round 1 did:
20 % of the DPCCN step was im2col and the GEMMs streamed 9x the activation bytes).

  Conv2d           y  = view0(x)  W2^T          W2[co][(tap)*Cin + ci]  = w[co, ci, ky, kx]
                   dx = view1(dy) Wd^T          Wd[ci][(tap)*Cout + co] = w[co, ci, ky, kx]
                   dW2 = dy^T view0(x)
  ConvTranspose2d  y  = view1(x)  Wt^T          Wt[co][(tap)*Cin + ci]  = w[ci, co, ky, kx]
                   dx = view0(dy) Wx^T          Wx[ci][(tap)*Cout + co] = w[ci, co, ky, kx]
                   dWx^T = x^T view0(dy)
view0 = convolution view (input pixel o*s + tap - p), view1 = transposed view (input pixel (o + p - tap)/s when exact).
Channel counts that are not a multiple of 4 (the single-channel first layer of the ResNet) use the explicit patch matrix.
Reference lines: wesep/modules/dpccn/convs.py:28-110 (Conv2dBlock / ConvTrans2dBlock / DenseBlock)."""
import os

import torch

from . import dev
from .dev import ConvView
from .functional import _empty, _reduce_new
from .functional_tasnet import _gemm, _wgrad

MODE = "bf16x3"   # the implicit operand exists in the split-bf16 kernels only


def implicit_ok(C):
    return C % 4 == 0


def _out(n, k, s, p, dil):
    return (n + 2 * p - dil * (k - 1) - 1) // s + 1


def conv2d_fwd(x, B, H, W, Cin, W2, Cout, k, sh, sw, p, bias=None, dil=1, act=0):
    """x [B*H*W, Cin], W2 [Cout, k*k*Cin] -> [B*Ho*Wo, Cout] (act: the GEMM epilogue's activation, 2 = ReLU)."""
    Ho, Wo = _out(H, k, sh, p, dil), _out(W, k, sw, p, dil)
    return _gemm(x, B * Ho * Wo, k * k * Cin, W2, Cout, bias=bias, act=act, vec=3, mode=MODE,
                 conv=ConvView(0, H, W, Cin, Ho, Wo, k, sh, sw, p, dil))


def _one_pass_wgrad(G, M, Nn, X, conv, with_bias):
    """G^T view0(X) through the one-pass kernel (few gradient columns, all patch columns per workgroup)."""
    Kk = conv.k * conv.k * conv.C
    tiles = -(-M // 32)
    nsplit = max(1, min(512, tiles // 8))                  # >= 8 tiles per workgroup, up to two waves of workgroups
    tps = -(-tiles // nsplit)
    nsplit = -(-tiles // tps)
    d = G.device
    slab = _empty(d, nsplit, Nn * Kk)
    bslab = _empty(d, nsplit, Nn) if with_bias else None
    dev.conv_wgrad(G=G, ldg=Nn, X=X, M=M, Nn=Nn, conv=conv, slab=slab, nsplit=nsplit, tiles_per_split=tps, bslab=bslab)
    dW = _reduce_new(slab, nsplit, Nn * Kk, (Nn, Kk))
    db = _reduce_new(bslab, nsplit, Nn, (Nn,)) if with_bias else None
    return dW, db


def halo_wgrad_ok(Cin, Cout, k, sh, sw, p, dil=1):
    """The halo-tile weight gradient (conv3x3.hip) takes this convolution: 3 x 3, padding 1, stride (1, 1) or (1, 2)."""
    return (k, sh, p, dil) == (3, 1, 1, 1) and sw in (1, 2) and Cin % 4 == 0 and Cout % 4 == 0 and \
        os.environ.get("WESEP_CONV3X3_WGRAD", "1") != "0"


def halo_wgrad(G, Nn, X, ldx, B, H, W, Cin, with_bias, sw=1, Wx=0, ldg=0, g_off=0):
    """dW2 [Nn, 9*Cin] (+ db) of a 3 x 3 / padding 1 convolution with stride (1, sw): G [B*H*W, Nn] gradient rows on the
    output grid (or columns [g_off, g_off + Nn) of rows of stride ldg), X the image [B, H, Wx] with pixel stride ldx (its
    first Cin channels)."""
    tiles = dev.conv3x3_wgrad_tiles(B, H, W)
    groups = (-(-Cin // 32)) * (-(-Nn // 32))              # workgroups per split: one per (input chunk, output tile)
    # ~4 workgroups per CU in flight, >= 4 tiles each.  ROUNDED DOWN: two workgroups of this kernel fit a CU, 512 are resident,
    # and ceil(1024 / 3) * 3 = 1026 workgroups ran a third round for the last two (round 6: the 80-channel layers' 3.1 ms launches)
    nsplit = max(1, min(tiles // 4, max(1, 1024 // groups)))
    tps = -(-tiles // nsplit)
    nsplit = -(-tiles // tps)
    d = G.device
    slab = _empty(d, nsplit, Nn * 9 * Cin)
    bslab = _empty(d, nsplit, Nn) if with_bias else None
    dev.conv3x3_wgrad(G=G, ldg=ldg or Nn, X=X, ldx=ldx, B=B, H=H, Wd=W, Cin=Cin, Nn=Nn, slab=slab, nsplit=nsplit,
                      tiles_per_split=tps, bslab=bslab, sw=sw, Wx=Wx or W, g_off=g_off)
    dW = _reduce_new(slab, nsplit, Nn * 9 * Cin, (Nn, 9 * Cin))
    db = _reduce_new(bslab, nsplit, Nn, (Nn,)) if with_bias else None
    return dW, db


def conv2d_wgrad(dy, x, B, H, W, Cin, Cout, k, sh, sw, p, with_bias=True, dil=1):
    """dW2 [Cout, k*k*Cin] = dy^T view0(x) (+ db)."""
    Ho, Wo = _out(H, k, sh, p, dil), _out(W, k, sw, p, dil)
    if halo_wgrad_ok(Cin, Cout, k, sh, sw, p, dil):
        return halo_wgrad(dy, Cout, x, Cin, B, Ho, Wo, Cin, with_bias, sw, W)
    conv = ConvView(0, H, W, Cin, Ho, Wo, k, sh, sw, p, dil)
    if dev.conv_wgrad_ok(Cout, conv):
        return _one_pass_wgrad(dy, B * Ho * Wo, Cout, x, conv, with_bias)
    return _wgrad(dy, B * Ho * Wo, Cout, x, k * k * Cin, with_bias=with_bias, vec=1, mode=MODE, conv=conv)


def conv2d_dx(dy, B, H, W, Cin, Wd, Cout, k, sh, sw, p, dil=1):
    """dy [B*Ho*Wo, Cout], Wd [Cin, k*k*Cout] -> dx [B*H*W, Cin]: the transposed view of dy, one row per INPUT pixel."""
    Ho, Wo = _out(H, k, sh, p, dil), _out(W, k, sw, p, dil)
    return _gemm(dy, B * H * W, k * k * Cout, Wd, Cin, vec=3, mode=MODE,
                 conv=ConvView(1, Ho, Wo, Cout, H, W, k, sh, sw, p, dil))


def conv2d_weights(w):
    """w [Cout, Cin, k, k] -> (W2 [Cout, k*k*Cin], Wd [Cin, k*k*Cout])."""
    Cout, Cin, k, _ = w.shape
    W2 = w.permute(0, 2, 3, 1).reshape(Cout, k * k * Cin).contiguous()
    Wd = w.permute(1, 2, 3, 0).reshape(Cin, k * k * Cout).contiguous()
    return W2, Wd


def convT2d_fwd(x, B, H, W, Cin, Wt, Cout, k, sh, sw, p, bias=None):
    """x [B*H*W, Cin], Wt [Cout, k*k*Cin] -> [B*Ht*Wt, Cout], Ht = (H - 1)*sh - 2p + k."""
    Ht, Wt_ = (H - 1) * sh - 2 * p + k, (W - 1) * sw - 2 * p + k
    return _gemm(x, B * Ht * Wt_, k * k * Cin, Wt, Cout, bias=bias, vec=3, mode=MODE,
                 conv=ConvView(1, H, W, Cin, Ht, Wt_, k, sh, sw, p))


def convT2d_dx(dy, B, H, W, Cin, Wx, Cout, k, sh, sw, p):
    """dy [B*Ht*Wt, Cout], Wx [Cin, k*k*Cout] -> dx [B*H*W, Cin]: the convolution view of dy."""
    Ht, Wt_ = (H - 1) * sh - 2 * p + k, (W - 1) * sw - 2 * p + k
    return _gemm(dy, B * H * W, k * k * Cout, Wx, Cin, vec=3, mode=MODE,
                 conv=ConvView(0, Ht, Wt_, Cout, H, W, k, sh, sw, p))


def convT2d_wgrad(x, dy, B, H, W, Cin, Cout, k, sh, sw, p):
    """dWx^T [Cin, k*k*Cout] = x^T view0(dy)."""
    Ht, Wt_ = (H - 1) * sh - 2 * p + k, (W - 1) * sw - 2 * p + k
    if halo_wgrad_ok(Cout, Cin, k, sh, sw, p):      # image = dy [Ht, Wt_, Cout], gradient rows = x on the [H, W] grid
        return halo_wgrad(x, Cin, dy, Cout, B, H, W, Cout, False, sw, Wt_)[0]
    conv = ConvView(0, Ht, Wt_, Cout, H, W, k, sh, sw, p)
    if dev.conv_wgrad_ok(Cin, conv):
        return _one_pass_wgrad(x, B * H * W, Cin, dy, conv, False)[0]
    dWxT, _ = _wgrad(x, B * H * W, Cin, dy, k * k * Cout, with_bias=False, vec=1, mode=MODE, conv=conv)
    return dWxT
