"""Autograd shims of the wespeaker ResNet speaker encoder over the C ABI.

Channels-last activations [R*H*W, C] (H = mel bin, W = frame).  conv2d = im2col + one split-bf16 MFMA GEMM
(K = k*k*Cin); BatchNorm2d / ReLU / residual are the channels-last BatchNorm kernels of tasnet.hip (a ReLU is a
PReLU of slope 0, "no activation" one of slope 1); the input gradient is GEMM + col2im (gather), the weight
gradient the TN GEMM on the recomputed patch matrix (patches are never kept for the backward)."""
import torch

from . import _lib as L
from . import dev
from . import functional_conv as FC
from .functional import _empty, _need_cuda
from .functional_tasnet import _gemm, _transposed, _wgrad


class ConvBnActFn(torch.autograd.Function):
    """x [R*H*W, Cin] -> act(BN(conv2d(x, w)) (+ res)) [R*Ho*Wo, Cout]; act = ReLU or identity.  `stride` of geo: an int,
    or (sh, sw) for the mel-axis-only strides of CAM++'s FCM head (implicit-patch path, Cin % 4 == 0)."""

    @staticmethod
    def forward(ctx, x, res, geo, w, gamma, beta, rm, rv):
        _need_cuda(x, "ResNet speaker encoder")
        R, H, W, stride, relu, training = geo
        Cout, Cin, k, _ = w.shape
        pad = k // 2
        sh, sw = stride if isinstance(stride, tuple) else (stride, stride)
        if sh != sw and not (FC.implicit_ok(Cin) and FC.implicit_ok(Cout) and max(sh, sw) <= 2):
            raise L.WesepHipError("ConvBnActFn: unequal strides need channel counts that are multiples of 4, strides <= 2")
        Ho, Wo = dev.conv_out(H, k, sh, pad), dev.conv_out(W, k, sw, pad)
        M = R * Ho * Wo
        x = x.contiguous()
        d = x.device
        Kk = k * k * Cin
        ldp = -(-Kk // 4) * 4                                   # first layer: 9 taps padded to 12 columns
        W2 = torch.zeros(Cout, ldp, device=d, dtype=torch.float32)
        W2[:, :Kk] = w.permute(0, 2, 3, 1).reshape(Cout, Kk)     # column (ky*k + kx)*Cin + c, like the patches
        if FC.implicit_ok(Cin):
            # the patch matrix stays implicit in the GEMM's operand loader (functional_conv): nothing is unfolded
            c = FC.conv2d_fwd(x, R, H, W, Cin, W2, Cout, k, sh, sw, pad)
        else:                                                   # single-channel first layer: 12-column patch rows
            patches = ConvBnActFn._patches(x, R, H, W, Cin, k, sh, pad, M, ldp)
            c = _gemm(patches, M, ldp, W2, Cout)
            del patches
        st = _empty(d, 2, Cout)
        if training:
            dev.bn_stats(c, M, Cout, rm, rv, st)
        else:
            st[0].copy_(rm)
            st[1].copy_(torch.rsqrt(rv + dev.BN_EPS))
        slope = torch.full((1,), 0.0 if relu else 1.0, device=d, dtype=torch.float32)
        u, y = _empty(d, M, Cout), _empty(d, M, Cout)
        dev.bn_prelu_fwd(c, st, gamma, beta, res.contiguous() if res is not None else None, slope, M, Cout, u, y)
        ctx.save_for_backward(x, c, st, u, W2, gamma, slope)
        ctx.geo = (R, H, W, Cin, Cout, k, (sh, sw), pad, Ho, Wo, ldp, res is not None, training, w.shape)
        return y

    @staticmethod
    def _patches(x, R, H, W, Cin, k, stride, pad, M, ldp):
        if ldp != k * k * Cin:
            patches = torch.zeros(M, ldp, device=x.device, dtype=torch.float32)
        else:
            patches = _empty(x.device, M, ldp)
        dev.im2col(x, R, H, W, Cin, k, stride, pad, patches, ldp)
        return patches

    @staticmethod
    def backward(ctx, dy):
        x, c, st, u, W2, gamma, slope = ctx.saved_tensors
        R, H, W, Cin, Cout, k, (sh, sw), pad, Ho, Wo, ldp, has_res, training, wshape = ctx.geo
        stride = sh
        M = R * Ho * Wo
        d = x.device
        du = dy.contiguous().clone()
        dev.prelu_bwd(u, du, slope, du)                          # ReLU' (slope 0) or identity (slope 1), in place
        dres = du if has_res else None
        dc = _empty(d, M, Cout)
        sums = dev.bn_bwd_any(c, du, st, gamma, M, Cout, dc, training)
        Kk = k * k * Cin
        if FC.implicit_ok(Cin):
            dW2, _ = FC.conv2d_wgrad(dc, x, R, H, W, Cin, Cout, k, sh, sw, pad, with_bias=False)
        else:
            patches = ConvBnActFn._patches(x, R, H, W, Cin, k, stride, pad, M, ldp)
            dW2, _ = _wgrad(dc, M, Cout, patches, ldp, with_bias=False)
            del patches
        dw = dW2[:, :Kk].reshape(Cout, k, k, Cin).permute(0, 3, 1, 2).contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            if FC.implicit_ok(Cin) and FC.implicit_ok(Cout) and max(sh, sw) <= 2:
                # transposed view of dc, one row per input pixel: Wd[ci][(tap)*Cout + co] = w[co, ci, ky, kx]
                Wd = W2.view(Cout, k * k, Cin).permute(2, 1, 0).reshape(Cin, k * k * Cout).contiguous()
                dx = FC.conv2d_dx(dc, R, H, W, Cin, Wd, Cout, k, sh, sw, pad)
            else:
                dpatches = _gemm(dc, M, Cout, _transposed(W2, Cout, ldp), ldp)
                dx = _empty(d, R * H * W, Cin)
                dev.col2im(dpatches, R, H, W, Cin, k, stride, pad, dx)
        return dx, dres, None, dw.view(wshape), sums[1].contiguous(), sums[0].contiguous(), None, None


class TstpFn(torch.autograd.Function):
    """x [R*F*T, C] -> [R, 2*C*F]: mean || sqrt(unbiased var + 1e-7) over T, feature index c*F + f."""

    @staticmethod
    def forward(ctx, x, geo):
        R, Fq, T = geo
        x = x.contiguous()
        Cc = x.shape[1]
        stats = _empty(x.device, R, 2 * Cc * Fq)
        dev.tstp_fwd(x, R, Fq, T, Cc, stats)
        ctx.save_for_backward(x, stats)
        ctx.geo = (R, Fq, T, Cc)
        return stats

    @staticmethod
    def backward(ctx, dstats):
        x, stats = ctx.saved_tensors
        R, Fq, T, Cc = ctx.geo
        dx = torch.empty_like(x)
        dev.tstp_bwd(x, stats, dstats.contiguous(), R, Fq, T, Cc, dx)
        return dx, None
