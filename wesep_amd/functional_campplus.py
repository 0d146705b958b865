"""Autograd shims of the wespeaker CAM++ speaker encoder (`CAMPPlus`; SURVEY section 8 row a12: the recipe's alternative
encoder, examples/librimix/tse/v2/confs/bsrnn.yaml:66-74, instantiated at wesep/models/bsrnn.py:217) over the C ABI.

Channels-last activations [R*T, C] (row = frame).  CAM++'s D-TDNN layers are pre-activation (BatchNorm -> ReLU -> Conv1d),
so the pieces compose freely: `BnActFn` is the channels-last BatchNorm kernel pair of tasnet.hip with a ReLU (PReLU of
slope 0) or no activation, `Conv1dFn` one split-bf16 GEMM -- k = 1 on the rows, k = 3 / 5 (dilated, strided) on the
implicit patch matrix of the one-row image [R][1][T][C] (functional_conv; the taps of the other kernel rows fall outside
the image).  The context-aware mask pools each 100-frame segment (`ws_seg_sums`) and multiplies the mask back segment
by segment (`ws_seg_scale`); both gradients of that product are the same two kernels."""
import torch

from . import _lib as L
from . import dev
from . import functional_conv as FC
from .functional import _empty, _need_cuda
from .functional_tasnet import _gemm, _wgrad


class BnActFn(torch.autograd.Function):
    """act(BatchNorm1d(x)) on the rows of [M, C]; gamma / beta None: affine = False; act = ReLU or identity."""

    @staticmethod
    def forward(ctx, x, gamma, beta, rm, rv, training, relu):
        _need_cuda(x, "CAM++")
        x = x.contiguous()
        M, Cc = x.shape
        d = x.device
        affine = gamma is not None
        if not affine:
            gamma, beta = torch.ones(Cc, device=d, dtype=torch.float32), torch.zeros(Cc, device=d, dtype=torch.float32)
        st = _empty(d, 2, Cc)
        if training:
            dev.bn_stats(x, M, Cc, rm, rv, st)
        else:
            st[0].copy_(rm)
            st[1].copy_(torch.rsqrt(rv + dev.BN_EPS))
        slope = torch.full((1,), 0.0 if relu else 1.0, device=d, dtype=torch.float32)
        u, y = _empty(d, M, Cc), _empty(d, M, Cc)
        dev.bn_prelu_fwd(x, st, gamma, beta, None, slope, M, Cc, u, y)
        ctx.save_for_backward(x, st, gamma, y)
        ctx.flags = (training, relu, affine)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, st, gamma, y = ctx.saved_tensors
        training, relu, affine = ctx.flags
        M, Cc = x.shape
        du = dy.contiguous()
        if relu:
            du = du.clone()
            dev.relu_mask(du, y)                                  # ReLU' from its saved output
        dx = torch.empty_like(x)
        sums = dev.bn_bwd_any(x, du, st, gamma, M, Cc, dx, training)
        if not affine:
            return dx, None, None, None, None, None, None
        return dx, sums[1].contiguous(), sums[0].contiguous(), None, None, None, None


class Conv1dFn(torch.autograd.Function):
    """x [R*T, Cin] -> conv1d(x, w [Cout, Cin, k], b; stride, dilation, padding dil * (k // 2)) [R*To, Cout].  Also the
    dilated convolutions of Conv-TasNet's Deep encoder / decoder (modules/tasnet)."""

    @staticmethod
    def forward(ctx, x, geo, w, b):
        _need_cuda(x, "CAM++")
        R, T, stride, dil = geo
        Cout, Cin, k = w.shape
        x = x.contiguous()
        p = dil * (k // 2)
        if k == 1 and stride == 1:
            To = T
            W2 = w.reshape(Cout, Cin).contiguous()
            y = _gemm(x, R * T, Cin, W2, Cout, bias=b)
        else:
            if Cin % 4 or Cout % 4 or k % 2 == 0 or k > 5 or stride > 2:
                # (k <= 5: the weight-gradient entry points are built for at most 5 x 5 taps -- refuse in the forward,
                #  not halfway through the backward)
                raise L.WesepHipError(f"Conv1dFn: channels % 4, odd kernel <= 5, stride <= 2 (got {Cin}, {Cout}, {k}, {stride})")
            To = (T + 2 * p - dil * (k - 1) - 1) // stride + 1
            # the Conv1d as the middle kernel row of a k x k view of the one-row image; the other rows are masked taps
            W2 = torch.zeros(Cout, k, k, Cin, device=x.device, dtype=torch.float32)
            W2[:, k // 2] = w.permute(0, 2, 1)
            W2 = W2.view(Cout, k * k * Cin)
            y = FC.conv2d_fwd(x, R, 1, T, Cin, W2, Cout, k, 1, stride, p, bias=b, dil=dil)
        ctx.save_for_backward(x, W2)
        ctx.geo = (R, T, To, stride, dil, p, Cin, Cout, k, w.shape, b is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W2 = ctx.saved_tensors
        R, T, To, stride, dil, p, Cin, Cout, k, wshape, has_b = ctx.geo
        dy = dy.contiguous()
        if k == 1 and stride == 1:
            dW2, db = _wgrad(dy, R * T, Cout, x, Cin, with_bias=has_b)
            dw = dW2.view(wshape)
            dx = _gemm(dy, R * T, Cout, W2.t().contiguous(), Cin) if ctx.needs_input_grad[0] else None
        else:
            dW2, db = FC.conv2d_wgrad(dy, x, R, 1, T, Cin, Cout, k, 1, stride, p, with_bias=has_b, dil=dil)
            dw = dW2.view(Cout, k, k, Cin)[:, k // 2].permute(0, 2, 1).contiguous()
            dx = None
            if ctx.needs_input_grad[0]:
                Wd = W2.view(Cout, k * k, Cin).permute(2, 1, 0).reshape(Cin, k * k * Cout).contiguous()
                dx = FC.conv2d_dx(dy, R, 1, T, Cin, Wd, Cout, k, 1, stride, p, dil=dil)
        return dx, None, dw, (db if has_b else None)


def _seg_counts(T, seg_len, device):
    nseg = -(-T // seg_len)
    cnt = torch.full((nseg,), float(seg_len), device=device, dtype=torch.float32)
    cnt[-1] = T - (nseg - 1) * seg_len
    return cnt


class SegContextFn(torch.autograd.Function):
    """x [R*T, C] -> context [R*nseg, C] = mean over the utterance + mean over each `seg_len`-frame segment (CAMLayer:
    x.mean(-1) + seg_pooling(x); one row per segment -- the reference expands it over the frames before its 1x1 convs)."""

    @staticmethod
    def forward(ctx, x, geo):
        _need_cuda(x, "CAM++")
        R, T, seg_len = geo
        x = x.contiguous()
        Cc = x.shape[1]
        nseg = -(-T // seg_len)
        sums = _empty(x.device, R, nseg, Cc)
        dev.seg_sums(x, None, R, T, Cc, seg_len, sums)
        cnt = _seg_counts(T, seg_len, x.device)
        ctx.geo = (R, T, seg_len, Cc, nseg)
        ctx.save_for_backward(cnt)
        return (sums / cnt[None, :, None] + sums.sum(1, keepdim=True) / T).view(R * nseg, Cc)     # [R, nseg, C]: tiny

    @staticmethod
    def backward(ctx, dctx):
        (cnt,) = ctx.saved_tensors
        R, T, seg_len, Cc, nseg = ctx.geo
        dctx = dctx.view(R, nseg, Cc)
        dsums = (dctx / cnt[None, :, None] + dctx.sum(1, keepdim=True) / T).contiguous()
        dx = _empty(dctx.device, R * T, Cc)
        dev.seg_scale(None, dsums, R, T, Cc, seg_len, dx)
        return dx, None


class SegGateFn(torch.autograd.Function):
    """y [R*T, C] * m [R*nseg, C] broadcast over the frames of each segment (the context-aware mask product)."""

    @staticmethod
    def forward(ctx, y, m, geo):
        _need_cuda(y, "CAM++")
        R, T, seg_len = geo
        y, m = y.contiguous(), m.contiguous()
        Cc = y.shape[1]
        out = torch.empty_like(y)
        dev.seg_scale(y, m, R, T, Cc, seg_len, out)
        ctx.save_for_backward(y, m)
        ctx.geo = (R, T, seg_len, Cc)
        return out

    @staticmethod
    def backward(ctx, dout):
        y, m = ctx.saved_tensors
        R, T, seg_len, Cc = ctx.geo
        dout = dout.contiguous()
        dy = torch.empty_like(y)
        dev.seg_scale(dout, m, R, T, Cc, seg_len, dy)
        dm = torch.empty_like(m)
        dev.seg_sums(dout, y, R, T, Cc, seg_len, dm)
        return dy, dm, None
