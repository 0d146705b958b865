"""Autograd shims of the wespeaker ECAPA-TDNN speaker encoder (the reference's
published checkpoint is `bsrnn_ecapa_vox1`, wesep/cli/hub.py:86-95; recipe alternative bsrnn.yaml:66-71) over the C ABI.

Channels-last activations [R*T, C] (row = frame).  A Conv1d is ONE split-bf16 GEMM: k = 1 on the rows themselves, the
dilated k = 3 / 5 convolutions on the implicit patch matrix of the one-row image [R][1][T][C] (`dev.ConvView` with
dilation: the taps of the other kernel rows fall outside the image and are masked, functional_conv).  wespeaker's
blocks are Conv -> ReLU -> BatchNorm: the ReLU rides in the GEMM epilogue, BatchNorm1d is the channels-last
BatchNorm kernel pair of tasnet.hip (identity activation).  Attentive statistics pooling and the row-bias tanh /
sigmoid are csrc/conv2d.hip.  The [R, C]-sized glue of the SE gate is torch arithmetic on a few thousand numbers."""
import torch

from . import _lib as L
from . import dev
from . import functional_conv as FC
from .functional import _empty, _need_cuda, _reduce_new
from .functional_tasnet import _gemm, _wgrad


class Conv1dReluBnFn(torch.autograd.Function):
    """x [R*T, Cin] -> BN(ReLU(conv1d(x, w [Cout, Cin, k], b; dilation, 'same' padding))) [R*T, Cout]."""

    @staticmethod
    def forward(ctx, x, geo, w, b, gamma, beta, rm, rv):
        _need_cuda(x, "ECAPA-TDNN")
        R, T, dil, training = geo
        Cout, Cin, k = w.shape
        M = R * T
        x = x.contiguous()
        d = x.device
        if k == 1:
            W2 = w.reshape(Cout, Cin).contiguous()
            c = _gemm(x, M, Cin, W2, Cout, bias=b, act=2)
        else:
            if Cin % 4 or Cout % 4:
                raise L.WesepHipError(f"ECAPA-TDNN Conv1d: channel counts must be multiples of 4 (got {Cin}, {Cout})")
            # the Conv1d as the middle kernel row of a k x k view of the one-row image; the other rows are masked taps
            W2 = torch.zeros(Cout, k, k, Cin, device=d, dtype=torch.float32)
            W2[:, k // 2] = w.permute(0, 2, 1)
            W2 = W2.view(Cout, k * k * Cin)
            c = FC.conv2d_fwd(x, R, 1, T, Cin, W2, Cout, k, 1, 1, dil * (k // 2), bias=b, dil=dil, act=2)
        st = _empty(d, 2, Cout)
        if training:
            dev.bn_stats(c, M, Cout, rm, rv, st)
        else:
            st[0].copy_(rm)
            st[1].copy_(torch.rsqrt(rv + dev.BN_EPS))
        one = torch.ones(1, device=d, dtype=torch.float32)
        u, y = _empty(d, M, Cout), _empty(d, M, Cout)
        dev.bn_prelu_fwd(c, st, gamma, beta, None, one, M, Cout, u, y)       # slope 1: no activation after the norm
        del u
        ctx.save_for_backward(x, c, st, W2, gamma)
        ctx.geo = (R, T, dil, training, Cin, Cout, k, w.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, c, st, W2, gamma = ctx.saved_tensors
        R, T, dil, training, Cin, Cout, k, wshape = ctx.geo
        M = R * T
        d = x.device
        dc = _empty(d, M, Cout)
        sums = dev.bn_bwd_any(c, dy.contiguous(), st, gamma, M, Cout, dc, training)
        dev.relu_mask(dc, c)                                               # ReLU' from its saved output
        if k == 1:
            dW2, db = _wgrad(dc, M, Cout, x, Cin)
            dw = dW2.view(wshape)
            dx = _gemm(dc, M, Cout, W2.t().contiguous(), Cin) if ctx.needs_input_grad[0] else None
        else:
            p = dil * (k // 2)
            dW2, db = FC.conv2d_wgrad(dc, x, R, 1, T, Cin, Cout, k, 1, 1, p, dil=dil)
            dw = dW2.view(Cout, k, k, Cin)[:, k // 2].permute(0, 2, 1).contiguous()
            dx = None
            if ctx.needs_input_grad[0]:
                Wd = W2.view(Cout, k * k, Cin).permute(2, 1, 0).reshape(Cin, k * k * Cout).contiguous()
                dx = FC.conv2d_dx(dc, R, 1, T, Cin, Wd, Cout, k, 1, 1, p, dil=dil)
        return dx, None, dw, db, sums[1].contiguous(), sums[0].contiguous(), None, None


class TimeMeanFn(torch.autograd.Function):
    """x [R*T, C] -> mean over the T frames of each row [R, C] (SE squeeze)."""

    @staticmethod
    def forward(ctx, x, geo):
        R, T = geo
        x = x.contiguous()
        Cc = x.shape[1]
        ctx.geo = (R, T, Cc)
        return dev.chan_sums(x, None, None, 1, T, R, Cc)[:, 0].contiguous() / T

    @staticmethod
    def backward(ctx, dm):
        R, T, Cc = ctx.geo
        dx = _empty(dm.device, R * T, Cc)
        dev.bcast_rows(dm.contiguous(), 1.0 / T, T, R * T, Cc, dx)
        return dx, None


class RowBiasActFn(torch.autograd.Function):
    """y = act(x + rb[row // rows_per_r]) on [rows, C]; act 1 = tanh, 3 = sigmoid; rb [rows / rows_per_r, C] or None."""

    @staticmethod
    def forward(ctx, x, rb, rows_per_r, act):
        _need_cuda(x, "ECAPA-TDNN")
        x = x.contiguous()
        rows, Cc = x.shape
        y = torch.empty_like(x)
        dev.rowbias_act_fwd(x, rb.contiguous() if rb is not None else None, rows, Cc, rows_per_r, act, y)
        ctx.save_for_backward(y)
        ctx.geo = (rows, Cc, rows_per_r, act, rb is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        rows, Cc, rpr, act, has_rb = ctx.geo
        dx = torch.empty_like(y)
        dev.act_bwd(y, dy.contiguous(), act, dx)
        drb = None
        if has_rb:
            drb = dev.chan_sums(dx, None, None, 1, rpr, rows // rpr, Cc)[:, 0].contiguous()
        return dx, drb, None, None


class AstpFn(torch.autograd.Function):
    """x, logits [R*T, C] -> [R, 2C] = softmax_T(logits)-weighted mean || std."""

    @staticmethod
    def forward(ctx, x, logits, geo):
        _need_cuda(x, "ECAPA-TDNN")
        R, T = geo
        x, logits = x.contiguous(), logits.contiguous()
        Cc = x.shape[1]
        out, aux = _empty(x.device, R, 2 * Cc), _empty(x.device, R, 4 * Cc)
        dev.astp_fwd(x, logits, R, T, Cc, out, aux)
        ctx.save_for_backward(x, logits, out, aux)
        ctx.geo = (R, T, Cc)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, logits, out, aux = ctx.saved_tensors
        R, T, Cc = ctx.geo
        dx, dl = torch.empty_like(x), torch.empty_like(x)
        dev.astp_bwd(x, logits, out, aux, dout.contiguous(), R, T, Cc, dx, dl)
        return dx, dl, None


class BatchNormRowsFn(torch.autograd.Function):
    """BatchNorm1d over the rows of [R, C] (training statistics; running buffers updated in place)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, rm, rv, training):
        _need_cuda(x, "ECAPA-TDNN")
        x = x.contiguous()
        M, Cc = x.shape
        d = x.device
        st = _empty(d, 2, Cc)
        if training:
            dev.bn_stats(x, M, Cc, rm, rv, st)
        else:
            st[0].copy_(rm)
            st[1].copy_(torch.rsqrt(rv + dev.BN_EPS))
        one = torch.ones(1, device=d, dtype=torch.float32)
        u, y = _empty(d, M, Cc), _empty(d, M, Cc)
        dev.bn_prelu_fwd(x, st, gamma, beta, None, one, M, Cc, u, y)
        ctx.save_for_backward(x, st, gamma)
        ctx.training = training
        return y

    @staticmethod
    def backward(ctx, dy):
        x, st, gamma = ctx.saved_tensors
        M, Cc = x.shape
        dx = torch.empty_like(x)
        sums = dev.bn_bwd_any(x, dy.contiguous(), st, gamma, M, Cc, dx, ctx.training)
        return dx, sums[1].contiguous(), sums[0].contiguous(), None, None, None


class LinearReluFn(torch.autograd.Function):
    """relu(x W^T + b) on [M, K] rows (the 1x1 aggregation convolution): ReLU in the GEMM epilogue."""

    @staticmethod
    def forward(ctx, x, W, b):
        _need_cuda(x, "ECAPA-TDNN")
        x, W = x.contiguous(), W.contiguous()
        M, K = x.shape
        N = W.shape[0]
        y = _gemm(x, M, K, W, N, bias=b, act=2)
        ctx.save_for_backward(x, W, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W, y = ctx.saved_tensors
        M, K = x.shape
        N = W.shape[0]
        dy = dy.contiguous().clone()
        dev.relu_mask(dy, y)
        dW, db = _wgrad(dy, M, N, x, K)
        dx = _gemm(dy, M, N, W.t().contiguous(), K) if ctx.needs_input_grad[0] else None
        return dx, dW, db


class GateFn(torch.autograd.Function):
    """y[r, t, c] = x[r, t, c] * g[r, c] (the SE excitation) on [R*T, C]: the gate is broadcast over the frames and
    multiplied with the Conv-TasNet mask-product kernels (any channel count; ws_affine_* stops at 128 columns)."""

    @staticmethod
    def forward(ctx, x, g, geo):
        _need_cuda(x, "ECAPA-TDNN")
        R, T = geo
        x, g = x.contiguous(), g.contiguous()
        M, Cc = x.shape
        gf = _empty(x.device, M, Cc)
        dev.bcast_rows(g, 1.0, T, M, Cc, gf)
        y = _empty(x.device, M, Cc)
        dev.maskmul_fwd(x, 0, Cc, gf, M, Cc, y)
        ctx.save_for_backward(x, g)
        ctx.geo = (R, T, M, Cc)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, g = ctx.saved_tensors
        R, T, M, Cc = ctx.geo
        gf = _empty(x.device, M, Cc)
        dev.bcast_rows(g, 1.0, T, M, Cc, gf)
        dx, dgf = _empty(x.device, M, Cc), _empty(x.device, M, Cc)
        dy = dy.contiguous()
        # two plain products (ws_maskmul_bwd folds a ReLU derivative of its mask operand in: a gate may be negative)
        dev.maskmul_fwd(dy, 0, Cc, gf, M, Cc, dx)
        dev.maskmul_fwd(dy, 0, Cc, x, M, Cc, dgf)
        dg = dev.chan_sums(dgf, None, None, 1, T, R, Cc)[:, 0].contiguous()
        return dx, dg, None


class RowBiasAddFn(torch.autograd.Function):
    """y[r, t, c] = x[r, t, c] + b[r, c] on [R*T, C] (additive speaker fusion / FiLM shift): the PReLU kernel with
    slope 1 and a per-row-group bias."""

    @staticmethod
    def forward(ctx, x, b, geo):
        _need_cuda(x, "speaker fusion")
        R, T = geo
        x, b = x.contiguous(), b.contiguous()
        M, Cc = x.shape
        y = _empty(x.device, M, Cc)
        one = torch.ones(1, device=x.device, dtype=torch.float32)
        dev.prelu_fwd(x, b, one, M, Cc, T, y)
        ctx.geo = (R, T, Cc)
        return y

    @staticmethod
    def backward(ctx, dy):
        R, T, Cc = ctx.geo
        dy = dy.contiguous()
        return dy, dev.chan_sums(dy, None, None, 1, T, R, Cc)[:, 0].contiguous(), None
