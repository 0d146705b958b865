"""Autograd shims of the DPCCN path over the C ABI.

Channels-last grids: a reference tensor [B, C, T, F] lives here as [B*T*F, C] (H = frame t, W = frequency bin f).
  * Conv2d            = im2col (per-axis strides) + one split-bf16 MFMA GEMM (K = 9*Cin) with the bias in the epilogue
  * ConvTranspose2d   = GEMM to [rows, 9*Cout] + the col2im gather (deterministic), i.e. the adjoint pair of the above
  * ELU, InstanceNorm (no affine, statistics per batch row and channel), AvgPool2d, bilinear upsampling, the speaker
    fusion scale: csrc/conv2d.hip
  * TCN block: InstanceNorm1d - ELU - depthwise dilated conv (the Conv-TasNet kernel with an identity norm) - IN - ELU -
    1x1 conv (+ residual)
  * STFT / iSTFT (hann, centred, reflect): framing kernel + windowed (inverse) DFT as exact-fp32 MFMA GEMMs on
    hop-strided row views + overlap-add with the window envelope divided out
Reference lines: wesep/models/dpccn.py:206-290, wesep/modules/dpccn/convs.py:28-152, speaker.py:102-121."""
import math

import torch

from . import dev
from . import functional_conv as FC
from .dev import Rows, flat
from .functional import _empty, _need_cuda
from .functional_tasnet import _gemm, _transposed, _wgrad


def _pad4(n):
    return -(-n // 4) * 4


# ---------------------------------------------------------------------------------------------
# convolutions
# ---------------------------------------------------------------------------------------------
class Conv2dFn(torch.autograd.Function):
    """x [B*H*W, Cin] -> conv2d(k x k, stride (sh, sw), padding k//2) + bias: [B*Ho*Wo, Cout].  One GEMM per pass on
    the implicit patch matrix (functional_conv); 1x1 / stride-1 convolutions are plain GEMMs on the rows."""

    @staticmethod
    def forward(ctx, x, w, b, geo):
        _need_cuda(x, "DPCCN")
        B, H, W, sh, sw = geo
        Cout, Cin, k, _ = w.shape
        p = k // 2
        x = x.contiguous()
        if not FC.implicit_ok(Cin):
            raise dev.L.WesepHipError(f"DPCCN Conv2d: input channels must be a multiple of 4 (got {Cin})")
        W2, Wd = FC.conv2d_weights(w)
        if k == 1 and sh == 1 and sw == 1:
            y = _gemm(x, B * H * W, Cin, W2, Cout, bias=b)
        else:
            y = FC.conv2d_fwd(x, B, H, W, Cin, W2, Cout, k, sh, sw, p, bias=b)
        ctx.save_for_backward(x, Wd)
        ctx.geo = (B, H, W, Cin, Cout, k, sh, sw, p, w.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, Wd = ctx.saved_tensors
        B, H, W, Cin, Cout, k, sh, sw, p, wshape = ctx.geo
        dy = dy.contiguous()
        plain = k == 1 and sh == 1 and sw == 1
        if plain:
            dW2, db = _wgrad(dy, B * H * W, Cout, x, Cin)
        else:
            dW2, db = FC.conv2d_wgrad(dy, x, B, H, W, Cin, Cout, k, sh, sw, p)
        dw = dW2.reshape(Cout, k, k, Cin).permute(0, 3, 1, 2).contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            if plain:
                dx = _gemm(dy, B * H * W, Cout, Wd, Cin)
            elif FC.implicit_ok(Cout):
                dx = FC.conv2d_dx(dy, B, H, W, Cin, Wd, Cout, k, sh, sw, p)
            else:  # output channels not a multiple of 4: explicit patch gradients + the col2im gather
                W2 = Wd.view(Cin, k * k, Cout).permute(2, 1, 0).reshape(Cout, k * k * Cin).contiguous()
                M = dy.shape[0]
                dpatches = _gemm(dy, M, Cout, _transposed(W2, Cout, k * k * Cin), k * k * Cin)
                dx = _empty(x.device, B * H * W, Cin)
                dev.col2im_hw(dpatches, B, H, W, Cin, k, sh, sw, p, dx)
        return dx, dw.view(wshape), db, None


class ConvTranspose2dFn(torch.autograd.Function):
    """x [B*H*W, Cin] -> conv_transpose2d(w [Cin, Cout, k, k], stride (sh, sw), padding k//2) + bias:
    [B*Ht*Wt, Cout], Ht = (H - 1) * sh - 2p + k.  Forward = the transposed view of x, input gradient = the convolution
    view of dy, weight gradient = x^T view0(dy): one GEMM each, nothing unfolded."""

    @staticmethod
    def forward(ctx, x, w, b, geo):
        _need_cuda(x, "DPCCN")
        B, H, W, sh, sw = geo
        Cin, Cout, k, _ = w.shape
        p = k // 2
        if not FC.implicit_ok(Cin) or sh > 2 or sw > 2:
            raise dev.L.WesepHipError(f"DPCCN ConvTranspose2d: Cin % 4 == 0 and strides <= 2 (got {Cin}, {sh}, {sw})")
        x = x.contiguous()
        Wt = w.permute(1, 2, 3, 0).reshape(Cout, k * k * Cin).contiguous()      # [co][(tap)*Cin + ci]
        y = FC.convT2d_fwd(x, B, H, W, Cin, Wt, Cout, k, sh, sw, p, bias=b)
        ctx.save_for_backward(x, w)
        ctx.geo = (B, H, W, Cin, Cout, k, sh, sw, p)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        B, H, W, Cin, Cout, k, sh, sw, p = ctx.geo
        d = x.device
        Ht, Wt_ = (H - 1) * sh - 2 * p + k, (W - 1) * sw - 2 * p + k
        Cp = _pad4(Cout)                                        # the implicit operand moves 16-byte channel groups
        if Cp != Cout:
            dyp = torch.zeros(B * Ht * Wt_, Cp, device=d, dtype=torch.float32)
            dyp[:, :Cout] = dy
            dy = dyp
        dy = dy.contiguous()
        db = dev.chan_sums(dy, None, None, 1, B * Ht * Wt_, 1, Cp)[0, 0, :Cout].contiguous()
        Wx = torch.zeros(Cin, k * k, Cp, device=d, dtype=torch.float32)          # [ci][(tap)*Cp + co]
        Wx[:, :, :Cout] = w.permute(0, 2, 3, 1).reshape(Cin, k * k, Cout)
        dWxT = FC.convT2d_wgrad(x, dy, B, H, W, Cin, Cp, k, sh, sw, p)         # [Cin, k*k*Cp]
        dw = dWxT.view(Cin, k, k, Cp)[:, :, :, :Cout].permute(0, 3, 1, 2).contiguous()
        dx = FC.convT2d_dx(dy, B, H, W, Cin, Wx.view(Cin, k * k * Cp), Cp, k, sh, sw, p) \
            if ctx.needs_input_grad[0] else None
        return dx, dw, db, None


class Conv1x1ResFn(torch.autograd.Function):
    """x [M, K] -> x W^T + b (+ res): the TCN's pointwise convolution with the block's residual (convs.py:148-151)."""

    @staticmethod
    def forward(ctx, x, w, b, res):
        _need_cuda(x, "DPCCN")
        x = x.contiguous()
        N, K = w.shape[0], w.shape[1]
        W2 = w.reshape(N, K).contiguous()
        y = _gemm(x, x.shape[0], K, W2, N, bias=b, R=res.contiguous() if res is not None else None)
        ctx.save_for_backward(x, W2)
        ctx.wshape, ctx.has_res = w.shape, res is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W2 = ctx.saved_tensors
        dy = dy.contiguous()
        M, (N, K) = x.shape[0], W2.shape
        dW, db = _wgrad(dy, M, N, x, K)
        dx = _gemm(dy, M, N, _transposed(W2, N, K), K)
        return dx, dW.view(ctx.wshape), db, (dy if ctx.has_res else None)


class DwConvFn(torch.autograd.Function):
    """Depthwise dilated Conv1d along the rows of each batch row ([B*L, C]; convs.py:134-143) + bias."""

    @staticmethod
    def forward(ctx, x, w, b, geo):
        _need_cuda(x, "DPCCN")
        B, Lr, dil = geo[:3]
        causal = bool(geo[3]) if len(geo) > 3 else False
        C, _, P = w.shape
        x = x.contiguous()
        d = x.device
        ident = torch.tensor([[0.0, 1.0]], device=d).repeat(B, 1).contiguous()      # (mean 0, rstd 1) per row
        ones, zeros = torch.ones(C, device=d), torch.zeros(C, device=d)
        wf = w.reshape(C, P).contiguous()
        y = _empty(d, B * Lr, C)
        dev.dwconv_fwd(x, ident, ones, zeros, wf, b, B, Lr, C, P, dil, Lr, y, causal=causal)
        ctx.save_for_backward(x, ident, ones, zeros, wf)
        ctx.geo = (B, Lr, C, P, dil, w.shape, causal)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, ident, ones, zeros, wf = ctx.saved_tensors
        B, Lr, C, P, dil, wshape, causal = ctx.geo
        dx = torch.empty_like(x)
        dw, db = dev.dwconv_bwd(dy.contiguous(), x, ident, ones, zeros, wf, B, Lr, C, P, dil, Lr, dx, causal=causal)
        return dx, dw.reshape(wshape), db, None


# ---------------------------------------------------------------------------------------------
# pointwise / normalisation / resampling
# ---------------------------------------------------------------------------------------------
class EluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _need_cuda(x, "DPCCN")
        x = x.contiguous()
        y = torch.empty_like(x)
        dev.elu_fwd(x, y)
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dx = torch.empty_like(x)
        dev.elu_bwd(x, dy.contiguous(), dx)
        return dx


class InstNormFn(torch.autograd.Function):
    """InstanceNorm{1,2}d without affine: statistics per (batch row, channel) over the row's P positions."""

    @staticmethod
    def forward(ctx, x, geo):
        _need_cuda(x, "DPCCN")
        G, P = geo
        x = x.contiguous()
        y = torch.empty_like(x)
        st = dev.inorm_fwd(x, G, P, x.shape[1], y)
        ctx.save_for_backward(y, st)
        ctx.geo = (G, P)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, st = ctx.saved_tensors
        G, P = ctx.geo
        dx = torch.empty_like(y)
        dev.inorm_bwd(y, dy.contiguous(), st, G, P, y.shape[1], dx)
        return dx, None


class DenseBlockFn(torch.autograd.Function):
    """Five densely connected conv3x3 - ELU - InstanceNorm layers (convs.py:80-112) WITHOUT the concatenations: the
    block's feature map lives in one [M, C0 + 4g] buffer; layer i convolves its first C0 + i*g channels (the halo-tile
    kernel reads a pixel-strided prefix) and its normalised output is written straight into the next g columns.
    Backward: the pre-activation gradients of the five layers live in one [M, 4g + Cout5] buffer, and the gradient of
    each CHANNEL BLOCK of the map (x's C0 channels, then each layer's g) is ONE convolution of the gradients of all
    later layers with the stacked, flipped kernel slices -- written once, where accumulating layer by layer read and
    re-wrote every prefix.  torch.cat copied the growing map once per layer forward and autograd split
    and re-summed it backward.  params = (w1, b1, ..., w5, b5); x [M, C0] -> [M, Cout5]."""

    @staticmethod
    def forward(ctx, x, geo, *params):
        _need_cuda(x, "DPCCN")
        B, H, W = geo
        M = B * H * W
        ws, bs = params[0::2], params[1::2]
        C0, g = x.shape[1], ws[0].shape[0]
        Ctot = C0 + 4 * g
        d = x.device
        for i, w in enumerate(ws):
            if tuple(w.shape[1:]) != (C0 + i * g, 3, 3) or (i < 4 and w.shape[0] != g) or not FC.implicit_ok(w.shape[1]) \
                    or not FC.implicit_ok(w.shape[0]):
                raise dev.L.WesepHipError(f"DPCCN DenseBlock: layer {i + 1} weight {tuple(w.shape)} does not fit the dense "
                                          f"layout (C0 = {C0}, growth {g}, channel counts %% 4)")
        big = _empty(d, M, Ctot)
        big[:, :C0].copy_(x)
        saved, out = [], None
        for i in range(5):
            Ci, Co = C0 + i * g, ws[i].shape[0]
            # the packed [co][(ky*3 + kx)*Ci + ci] weights straight from the [co][ci][3][3] tensor: one launch
            W2 = dev.conv3x3_pack_srcs([(ws[i].contiguous(), 0, 9 * Ci, 9, 1, 0, Ci)], Ci, Co)
            pre = _empty(d, M, Co)
            dev.conv3x3(X=big, ldx=Ctot, W=W2, ldw=9 * Ci, B=B, H=H, Wd=W, Cin=Ci, Cout=Co, Y=pre, ldy=Co, bias=bs[i])
            if i < 4:                    # the layer's output IS the next g columns of the map
                st = dev.in_act_fwd(pre, B, H * W, Co, dev.IN_ELU_PRE, big, y_ld=Ctot, y_off=Ci)
            else:
                out = _empty(d, M, Co)
                st = dev.in_act_fwd(pre, B, H * W, Co, dev.IN_ELU_PRE, out)
            saved += [pre, st]
        ctx.save_for_backward(big, *saved, *ws)
        ctx.geo = (B, H, W, C0, g, tuple(w.shape for w in ws))
        return out

    @staticmethod
    def backward(ctx, dout):
        big = ctx.saved_tensors[0]
        saved, ws = ctx.saved_tensors[1:11], ctx.saved_tensors[11:16]
        B, H, W, C0, g, wshapes = ctx.geo
        M, Ctot, Co5 = B * H * W, C0 + 4 * g, wshapes[4][0]
        Dtot = 4 * g + Co5                  # columns [i*g, ...) of dpre = the pre-activation gradient of layer i
        d = big.device
        dbig = _empty(d, M, Ctot)           # every column block is written exactly once below
        dpre = _empty(d, M, Dtot)
        # input gradient = the correlation of d(pre) with the flipped kernel: [ci][ky][kx][co] = w[co][ci][2 - ky][2 - kx]
        grads = [None] * 10
        for i in range(4, -1, -1):
            pre, st = saved[2 * i:2 * i + 2]
            Ci, Co = C0 + i * g, wshapes[i][0]
            if i < 4:                       # columns [Ci, Ci + g) of dbig: written by the block convolution of iteration i + 1
                dev.in_act_bwd(pre, dbig, st, B, H * W, Co, dev.IN_ELU_PRE, dpre, dy_ld=Ctot, dy_off=Ci, dx_ld=Dtot, dx_off=i * g)
            else:
                dev.in_act_bwd(pre, dout.contiguous(), st, B, H * W, Co, dev.IN_ELU_PRE, dpre, dx_ld=Dtot, dx_off=i * g)
            if FC.halo_wgrad_ok(Ci, Co, 3, 1, 1, 1):
                dW2, db = FC.halo_wgrad(dpre, Co, big, Ctot, B, H, W, Ci, True, ldg=Dtot, g_off=i * g)
            else:
                d_pre = dpre[:, i * g:i * g + Co].contiguous()
                conv = dev.ConvView(0, H, W, Ci, H, W, 3, 1, 1, 1, 1, Ctot)
                if dev.conv_wgrad_ok(Co, conv):
                    dW2, db = FC._one_pass_wgrad(d_pre, M, Co, big, conv, True)
                else:
                    dW2, db = _wgrad(d_pre, M, Co, big, 9 * Ci, with_bias=True, vec=1, mode=FC.MODE, conv=conv)
            grads[2 * i] = dW2.reshape(Co, 3, 3, Ci).permute(0, 3, 1, 2).contiguous().view(wshapes[i])
            grads[2 * i + 1] = db
            # the channel block this layer's INPUT ends with (layer i - 1's output; x for i = 0) is complete once layers
            # i .. 4 have their d(pre): one convolution over the columns [i*g, Dtot) of dpre
            lo, hi = (0, C0) if i == 0 else (C0 + (i - 1) * g, C0 + i * g)
            if i == 0 and not ctx.needs_input_grad[0]:
                break
            cin = Dtot - i * g
            # rows = the block's input channels lo .. hi, columns = the output channels of layers i .. 4 side by side, taps
            # flipped: [c][ky][kx][co] = w_k[co][lo + c][2 - ky][2 - kx] -- one pack launch from the five weight tensors
            srcs, off = [], 0
            for k in range(i, 5):
                Ck, Cok = C0 + k * g, wshapes[k][0]
                srcs.append((ws[k].contiguous(), lo * 9, 9, 9 * Ck, 1, off, Cok))
                off += Cok
            dev.conv3x3(X=dpre, ldx=Dtot, x_off=i * g, W=dev.conv3x3_pack_srcs(srcs, cin, hi - lo, flip=True), ldw=9 * cin,
                        B=B, H=H, Wd=W, Cin=cin, Cout=hi - lo, Y=dbig, ldy=Ctot, y_off=lo)
        dx = dbig[:, :C0].contiguous() if ctx.needs_input_grad[0] else None
        return (dx, None) + tuple(grads)


class EluInstNormFn(torch.autograd.Function):
    """InstanceNorm fused with its neighbouring ELU: order 'pre' = IN(ELU(x)) (conv - ELU - IN, convs.py:28-77), order
    'post' = ELU(IN(x)) (IN - ELU - conv, convs.py:115-152).  Three passes forward, five backward, only x is kept
    (dev.in_act_*); `WESEP_IN_ELU_FUSED=0` composes EluFn and InstNormFn instead."""

    @staticmethod
    def forward(ctx, x, geo, order):
        _need_cuda(x, "DPCCN")
        G, P = geo
        x = x.contiguous()
        y = torch.empty_like(x)
        flags = dev.IN_ELU_PRE if order == "pre" else dev.IN_ELU_POST
        st = dev.in_act_fwd(x, G, P, x.shape[1], flags, y)
        ctx.save_for_backward(x, st)
        ctx.geo = (G, P, flags)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, st = ctx.saved_tensors
        G, P, flags = ctx.geo
        dx = torch.empty_like(x)
        dev.in_act_bwd(x, dy.contiguous(), st, G, P, x.shape[1], flags, dx)
        return dx, None, None


def elu_inorm(x, geo, order):
    import os
    if os.environ.get("WESEP_IN_ELU_FUSED", "1") == "0":
        return InstNormFn.apply(EluFn.apply(x), geo) if order == "pre" else EluFn.apply(InstNormFn.apply(x, geo))
    return EluInstNormFn.apply(x, geo, order)


class AvgPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, geo):
        B, H, W, sz = geo
        x = x.contiguous()
        y = _empty(x.device, B * (H // sz) * (W // sz), x.shape[1])
        dev.avgpool_fwd(x, B, H, W, x.shape[1], sz, y)
        ctx.geo = (B, H, W, sz, x.shape[1])
        return y

    @staticmethod
    def backward(ctx, dy):
        B, H, W, sz, C = ctx.geo
        dx = _empty(dy.device, B * H * W, C)
        dev.avgpool_bwd(dy.contiguous(), B, H, W, C, sz, dx)
        return dx, None


class BilinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, geo):
        B, h, w, H, W = geo
        x = x.contiguous()
        y = _empty(x.device, B * H * W, x.shape[1])
        dev.bilinear_fwd(x, B, h, w, H, W, x.shape[1], y)
        ctx.geo = (B, h, w, H, W, x.shape[1])
        return y

    @staticmethod
    def backward(ctx, dy):
        B, h, w, H, W, C = ctx.geo
        dx = _empty(dy.device, B * h * w, C)
        dev.bilinear_bwd(dy.contiguous(), B, h, w, H, W, C, dx)
        return dx, None


class ScaleBFFn(torch.autograd.Function):
    """x [B*T*F, C] * s[b, f] (mode 0) or + s[b, f] (mode 1)."""

    @staticmethod
    def forward(ctx, x, s, geo):
        B, T, Fq, mode = geo
        x, s = x.contiguous(), s.contiguous()
        y = torch.empty_like(x)
        dev.scale_bf_fwd(x, s, B, T, Fq, x.shape[1], mode, y)
        ctx.save_for_backward(x, s)
        ctx.geo = geo
        return y

    @staticmethod
    def backward(ctx, dy):
        x, s = ctx.saved_tensors
        B, T, Fq, mode = ctx.geo
        dx, ds = torch.empty_like(x), torch.empty_like(s)
        dev.scale_bf_bwd(x, dy.contiguous(), s, B, T, Fq, x.shape[1], mode, dx, ds)
        return dx, ds, None


# ---------------------------------------------------------------------------------------------
# STFT / iSTFT (torch.stft / torch.istft with a hann window, centre = True, reflect padding; dpccn.py:212-220,280-288)
# ---------------------------------------------------------------------------------------------
_TABLES = {}


def _dft_tables(n, device):
    """(analysis basis [2*nf padded, n], synthesis basis [n, 2*nf padded], window) for a periodic hann window."""
    key = (n, device.type, device.index)
    if key not in _TABLES:
        nf = n // 2 + 1
        win = torch.hann_window(n, dtype=torch.float64)
        k = torch.arange(n, dtype=torch.float64)
        ang = 2.0 * math.pi * torch.arange(nf, dtype=torch.float64).unsqueeze(1) * k.unsqueeze(0) / n
        ld = _pad4(2 * nf)
        ana = torch.zeros(ld, n, dtype=torch.float64)
        ana[0:2 * nf:2] = torch.cos(ang) * win
        ana[1:2 * nf:2] = -torch.sin(ang) * win
        # irfft: x[m] = (1/n) * (X0 + (-1)^m X_{n/2} + 2 * sum_{k=1}^{n/2-1} (Re X_k cos - Im X_k sin)); times the window
        ck = torch.full((nf, 1), 2.0, dtype=torch.float64)
        ck[0], ck[-1] = 1.0, 1.0
        syn = torch.zeros(n, ld, dtype=torch.float64)
        syn[:, 0:2 * nf:2] = (ck * torch.cos(ang)).t() / n * win.unsqueeze(1)
        sinp = (ck * torch.sin(ang)).t()
        sinp[:, 0], sinp[:, -1] = 0.0, 0.0                     # imaginary parts of DC / Nyquist are ignored
        syn[:, 1:2 * nf:2] = -sinp / n * win.unsqueeze(1)
        _TABLES[key] = (ana.float().to(device).contiguous(), syn.float().to(device).contiguous(), win.float())
    return _TABLES[key]


def stft_ri(wav, n=512, hop=128):
    """wav [B, T] -> (spec [B*Tf, ld] with columns (re, im) interleaved per bin, Tf); no gradient (mixture input)."""
    _need_cuda(wav, "DPCCN")
    with torch.no_grad():
        wav = wav.float().contiguous()
        B, T = wav.shape
        pad = n // 2
        ana, _, _ = _dft_tables(n, wav.device)
        ldo = _pad4(T + 2 * pad)
        xp = torch.zeros(B, ldo, device=wav.device, dtype=torch.float32)
        dev.preemph_pad(wav, B, T, pad, ldo, 0.0, xp)           # coef 0: plain centred reflect padding
        Tf = 1 + T // hop
        spec = _empty(wav.device, B * Tf, ana.shape[0])
        dev.gemm_nt(A=xp, a_rows=Rows(Tf, ldo, hop), M=B * Tf, N=ana.shape[0], K=n, W=ana, ldw=n, C_out=spec,
                    c_rows=flat(ana.shape[0]), vec=3, mode="f32")
    return spec, Tf


class IstftFn(torch.autograd.Function):
    """spec [B*Tf, ld] (re, im interleaved) -> wav [B, nsample]  (torch.istft, hann, centre, length = nsample)."""

    @staticmethod
    def forward(ctx, spec, geo):
        B, Tf, nsample, n, hop = geo
        d = spec.device
        _, syn, win = _dft_tables(n, d)
        pad = n // 2
        spec = spec.contiguous()
        fr = _gemm(spec, B * Tf, syn.shape[1], syn, n, mode="f32")
        full = pad + nsample
        if full > (Tf - 1) * hop + n:
            raise RuntimeError("iSTFT: requested length exceeds the frames")
        y = _empty(d, B, full)
        dev.ola_fwd(fr, None, B, Tf, n, hop, full, y)
        key = ("env", n, hop, Tf, nsample, d.type, d.index)
        if key not in _TABLES:
            env = torch.zeros((Tf - 1) * hop + n, dtype=torch.float64)
            w2 = win.double() ** 2
            for t in range(Tf):
                env[t * hop: t * hop + n] += w2
            inv = torch.zeros(1, _pad4(nsample))
            inv[0, :nsample] = (1.0 / env[pad: pad + nsample]).float()
            _TABLES[key] = inv.to(d).contiguous()
        inv = _TABLES[key]
        ld = inv.shape[1]
        out = torch.zeros(B, ld, device=d, dtype=torch.float32)
        out[:, :nsample] = y[:, pad:]
        dev.affine_fwd(out, inv, None, 0.0, B, B, ld, out)
        ctx.geo = (B, Tf, nsample, n, hop, ld)
        ctx.save_for_backward(inv, syn)
        return out[:, :nsample].contiguous()

    @staticmethod
    def backward(ctx, dwav):
        inv, syn = ctx.saved_tensors
        B, Tf, nsample, n, hop, ld = ctx.geo
        d = dwav.device
        pad = n // 2
        g = torch.zeros(B, ld, device=d, dtype=torch.float32)
        g[:, :nsample] = dwav
        dev.affine_fwd(g, inv, None, 0.0, B, B, ld, g)
        full = pad + nsample
        dy = torch.zeros(B, full, device=d, dtype=torch.float32)
        dy[:, pad:] = g[:, :nsample]
        dfr = _empty(d, B * Tf, n)
        dev.ola_bwd(dy, B, Tf, n, hop, full, dfr)
        dspec = _gemm(dfr, B * Tf, n, _transposed(syn, n, syn.shape[1]), syn.shape[1], mode="f32")
        return dspec, None
