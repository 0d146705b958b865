"""GPU parity of the DPCCN path: every new conv2d.hip kernel against its torch statement
(tests/emu_dev.py, the same functions that stand in for the device in the CPU host-logic test), and the assembled
model against the fixtures generated from the real reference (waveform <= 1e-3, loss <= 1e-2 dB, gradient norms)."""
import os

import numpy as np
import pytest
import torch

from tests import emu_dev as E
from tests.gradcheck import compare_grads
from tests.test_resnet_gpu import _record_relu_masks

pytestmark = pytest.mark.gpu
GRAD_TOL = 5e-3     # per parameter tensor, relative L2 against the oracle's autograd


def _cuda():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def test_dpccn_kernels_match_torch():
    from wesep_amd import dev
    d = _cuda()
    g = torch.Generator().manual_seed(0)
    B, H, W, C = 2, 9, 14, 8
    x = torch.randn(B * H * W, C, generator=g)
    # im2col / col2im with per-axis strides (1, 2), k = 3, p = 1
    Ho, Wo = H, (W + 2 - 3) // 2 + 1
    pr, pg = torch.zeros(B * Ho * Wo, 9 * C), torch.zeros(B * Ho * Wo, 9 * C, device=d)
    E.im2col_hw(x, B, H, W, C, 3, 1, 2, 1, pr, 9 * C)
    dev.im2col_hw(x.to(d), B, H, W, C, 3, 1, 2, 1, pg, 9 * C)
    assert torch.equal(pg.cpu(), pr)
    dp = torch.randn(B * Ho * Wo, 9 * C, generator=g)
    xr, xg = torch.zeros(B * H * W, C), torch.zeros(B * H * W, C, device=d)
    E.col2im_hw(dp, B, H, W, C, 3, 1, 2, 1, xr)
    dev.col2im_hw(dp.to(d), B, H, W, C, 3, 1, 2, 1, xg)
    assert rel(xg, xr) < 1e-6
    # ELU
    yr, yg = torch.empty_like(x), torch.empty_like(x, device=d)
    E.elu_fwd(x, yr)
    dev.elu_fwd(x.to(d), yg)
    assert rel(yg, yr) < 1e-6
    dy = torch.randn_like(x)
    gr, gg = torch.empty_like(x), torch.empty_like(x, device=d)
    E.elu_bwd(x, dy, gr)
    dev.elu_bwd(x.to(d), dy.to(d), gg)
    assert rel(gg, gr) < 1e-6
    # InstanceNorm over the H*W positions of each batch row
    x2 = x * 1.7 + 0.4
    yr, yg = torch.empty_like(x), torch.empty_like(x, device=d)
    sr = E.inorm_fwd(x2, B, H * W, C, yr)
    sg = dev.inorm_fwd(x2.to(d), B, H * W, C, yg)
    assert rel(yg, yr) < 1e-5 and rel(sg, sr) < 1e-5
    gr, gg = torch.empty_like(x), torch.empty_like(x, device=d)
    E.inorm_bwd(yr, dy, sr, B, H * W, C, gr)
    dev.inorm_bwd(yg, dy.to(d), sg, B, H * W, C, gg)
    assert rel(gg, gr) < 1e-4
    # AvgPool2d(2) and bilinear upsampling back
    ar, ag = torch.zeros(B * (H // 2) * (W // 2), C), torch.zeros(B * (H // 2) * (W // 2), C, device=d)
    E.avgpool_fwd(x, B, H, W, C, 2, ar)
    dev.avgpool_fwd(x.to(d), B, H, W, C, 2, ag)
    assert rel(ag, ar) < 1e-6
    gr, gg = torch.zeros(B * H * W, C), torch.zeros(B * H * W, C, device=d)
    E.avgpool_bwd(ar, B, H, W, C, 2, gr)
    dev.avgpool_bwd(ag, B, H, W, C, 2, gg)
    assert rel(gg, gr) < 1e-6
    ur, ug = torch.zeros(B * H * W, C), torch.zeros(B * H * W, C, device=d)
    E.bilinear_fwd(ar, B, H // 2, W // 2, H, W, C, ur)
    dev.bilinear_fwd(ag, B, H // 2, W // 2, H, W, C, ug)
    assert rel(ug, ur) < 1e-5
    br, bg = torch.zeros_like(ar), torch.zeros_like(ag)
    E.bilinear_bwd(dy, B, H // 2, W // 2, H, W, C, br)
    dev.bilinear_bwd(dy.to(d), B, H // 2, W // 2, H, W, C, bg)
    assert rel(bg, br) < 1e-5
    # speaker-fusion scale / shift per (row, bin)
    s = torch.randn(B, W, generator=g)
    for mode in (0, 1):
        yr, yg = torch.empty_like(x), torch.empty_like(x, device=d)
        E.scale_bf_fwd(x, s, B, H, W, C, mode, yr)
        dev.scale_bf_fwd(x.to(d), s.to(d), B, H, W, C, mode, yg)
        assert rel(yg, yr) < 1e-6
        dxr, dsr = torch.empty_like(x), torch.empty_like(s)
        dxg, dsg = torch.empty_like(x, device=d), torch.empty_like(s, device=d)
        E.scale_bf_bwd(x, dy, s, B, H, W, C, mode, dxr, dsr)
        dev.scale_bf_bwd(x.to(d), dy.to(d), s.to(d), B, H, W, C, mode, dxg, dsg)
        assert rel(dxg, dxr) < 1e-6 and rel(dsg, dsr) < 1e-5


@pytest.mark.parametrize("Cin,Cout,k,sh,sw", [(16, 16, 3, 1, 2), (32, 16, 3, 1, 1), (80, 16, 3, 1, 1), (4, 16, 3, 1, 1),
                                              (16, 32, 3, 2, 2), (12, 8, 5, 1, 2), (96, 16, 3, 1, 2), (176, 8, 3, 1, 1)])
def test_implicit_conv2d_and_transpose_match_torch(Cin, Cout, k, sh, sw):
    """Conv2d / ConvTranspose2d as GEMMs on the implicit patch matrix (ws_conv_view, both views, NT and TN kernels)
    against torch's convolutions: output, input gradient, weight and bias gradients, at a size with many row tiles,
    partial tiles and borders on every side.  Run twice: bit-identical."""
    from wesep_amd import functional_dpccn as FD
    d = _cuda()
    g = torch.Generator().manual_seed(100 * Cin + Cout + k + sh + sw)
    B, H, W = 3, 13, 37
    p = k // 2

    def cl(t):      # [B, C, H, W] -> channels-last rows
        return t.permute(0, 2, 3, 1).reshape(-1, t.shape[1]).contiguous()

    for transpose in (False, True):
        x = torch.randn(B, Cin, H, W, generator=g)
        if transpose:
            w = torch.randn(Cin, Cout, k, k, generator=g) * 0.1
            ref_fn = lambda x_, w_, b_: torch.nn.functional.conv_transpose2d(x_, w_, b_, stride=(sh, sw), padding=p)
            Fn = FD.ConvTranspose2dFn
        else:
            w = torch.randn(Cout, Cin, k, k, generator=g) * 0.1
            ref_fn = lambda x_, w_, b_: torch.nn.functional.conv2d(x_, w_, b_, stride=(sh, sw), padding=p)
            Fn = FD.Conv2dFn
        b = torch.randn(Cout, generator=g)
        xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
        yr = ref_fn(xr, wr, br)
        dy = torch.randn(*yr.shape, generator=g)
        yr.backward(dy.double())
        outs = []
        for _ in range(2):
            xg = cl(x).to(d).requires_grad_(True)
            wg, bg = w.to(d).requires_grad_(True), b.to(d).requires_grad_(True)
            yg = Fn.apply(xg, wg, bg, (B, H, W, sh, sw))
            yg.backward(cl(dy).to(d))
            outs.append((yg.detach(), xg.grad, wg.grad, bg.grad))
        for a, c in zip(*outs):
            assert torch.equal(a, c)
        yg, dxg, dwg, dbg = outs[0]
        assert rel(yg, cl(yr)) < 4e-5, ("fwd", transpose)
        assert rel(dxg, cl(xr.grad)) < 4e-5, ("dx", transpose)
        assert rel(dwg, wr.grad) < 4e-5, ("dw", transpose)
        assert rel(dbg, br.grad) < 4e-5, ("db", transpose)


@pytest.mark.parametrize("name", ["dpccn_multiply_r2_t4480", "dpccn_additive_xform_r2_t4608", "dpccn_film_r2_t4352",
                                  "dpccn_concat_xform_r2_t4352", "dpccn_causal_r2_t4480"])
def test_dpccn_model_matches_reference_fixture(name, golden_dir):
    from oracle import bsrnn_oracle as O
    from oracle import dpccn_oracle as DP
    from oracle.make_golden import DPCCN_CASES
    from wesep_amd.models import get_model
    from wesep_amd.utils.losses import parse_loss
    d = _cuda()
    kw, R, T, seed = DPCCN_CASES[name]
    cfg = DP.DPCCNConfig(**kw)
    params = DP.synth_params(cfg, seed)
    model = get_model("DPCCN")(**kw, joint_training=False)
    model.load_state_dict(params, strict=True)
    model = model.to(d).train()
    wav, tgt, emb = O.synth_batch(R, T, seed)
    est, dummy = model(wav.to(d), emb.to(d))
    assert dummy.dim() == 0
    loss = parse_loss("SISDR")[0](est, tgt.to(d))
    loss.backward()
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    assert rel(est, torch.from_numpy(g["est"])) < 1e-3
    assert abs(loss.item() - float(g["loss"])) < 1e-2
    # every parameter gradient, per tensor, against the oracle's autograd (the oracle itself is pinned to the
    # reference's gradient norms by tests/test_oracle_golden.py) -- and the reference's own norms from the fixture
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    O.sisdr_loss(DP.dpccn_forward(p, cfg, wav, emb), tgt).backward()
    worst, wname, bad = compare_grads(((k, prm.grad) for k, prm in model.named_parameters()),
                                      {k: v.grad for k, v in p.items()}, GRAD_TOL)
    print(f"{name}: est rel {rel(est, torch.from_numpy(g['est'])):.2e}, worst gradient rel-L2 {worst:.2e} ({wname})")
    assert not bad, bad[:8]
    top = max(float(g["gnorm/" + k]) for k, _ in model.named_parameters())
    for k, prm in model.named_parameters():
        gn = float(g["gnorm/" + k])
        if gn > 1e-6 * top:
            assert abs(float(prm.grad.norm()) - gn) <= GRAD_TOL * gn, (k, float(prm.grad.norm()), gn)


def test_dpccn_unbuilt_variants_fail_loudly():
    from wesep_amd.models import get_model
    for kw in (dict(joint_training=False, spk_fuse_type="nope"), dict(joint_training=False, stride2=(1, 1))):
        with pytest.raises(NotImplementedError):
            get_model("DPCCN")(**kw)
    m = get_model("DPCCN")(joint_training=False, tcn_blocks=1, tcn_layers=1)
    with pytest.raises(Exception):
        m(torch.randn(2, 4480), torch.randn(2, 256))            # CPU tensors: no CPU path


def test_baseline_config3_dpccn_with_joint_resnet34_vs_oracle_chain(monkeypatch):
    """BASELINE.json configs[2] -- pDPCCN + jointly-learned speaker encoder -- with the recipe's arguments
    (examples/librimix/tse/v2/confs/dpccn.yaml: multiply fusion, ResNet34 on 80-d fbank, 256-d embedding) at 4 rows x
    4 s, 398 enrollment frames: forward against oracle(ResNet restatement) -> oracle(DPCCN), loss, and gradient norms of
    the separator AND of the speaker encoder (the chain is differentiated end to end on the CPU): every parameter
    gradient per tensor (relative L2) against the oracle chain's autograd."""
    from oracle import bsrnn_oracle as O
    from oracle import dpccn_oracle as DP
    from oracle import resnet_oracle as RO
    from wesep_amd.models import get_model
    from wesep_amd.utils.losses import parse_loss
    d = _cuda()
    kw = dict(spk_fuse_type="multiply", use_spk_transform=False)
    cfg = DP.DPCCNConfig(**kw)
    sep = DP.synth_params(cfg, 21)
    spk = RO.synth_params(22, prefix="spk_model.")
    model = get_model("DPCCN")(**kw, joint_training=True, spk_model="ResNet34", spk_feat=True,
                               spk_args=dict(feat_dim=80, embed_dim=256, pooling_func="TSTP", two_emb_layer=False))
    model.load_state_dict({**spk, **sep}, strict=True)
    model = model.to(d).train()
    R, T = 4, 64000
    wav, tgt, _ = O.synth_batch(R, T, 21)
    fbank = torch.randn(R, 398, 80, generator=torch.Generator().manual_seed(23))
    fbank = fbank - fbank.mean(1, keepdim=True)
    masks = _record_relu_masks(monkeypatch)
    est, second = model(wav.to(d), fbank.to(d))
    loss = parse_loss("SISDR")[0](est, tgt.to(d))
    loss.backward()
    torch.cuda.synchronize()
    ps = {k: v.clone().requires_grad_(True) for k, v in sep.items()}
    pk = {k: (v.clone() if RO.is_buffer(k) else v.clone().requires_grad_(True)) for k, v in spk.items()}
    # (the encoder's ReLUs take the device's masks: the chain is differentiated on the SAME linear region, so that
    #  what is compared is arithmetic and not the few pre-activations within rounding distance of a kink)
    emb = RO.resnet_forward(pk, fbank, prefix="spk_model.", relu_masks=masks)
    ref = DP.dpccn_forward(ps, cfg, wav, emb)
    loss_o = O.sisdr_loss(ref, tgt)
    loss_o.backward()
    assert tuple(second.shape) == (R, 256) and rel(second, emb) < 1e-3
    assert rel(est, ref) < 1e-3, rel(est, ref)
    assert abs(loss.item() - loss_o.item()) < 1e-2
    want = {**{k: v.grad for k, v in ps.items()}, **{k: v.grad for k, v in pk.items() if not RO.is_buffer(k)}}
    worst, wname, bad = compare_grads(((k, prm.grad) for k, prm in model.named_parameters()), want, GRAD_TOL)
    print(f"config 3 (DPCCN + joint ResNet34, R=4 x 4 s): est rel {rel(est, ref):.2e}, "
          f"dloss {abs(loss.item() - loss_o.item()):.2e} dB, {len(want)} gradients, worst rel-L2 {worst:.2e} ({wname})")
    assert not bad, bad[:8]


@pytest.mark.parametrize("h,w,H,W", [(7, 8, 250, 257), (15, 16, 250, 257), (3, 5, 13, 11), (1, 1, 9, 7), (62, 64, 250, 257)])
def test_bilinear_adjoint_at_the_pooling_branch_scales(h, w, H, W):
    """The separable two-pass adjoint of nn.Upsample(size, mode='bilinear') (DPCCN's pooled branches upsample by 4 .. 32,
    dpccn.py:257-265) against autograd of F.interpolate, including sizes that are not multiples of each other."""
    from wesep_amd import dev
    d = _cuda()
    B, C = 2, 8
    g = torch.Generator().manual_seed(h * 1000 + W)
    dy = torch.randn(B * H * W, C, generator=g)
    x = torch.zeros(B, C, h, w, dtype=torch.float64, requires_grad=True)
    y = torch.nn.functional.interpolate(x, size=(H, W), mode="bilinear", align_corners=False)
    y.backward(dy.double().view(B, H, W, C).permute(0, 3, 1, 2))
    want = x.grad.permute(0, 2, 3, 1).reshape(B * h * w, C)
    outs = []
    for _ in range(2):
        got = torch.full((B * h * w, C), float("nan"), device=d)
        dev.bilinear_bwd(dy.to(d), B, h, w, H, W, C, got)
        outs.append(got)
    assert torch.equal(outs[0], outs[1])
    err = float((outs[0].double().cpu() - want).norm() / want.norm())
    assert err < 5e-6, err                      # fp32 source coordinates at non-integer scales (2.2e-6 at 250 / 62)


@pytest.mark.parametrize("order", ["pre", "post"])
@pytest.mark.parametrize("G,P,C", [(3, 1000, 16), (2, 517, 32), (4, 33, 384), (1, 4099, 8)])
def test_fused_elu_instancenorm_vs_torch(order, G, P, C):
    """dev.in_act_fwd / in_act_bwd (conv2d.hip ws_in_act_*): IN(ELU(x)) and ELU(IN(x)) against torch autograd in fp64,
    reproducible, and the same values as the two-kernel composition they replace."""
    from wesep_amd import dev
    from wesep_amd import functional_dpccn as FD
    d = _cuda()
    g = torch.Generator().manual_seed(G * 100 + C)
    x = (torch.randn(G * P, C, generator=g) * 1.5 + 0.2).to(d)
    dy = torch.randn(G * P, C, generator=g).to(d)
    xr = x.double().cpu().view(G, P, C).requires_grad_(True)
    inorm = lambda t: (t - t.mean(1, keepdim=True)) / torch.sqrt(t.var(1, unbiased=False, keepdim=True) + 1e-5)
    ref = inorm(torch.nn.functional.elu(xr)) if order == "pre" else torch.nn.functional.elu(inorm(xr))
    ref.backward(dy.double().cpu().view(G, P, C))
    flags = dev.IN_ELU_PRE if order == "pre" else dev.IN_ELU_POST
    outs = []
    for _ in range(2):
        y = torch.full((G * P, C), float("nan"), device=d)
        st = dev.in_act_fwd(x, G, P, C, flags, y)
        dx = torch.full((G * P, C), float("nan"), device=d)
        dev.in_act_bwd(x, dy, st, G, P, C, flags, dx)
        outs.append((y, dx))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    y, dx = outs[0]
    assert rel(y, ref.detach().view(G * P, C)) < 2e-5
    assert rel(dx, xr.grad.view(G * P, C)) < 2e-4
    # the composition it replaces
    x2 = x.clone().requires_grad_(True)
    y2 = FD.InstNormFn.apply(FD.EluFn.apply(x2), (G, P)) if order == "pre" else FD.EluFn.apply(FD.InstNormFn.apply(x2, (G, P)))
    y2.backward(dy)
    assert rel(y, y2.detach()) < 1e-6 and rel(dx, x2.grad) < 1e-5


@pytest.mark.parametrize("B,H,W,Cin,Cout,ldx,ldy,res", [(2, 9, 257, 16, 16, 80, 16, False), (1, 5, 130, 80, 16, 80, 16, False),
                                                        (2, 4, 65, 32, 160, 32, 160, True), (1, 3, 7, 20, 36, 24, 40, True),
                                                        (3, 2, 129, 160, 32, 160, 32, False), (1, 1, 1, 4, 4, 4, 4, False),
                                                        (1, 70, 37, 48, 64, 48, 64, True), (2, 33, 101, 16, 80, 16, 80, True)])
def test_conv3x3_halo_kernel_vs_torch(B, H, W, Cin, Cout, ldx, ldy, res):
    """ws_conv3x3 (conv3x3.hip): 3 x 3 / stride 1 / padding 1 on a channels-last image with pixel stride ldx into rows of
    stride ldy, with bias and the in-place residual (the input-gradient accumulation of DenseBlockFn), against
    F.conv2d in fp64; tiles (32 rows x 8 or 16 columns) that end inside the image, several row tiles, short channel
    chunks, several 64-channel output groups, columns beyond Cout untouched."""
    from wesep_amd import dev
    d = _cuda()
    g = torch.Generator().manual_seed(Cin * 7 + Cout)
    M = B * H * W
    X = torch.randn(M, ldx, generator=g)
    Wt = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.1
    bias = torch.randn(Cout, generator=g)
    Y0 = torch.randn(M, ldy, generator=g)
    W2 = Wt.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous()
    ref = torch.nn.functional.conv2d(X[:, :Cin].double().view(B, H, W, Cin).permute(0, 3, 1, 2), Wt.double(), bias.double(),
                                     padding=1).permute(0, 2, 3, 1).reshape(M, Cout)
    if res:
        ref = ref + Y0[:, :Cout].double()
    outs = []
    for _ in range(2):
        Y = Y0.clone().to(d)
        dev.conv3x3(X=X.to(d), ldx=ldx, W=dev.conv3x3_pack(W2.to(d), Cin, Cout), ldw=9 * Cin, B=B, H=H, Wd=W, Cin=Cin, Cout=Cout, Y=Y, ldy=ldy,
                    bias=bias.to(d), R=Y if res else None)
        outs.append(Y)
    assert torch.equal(outs[0], outs[1])
    assert rel(outs[0][:, :Cout], ref) < 2e-5
    assert torch.equal(outs[0][:, Cout:].cpu(), Y0[:, Cout:])


@pytest.mark.parametrize("B,H,W,Cin,Nn,ldx,sw", [(2, 9, 257, 16, 16, 80, 1), (1, 61, 9, 80, 32, 80, 1), (2, 30, 4, 32, 16, 32, 1),
                                                 (1, 31, 5, 20, 36, 24, 1), (3, 7, 33, 96, 64, 96, 1), (1, 1, 1, 4, 4, 4, 1),
                                                 (2, 9, 257, 16, 32, 16, 2), (1, 33, 64, 32, 64, 40, 2), (2, 5, 17, 64, 128, 64, 2),
                                                 (1, 2, 1, 4, 8, 4, 2)])
def test_conv3x3_halo_weight_gradient_vs_torch(B, H, W, Cin, Nn, ldx, sw):
    """ws_conv3x3_wgrad (conv3x3.hip): dW and db of a 3 x 3 / padding 1 convolution with stride (1, sw) against
    torch.nn.grad.conv2d_weight in fp64: tiles of 30 rows x 4 columns that end inside the grid, even and odd image
    widths under the stride, several input-channel chunks and output-channel tiles, an image that is the prefix of
    wider rows, run-to-run identity."""
    from wesep_amd import functional_conv as FC
    d = _cuda()
    g = torch.Generator().manual_seed(Cin * 5 + Nn + sw)
    Wo = (W - 1) // sw + 1
    X = torch.randn(B * H * W, ldx, generator=g)
    G = torch.randn(B * H * Wo, Nn, generator=g)
    img = X[:, :Cin].double().view(B, H, W, Cin).permute(0, 3, 1, 2)
    gg = G.double().view(B, H, Wo, Nn).permute(0, 3, 1, 2)
    ref = torch.nn.grad.conv2d_weight(img, (Nn, Cin, 3, 3), gg, stride=(1, sw), padding=1).permute(0, 2, 3, 1).reshape(Nn, 9 * Cin)
    outs = [FC.halo_wgrad(G.to(d), Nn, X.to(d), ldx, B, H, Wo, Cin, True, sw, W) for _ in range(2)]
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert rel(outs[0][0], ref) < 2e-5
    assert rel(outs[0][1], gg.sum((0, 2, 3))) < 1e-5


def test_fused_elu_instancenorm_on_a_column_range():
    """dev.in_act_fwd writing into / dev.in_act_bwd reading from columns of a wider tensor (the dense blocks' feature map)
    give the bits of the dense call, and leave the other columns alone."""
    from wesep_amd import dev
    d = _cuda()
    g = torch.Generator().manual_seed(9)
    G, P, C, ld, off = 3, 150, 16, 48, 20
    x = torch.randn(G * P, C, generator=g).to(d)
    dy = torch.randn(G * P, C, generator=g).to(d)
    wide0 = torch.randn(G * P, ld, generator=g).to(d)
    y = torch.empty(G * P, C, device=d)
    st = dev.in_act_fwd(x, G, P, C, dev.IN_ELU_PRE, y)
    wide = wide0.clone()
    st2 = dev.in_act_fwd(x, G, P, C, dev.IN_ELU_PRE, wide, y_ld=ld, y_off=off)
    assert torch.equal(st, st2) and torch.equal(wide[:, off:off + C], y)
    assert torch.equal(wide[:, :off], wide0[:, :off]) and torch.equal(wide[:, off + C:], wide0[:, off + C:])
    dx = torch.empty_like(x)
    dev.in_act_bwd(x, dy, st, G, P, C, dev.IN_ELU_PRE, dx)
    wide[:, off:off + C] = dy
    dx2 = torch.empty_like(x)
    dev.in_act_bwd(x, wide, st, G, P, C, dev.IN_ELU_PRE, dx2, dy_ld=ld, dy_off=off)
    assert torch.equal(dx, dx2)
    with pytest.raises(dev.L.WesepHipError):
        dev.in_act_fwd(x, G, P, C, dev.IN_ELU_PRE, wide, y_ld=ld, y_off=ld - 8)


@pytest.mark.parametrize("C0,g,Co5", [(16, 16, 16), (4, 16, 32), (32, 16, 64)])
def test_conv3x3_weight_pack_kernel_equals_the_composed_pack(C0, g, Co5):
    """ws_conv3x3_pack (ABI v19): the packed weights of ws_conv3x3 in one launch from strided views of the weight tensors -- bit
    for bit what dev.conv3x3_pack composes from torch ops, for a dense block's five forward layers ([co][ci][3][3] read in
    place) and for the input-gradient packs of its channel blocks (rows = a block's input channels, columns = the later layers'
    output channels side by side, taps flipped: convs.py:80-112)."""
    from wesep_amd import dev
    d = _cuda()
    gen = torch.Generator().manual_seed(C0 + Co5)
    ws = [torch.randn(g if i < 4 else Co5, C0 + i * g, 3, 3, generator=gen).to(d) for i in range(5)]
    for i, w in enumerate(ws):
        Co, Ci = w.shape[0], w.shape[1]
        want = dev.conv3x3_pack(w.permute(0, 2, 3, 1).reshape(Co, 9 * Ci), Ci, Co)
        got = dev.conv3x3_pack_srcs([(w, 0, 9 * Ci, 9, 1, 0, Ci)], Ci, Co)
        assert torch.equal(got.view(torch.int32), want.view(torch.int32)), i
    wflip = [w.flip(2, 3).permute(1, 2, 3, 0) for w in ws]
    Dtot = 4 * g + Co5
    for i in range(5):
        lo, hi = (0, C0) if i == 0 else (C0 + (i - 1) * g, C0 + i * g)
        cin = Dtot - i * g
        Wb = torch.cat([wflip[k][lo:hi] for k in range(i, 5)], 3).reshape(hi - lo, 9 * cin)
        want = dev.conv3x3_pack(Wb, cin, hi - lo)
        srcs, off = [], 0
        for k in range(i, 5):
            srcs.append((ws[k], lo * 9, 9, 9 * (C0 + k * g), 1, off, ws[k].shape[0]))
            off += ws[k].shape[0]
        got = dev.conv3x3_pack_srcs(srcs, cin, hi - lo, flip=True)
        assert torch.equal(got.view(torch.int32), want.view(torch.int32)), ("dx", i)
