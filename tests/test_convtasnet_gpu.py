"""GPU parity of the Conv-TasNet / SpEx+ path: the tasnet.hip kernels against
plain torch fp32 on the same inputs, and the assembled model (autograd Functions -> C ABI -> HIP kernels)
against the CPU oracle and the committed reference fixtures.

Tolerances (BASELINE.json north_star): separated waveforms <= 1e-3 relative L2, SI-SNR loss <= 1e-2 dB;
gradients <= 2e-3 relative L2 per tensor (with an absolute floor: the decoder biases have a mathematically
zero gradient under the DC-invariant SI-SDR loss)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

WAV_TOL = 1e-3
DB_TOL = 1e-2
GRAD_TOL = 2e-3


def _cuda():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


# ---- kernels ---------------------------------------------------------------------------------------
def test_flat_stats_matches_two_pass():
    from wesep_amd import dev
    d = _cuda()
    torch.manual_seed(0)
    R, n = 3, 6399 * 64
    x = (torch.randn(R, n, device=d) * 2 + 5).contiguous()
    st = torch.empty(R, 2, device=d)
    dev.flat_stats(x, R, n, st)
    mean = x.double().mean(1)
    var = x.double().var(1, unbiased=False)
    assert torch.allclose(st[:, 0].double(), mean, rtol=1e-6, atol=1e-6)
    assert torch.allclose(st[:, 1].double(), 1 / torch.sqrt(var + 1e-5), rtol=1e-5)


@pytest.mark.parametrize("with_rb", [False, True])
def test_prelu_fwd_bwd(with_rb):
    from wesep_amd import dev
    d = _cuda()
    torch.manual_seed(1)
    R, Tp, Cc = 3, 37, 40
    M = R * Tp
    x = torch.randn(M, Cc, device=d)
    rb = torch.randn(R, Cc, device=d) if with_rb else None
    a = torch.tensor([0.2], device=d)
    pre_ref = x + (rb.repeat_interleave(Tp, 0) if with_rb else 0)
    xin, y = x.clone(), torch.empty_like(x)
    dev.prelu_fwd(xin, rb, a, M, Cc, Tp, y)
    assert torch.equal(xin, pre_ref) or torch.allclose(xin, pre_ref)
    assert torch.allclose(y, F.prelu(pre_ref, a))
    dy = torch.randn(M, Cc, device=d)
    dx = torch.empty_like(dy)
    da = dev.prelu_bwd(xin, dy, a, dx)
    assert torch.allclose(dx, torch.where(pre_ref > 0, dy, a * dy))
    ref_da = (dy * torch.clamp(pre_ref, max=0)).double().sum()
    assert abs(da.double().item() - ref_da.item()) <= 1e-5 * abs(ref_da.item()) + 1e-4


@pytest.mark.parametrize("norm,dil,P", [("gLN", 1, 3), ("gLN", 4, 3), ("cLN", 2, 5)])
def test_dwconv_fwd_bwd(norm, dil, P):
    from wesep_amd import dev
    from wesep_amd import functional_tasnet as FT
    d = _cuda()
    torch.manual_seed(2)
    R, Tp, Cc = 2, 53, 24
    M = R * Tp
    x = torch.randn(M, Cc, device=d) * 1.5 + 0.3
    gamma, beta = torch.rand(Cc, device=d) + 0.5, torch.randn(Cc, device=d) * 0.1
    w, b = torch.randn(Cc, P, device=d) * 0.5, torch.randn(Cc, device=d) * 0.1
    st = FT.norm_stats(x, norm, R, Tp, Cc)
    st_div = Tp if norm == "gLN" else 1
    y = torch.empty(M, Cc, device=d)
    dev.dwconv_fwd(x, st, gamma, beta, w, b, R, Tp, Cc, P, dil, st_div, y)
    # torch reference on [R, C, T]
    xr = x.view(R, Tp, Cc).permute(0, 2, 1).contiguous().requires_grad_(True)
    if norm == "gLN":
        mean = xr.mean((1, 2), keepdim=True)
        var = ((xr - mean) ** 2).mean((1, 2), keepdim=True)
        xn = gamma.view(1, -1, 1) * (xr - mean) / torch.sqrt(var + 1e-5) + beta.view(1, -1, 1)
    else:
        xn = F.layer_norm(xr.transpose(1, 2), (Cc,), gamma, beta, 1e-5).transpose(1, 2)
    xn.retain_grad()
    wr = w.view(Cc, 1, P).clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    yr = F.conv1d(xn, wr, br, padding=dil * (P - 1) // 2, dilation=dil, groups=Cc)
    assert rel(y.view(R, Tp, Cc).permute(0, 2, 1), yr) < 1e-5
    dy = torch.randn(M, Cc, device=d)
    yr.backward(dy.view(R, Tp, Cc).permute(0, 2, 1))
    dxn = torch.empty(M, Cc, device=d)
    dw, db = dev.dwconv_bwd(dy, x, st, gamma, beta, w, R, Tp, Cc, P, dil, st_div, dxn)
    assert rel(dxn.view(R, Tp, Cc).permute(0, 2, 1), xn.grad) < 1e-5
    assert rel(dw, wr.grad.view(Cc, P)) < 1e-4
    assert rel(db, br.grad) < 1e-4
    # norm backward on top of it
    gx = FT.norm_backward(x, dxn, st, gamma, norm, R, Tp, Cc)[0]
    assert rel(gx.view(R, Tp, Cc).permute(0, 2, 1), xr.grad) < 1e-4


def test_ola_and_frame_gather():
    from wesep_amd import dev
    d = _cuda()
    torch.manual_seed(3)
    R, Tp, Lk, hop, N = 2, 41, 80, 10, 12
    s = torch.randn(R, N, Tp, device=d)
    w = torch.randn(N, 1, Lk, device=d)
    b = torch.randn(1, device=d)
    full = F.conv_transpose1d(s, w, b, stride=hop).squeeze(1)
    xlen = (Tp - 1) * hop + 20
    frames = torch.einsum("rnt,nk->rtk", s, w[:, 0]).reshape(R * Tp, Lk).contiguous()
    est = torch.empty(R, xlen, device=d)
    dev.ola_fwd(frames, b, R, Tp, Lk, hop, xlen, est)
    assert rel(est, full[:, :xlen]) < 1e-5
    dest = torch.randn(R, xlen, device=d)
    dfr = torch.empty(R * Tp, Lk, device=d)
    dev.ola_bwd(dest, R, Tp, Lk, hop, xlen, dfr)
    pad = torch.zeros(R, (Tp - 1) * hop + Lk, device=d)
    pad[:, :xlen] = dest
    ref = pad.unfold(1, Lk, hop).reshape(R * Tp, Lk)
    assert torch.equal(dfr, ref)
    assert abs(dev.total_sum(dest).item() - dest.double().sum().item()) < 1e-3


def test_gemm_relu_epilogue_and_strided_frames():
    from wesep_amd import functional_tasnet as FT
    from wesep_amd.dev import Rows
    d = _cuda()
    torch.manual_seed(4)
    R, T, Lk, hop, N = 3, 500, 20, 10, 16
    Tp = (T - Lk) // hop + 1
    x = torch.randn(R, T, device=d)
    w, b = torch.randn(N, Lk, device=d), torch.randn(N, device=d)
    out = FT._gemm(x, R * Tp, Lk, w, N, bias=b, act=2, a_rows=Rows(Tp, T, hop), vec=2)
    ref = F.relu(F.conv1d(x.unsqueeze(1), w.unsqueeze(1), b, stride=hop)).permute(0, 2, 1).reshape(R * Tp, N)
    assert rel(out, ref) < 1e-5
    g = torch.randn(R * Tp, N, device=d)
    dW, db = FT._wgrad(g, R * Tp, N, x, Lk, a_rows=Rows(Tp, T, hop), vec=0)
    fr = x.unfold(1, Lk, hop).reshape(R * Tp, Lk)
    assert rel(dW, g.t() @ fr) < 2e-4
    assert rel(db, g.sum(0)) < 1e-4


# ---- assembled model -------------------------------------------------------------------------------
def _build(cfg_kw, seed, d):
    from oracle import convtasnet_oracle as CT
    from wesep_amd.models import get_model
    cfg = CT.ConvTasNetConfig(**cfg_kw)
    params = CT.synth_params(cfg, seed)
    model = get_model("ConvTasNet")(
        N=cfg.N, L=cfg.L, B=cfg.B, H=cfg.H, P=cfg.P, X=cfg.X, R=cfg.R, spk_emb_dim=cfg.spk_emb_dim,
        norm=cfg.norm, spk_fuse_type=cfg.spk_fuse_type, multi_fuse=cfg.multi_fuse,
        use_spk_transform=cfg.use_spk_transform, joint_training=False)
    model.load_state_dict(params, strict=True)
    return cfg, params, model.to(d)


def _gpu_loss(ests, tgt):
    from wesep_amd.functional import SISDRFn
    return sum(w * SISDRFn.apply(e, tgt, 1e-8) for w, e in zip((0.8, 0.1, 0.1), ests))


def _cases():
    from oracle.make_golden import TASNET_CASES
    return sorted(TASNET_CASES)


@pytest.mark.parametrize("name", ["convtasnet_gln_r2_t1600", "convtasnet_cln_xform_r4_t2000",
                                  "convtasnet_gln_l16_r2_t1200", "convtasnet_multiply_r2_t1600",
                                  "convtasnet_additive_cln_r2_t1600", "convtasnet_film_r2_t1600",
                                  "convtasnet_concat_r2_t1600"])
def test_model_matches_oracle_and_reference_fixture(name, golden_dir):
    from oracle import bsrnn_oracle as O
    from oracle import convtasnet_oracle as CT
    from oracle.make_golden import TASNET_CASES
    d = _cuda()
    kw, R, T, seed = TASNET_CASES[name]
    cfg, params, model = _build(kw, seed, d)
    wav, tgt, emb = O.synth_batch(R, T, seed)
    ests = model(wav.to(d), emb.to(d))
    loss = _gpu_loss(ests, tgt.to(d))
    loss.backward()
    # reference fixture (the real wesep model on CPU)
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    for i, e in enumerate(ests):
        assert e.shape == tgt.shape
        assert rel(e, torch.from_numpy(g[f"est{i + 1}"])) < WAV_TOL, i
    assert abs(loss.item() - float(g["loss"])) < DB_TOL
    # oracle autograd for every parameter gradient
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    oloss = CT.multiscale_sisdr_loss(CT.convtasnet_forward(p, cfg, wav, emb), tgt)
    oloss.backward()
    floor = 1e-5 * max(float(v.grad.norm()) for v in p.values())
    for k, prm in model.named_parameters():
        assert prm.grad is not None, k
        err = float((prm.grad.detach().cpu().double() - p[k].grad.double()).norm())
        # a PReLU slope's gradient is ONE scalar: a cancelling sum of dy * x over every negative element of the layer
        # (split-bf16 vs fp32 moves it by up to 6e-3 of its value in the multiply-fusion fixture); tensors: 2e-3
        tol = 1e-2 if prm.numel() == 1 else GRAD_TOL
        assert err <= tol * float(p[k].grad.norm()) + floor, (k, err, float(p[k].grad.norm()))
        assert abs(float(prm.grad.norm()) - float(g["gnorm/" + k])) <= tol * float(g["gnorm/" + k]) + floor, k


@pytest.mark.parametrize("dil,P", [(1, 3), (4, 3), (2, 5)])
def test_causal_dwconv_fwd_bwd(dil, P):
    """ws_dwconv_ex_*(causal = 1): every tap at or before t (convs.py:61-62,91-92: pad dil*(P-1), cut the tail) against
    torch, with the norm applied on load."""
    from wesep_amd import dev
    from wesep_amd import functional_tasnet as FT
    d = _cuda()
    torch.manual_seed(3)
    R, Tp, Cc = 3, 47, 16
    M = R * Tp
    x = torch.randn(M, Cc, device=d) * 1.5 + 0.3
    gamma, beta = torch.rand(Cc, device=d) + 0.5, torch.randn(Cc, device=d) * 0.1
    w, b = torch.randn(Cc, P, device=d) * 0.5, torch.randn(Cc, device=d) * 0.1
    st = FT.norm_stats(x, "cLN", R, Tp, Cc)
    y = torch.empty(M, Cc, device=d)
    dev.dwconv_fwd(x, st, gamma, beta, w, b, R, Tp, Cc, P, dil, 1, y, causal=True)
    xr = x.view(R, Tp, Cc).permute(0, 2, 1).contiguous()
    xn = F.layer_norm(xr.transpose(1, 2), (Cc,), gamma, beta, 1e-5).transpose(1, 2).detach().requires_grad_(True)
    wr, br = w.view(Cc, 1, P).clone().requires_grad_(True), b.clone().requires_grad_(True)
    pad = dil * (P - 1)
    yr = F.conv1d(xn, wr, br, padding=pad, dilation=dil, groups=Cc)[:, :, :-pad]
    assert rel(y.view(R, Tp, Cc).permute(0, 2, 1), yr) < 1e-5
    dy = torch.randn(M, Cc, device=d)
    yr.backward(dy.view(R, Tp, Cc).permute(0, 2, 1))
    dxn = torch.empty(M, Cc, device=d)
    dw, db = dev.dwconv_bwd(dy, x, st, gamma, beta, w, R, Tp, Cc, P, dil, 1, dxn, causal=True)
    assert rel(dxn.view(R, Tp, Cc).permute(0, 2, 1), xn.grad) < 1e-5
    assert rel(dw, wr.grad.view(Cc, P)) < 1e-4 and rel(db, br.grad) < 1e-4


@pytest.mark.parametrize("name", ["convtasnet_plain_skip_r2_t1600", "convtasnet_deep_causal_cln_r2_t1600",
                                  "convtasnet_multi_bn_skip_r4_t1600", "convtasnet_multi_causal_gln_r2_t1600",
                                  "convtasnet_plain_bn_film_r4_t1200"])
def test_variants_match_reference_fixture(name, golden_dir):
    """The rest of the reference constructor (convtasnet.py:16-46): plain / Deep encoder-decoder pairs, skip connections,
    causal blocks, norm = 'BN', sigmoid masks -- estimates, loss, every parameter gradient (element-wise where the tensor
    has <= 4096 entries) and the BatchNorm buffers after the step, from fixtures the REFERENCE produced with its own
    parameter initialisation (stored in the fixture)."""
    from oracle.make_golden import variant_loss
    from tests.test_tasnet_resnet_host_cpu import check_variant, load_variant
    from wesep_amd.models import get_model
    d = _cuda()
    kw, g, params = load_variant(name, golden_dir)
    model = get_model("ConvTasNet")(**kw)
    model.load_state_dict(params, strict=True)
    model = model.to(d).train()
    outs = model(torch.from_numpy(g["wav"]).to(d), torch.from_numpy(g["emb"]).to(d))
    loss = variant_loss(outs, torch.from_numpy(g["tgt"]).to(d))
    loss.backward()
    check_variant(model, g, outs, loss)


def test_joint_training_with_a_wespeaker_encoder_on_fbank(golden_dir):
    """spk_feat = True (convtasnet.py:100-115,188-192): the enrollment is fbank [R, Te, F] through a wespeaker encoder of
    models/resnet.py (ResNet18 here), its embedding conditions the separator, the multi-task head returns logits.  The
    three estimates against oracle(ResNet restatement) -> oracle(Conv-TasNet on that embedding); the loss reaches the
    encoder's first convolution."""
    from oracle import bsrnn_oracle as O
    from oracle import convtasnet_oracle as CT
    from oracle import resnet_oracle as RO
    from wesep_amd.models import get_model
    d = _cuda()
    kw = dict(N=32, L=20, B=32, H=64, P=3, X=2, R=2)
    cfg = CT.ConvTasNetConfig(**kw, spk_emb_dim=64)
    sep = CT.synth_params(cfg, 61)
    rkw = dict(num_blocks=RO.NUM_BLOCKS["ResNet18"], m=32, feat_dim=16, embed_dim=64)
    spk = {"spk_model." + k: v for k, v in RO.synth_params(62, **rkw).items()}
    model = get_model("ConvTasNet")(**kw, spk_emb_dim=64, use_spk_transform=False, joint_training=True, spk_feat=True,
                                    multi_task=True, spksInTrain=7, spk_model="ResNet18",
                                    spk_args=dict(feat_dim=16, embed_dim=64, pooling_func="TSTP", two_emb_layer=False))
    sd = model.state_dict()
    head = {k: v for k, v in sd.items() if k.startswith("pred_linear.")}
    model.load_state_dict({**sep, **spk, **head}, strict=True)
    model = model.to(d).train()
    wav, tgt, _ = O.synth_batch(4, 1600, 61)
    fbank = torch.randn(4, 40, 16, generator=torch.Generator().manual_seed(63))
    outs = model(wav.to(d), fbank.to(d))
    assert len(outs) == 4 and tuple(outs[3].shape) == (4, 7)
    emb = RO.resnet_forward({k[len("spk_model."):]: v.clone() for k, v in spk.items()}, fbank, num_blocks=rkw["num_blocks"], m=32)
    ref = CT.convtasnet_forward({k: v.clone() for k, v in sep.items()}, cfg, wav, emb.detach())
    for i in range(3):
        assert rel(outs[i], ref[i]) < 2e-3, i
    loss = _gpu_loss(outs[:3], tgt.to(d)) + outs[3].square().mean()
    loss.backward()
    gw = model.spk_model.conv1.weight.grad
    assert gw is not None and torch.isfinite(gw).all() and float(gw.norm()) > 0


def test_unbuilt_variants_fail_loudly():
    from wesep_amd.models import get_model
    cls = get_model("ConvTasNet")
    for kw in (dict(joint_training=True, spk_feat=False, feat_type="other"), dict(joint_training=False, encoder_type="Deep"),
               dict(joint_training=True, spk_feat=False, encoder_type="Deep", decoder_type="Deep"),
               dict(joint_training=False, activate="softmax", encoder_type=None, decoder_type=None),
               dict(joint_training=False, spk_fuse_type="nope"), dict(joint_training=False, multi_fuse=False)):
        with pytest.raises(NotImplementedError):
            cls(**kw)
    model = cls(N=16, L=20, B=16, H=32, X=2, R=1, joint_training=False, use_spk_transform=False)
    with pytest.raises(Exception):
        model(torch.randn(2, 400), torch.randn(2, 256))        # CPU tensors: no CPU path


def test_full_size_forward_matches_oracle():
    """SpEx+ size (N=B=256, H=512, X=8, R=4 -> 32 blocks), 2 rows x 4 s: waveforms vs the CPU oracle."""
    from oracle import bsrnn_oracle as O
    from oracle import convtasnet_oracle as CT
    d = _cuda()
    kw = dict(N=256, L=20, B=256, H=512, P=3, X=8, R=4)
    cfg, params, model = _build(kw, 31, d)
    wav, tgt, emb = O.synth_batch(2, 64000, 31)
    with torch.no_grad():
        ests = model(wav.to(d), emb.to(d))
        torch.set_num_threads(min(16, os.cpu_count() or 1))
        ref = CT.convtasnet_forward(params, cfg, wav, emb)
    for e, r in zip(ests, ref):
        assert e.shape == r.shape == wav.shape
        assert rel(e, r) < WAV_TOL


def test_speaker_encoder_kernels():
    """BatchNorm1d (training) + PReLU, MaxPool1d(3), cross entropy against torch."""
    from wesep_amd import dev
    d = _cuda()
    torch.manual_seed(5)
    R, T, Cc = 3, 50, 24
    M = R * T
    x = torch.randn(M, Cc, device=d) * 2 + 1
    gamma, beta = torch.rand(Cc, device=d) + 0.5, torch.randn(Cc, device=d) * 0.1
    res = torch.randn(M, Cc, device=d)
    a = torch.tensor([0.3], device=d)
    rm, rv = torch.zeros(Cc, device=d), torch.ones(Cc, device=d)
    st = torch.empty(2, Cc, device=d)
    dev.bn_stats(x, M, Cc, rm, rv, st)
    u, y = torch.empty_like(x), torch.empty_like(x)
    dev.bn_prelu_fwd(x, st, gamma, beta, res, a, M, Cc, u, y)
    xr = x.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm2, rv2 = torch.zeros(Cc, device=d), torch.ones(Cc, device=d)
    ur = F.batch_norm(xr, rm2, rv2, gr, br, True, 0.1, 1e-5) + res
    yr = F.prelu(ur, a)
    assert rel(u, ur) < 1e-5 and rel(y, yr) < 1e-5
    assert torch.allclose(rm, rm2, rtol=1e-5, atol=1e-6) and torch.allclose(rv, rv2, rtol=1e-5, atol=1e-6)
    du = torch.randn(M, Cc, device=d)
    ur.backward(du)
    dx = torch.empty_like(x)
    sums = dev.bn_bwd(x, du, st, gamma, M, Cc, dx)
    assert rel(dx, xr.grad) < 1e-4
    assert rel(sums[0], br.grad) < 1e-4 and rel(sums[1], gr.grad) < 1e-4
    # max pool
    z = torch.randn(R, T, Cc, device=d)
    z[0, 3:6, :4] = 1.5                                        # ties: the first position wins
    zr = z.permute(0, 2, 1).contiguous().requires_grad_(True)
    pr = F.max_pool1d(zr, 3)
    p = torch.empty(R * (T // 3), Cc, device=d)
    dev.maxpool3_fwd(z.view(M, Cc), R, T, Cc, p)
    assert torch.equal(p.view(R, T // 3, Cc).permute(0, 2, 1), pr)
    dp = torch.randn(R * (T // 3), Cc, device=d)
    pr.backward(dp.view(R, T // 3, Cc).permute(0, 2, 1))
    dz = torch.empty(M, Cc, device=d)
    dev.maxpool3_bwd(z.view(M, Cc), dp, R, T, Cc, dz)
    assert torch.equal(dz.view(R, T, Cc).permute(0, 2, 1), zr.grad)
    # cross entropy
    from wesep_amd.utils.losses import parse_loss
    logits = torch.randn(5, 37, device=d, requires_grad=True)
    label = torch.randint(0, 37, (5,), device=d)
    loss = parse_loss("CE")[0](logits, label)
    (2.0 * loss).backward()
    lr = logits.detach().clone().requires_grad_(True)
    ref = F.cross_entropy(lr, label)
    (2.0 * ref).backward()
    assert abs(loss.item() - ref.item()) < 1e-5
    assert rel(logits.grad, lr.grad) < 1e-5


def _joint_setup(d):
    from oracle import bsrnn_oracle as O
    from oracle import convtasnet_oracle as CT
    from oracle.make_golden import ENROLL_LEN, TASNET_CASES
    from wesep_amd.models import get_model
    name = "spexplus_joint_r4_t1600"
    kw, R, T, seed = TASNET_CASES[name]
    cfg = CT.ConvTasNetConfig(**kw)
    params = CT.synth_params(cfg, seed)
    model = get_model("ConvTasNet")(**kw, use_spk_transform=False)
    model.load_state_dict(params, strict=True)
    wav, tgt, _ = O.synth_batch(R, T, seed)
    enroll, label = CT.synth_enrollment(R, ENROLL_LEN, cfg.spksInTrain, seed)
    return name, cfg, params, model.to(d).train(), wav, tgt, enroll, label


def _check_grads(model, p, tol, CT, floor_rel=1e-5):
    gn = {k: float(v.grad.norm()) for k, v in p.items() if not CT.is_buffer(k)}
    floor = floor_rel * max(gn.values())
    bad = []
    for k, prm in model.named_parameters():
        assert prm.grad is not None, k
        err = float((prm.grad.detach().cpu().double() - p[k].grad.double()).norm())
        if err > tol * gn[k] + floor:
            bad.append((k, err / (gn[k] + 1e-30), gn[k]))
    assert not bad, sorted(bad, key=lambda t: -t[1])[:12]
    return floor


def test_spexplus_joint_forward_and_backward_kernels():
    """SpEx+ joint mode (enrollment through the shared encoder + ResNet4SpExplus, multi-task head), default
    split-bf16 products: forward against the reference fixture (waveforms, logits, loss, BatchNorm running
    statistics after the step), backward against the oracle's autograd under a WELL-CONDITIONED objective --
    a fixed random linear functional of the four outputs.  (The SI-SDR gradient itself is hypersensitive at
    random initialisation: est and target are uncorrelated, <est, target> is a near-total cancellation, and a
    1e-4 perturbation of est moves alpha -- and with it some gradients -- by percents.  That objective is
    checked with the exact-fp32 kernels in the next test.)"""
    from oracle import convtasnet_oracle as CT
    from wesep_amd.utils.losses import parse_loss
    d = _cuda()
    name, cfg, params, model, wav, tgt, enroll, label = _joint_setup(d)
    outs = model(wav.to(d), enroll.to(d))
    assert len(outs) == 4
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
    for i in range(3):
        assert rel(outs[i], torch.from_numpy(g[f"est{i + 1}"])) < WAV_TOL, i
    assert rel(outs[3], torch.from_numpy(g["logits"])) < WAV_TOL
    sisdr, ce = parse_loss(["SISDR", "CE"])
    loss = sum(w * sisdr(e, tgt.to(d)) for w, e in zip((0.8, 0.1, 0.1), outs[:3])) + 0.5 * ce(outs[3], label.to(d))
    assert abs(loss.item() - float(g["loss"])) < DB_TOL
    sd = model.state_dict()
    nbuf = 0
    for k in g.files:                                   # BatchNorm running statistics after the step
        if k.startswith("buf/"):
            assert torch.allclose(sd[k[4:]].cpu(), torch.from_numpy(g[k]), rtol=1e-3, atol=1e-5), k
            nbuf += 1
    assert nbuf == 12 and int(sd["spk_model.aux_enc3.2.batch_norm1.num_batches_tracked"]) == 1
    gen = torch.Generator().manual_seed(7)
    probes = [torch.randn(o.shape, generator=gen) for o in outs]
    sum((o * q.to(d)).sum() for o, q in zip(outs, probes)).backward()
    p = {k: (v.clone() if CT.is_buffer(k) else v.clone().requires_grad_(True)) for k, v in params.items()}
    sum((o * q).sum() for o, q in zip(CT.convtasnet_forward(p, cfg, wav, enroll), probes)).backward()
    # 1e-2 with a floor of 1e-3 of the largest gradient: ReLU / PReLU / MaxPool kinks (a 1e-4 perturbation flips
    # the side of entries near zero: measured 7.6e-3 on decoder.mask2) and scalar PReLU-slope gradients that are
    # near-total cancellations; kernel bugs show up as O(1) errors, and the next test holds 3e-3 with exact products
    _check_grads(model, p, 1e-2, CT, floor_rel=1e-3)


def test_spexplus_joint_training_loss_gradients_fp32(monkeypatch):
    """Same model, the training objective of spexplus.yaml (.8/.1/.1 SI-SDR + .5 CE), exact-fp32 products:
    every parameter gradient against the oracle and against the reference fixture's gradient norms."""
    from oracle import convtasnet_oracle as CT
    from wesep_amd.utils.losses import parse_loss
    monkeypatch.setenv("WESEP_GEMM", "f32")
    d = _cuda()
    name, cfg, params, model, wav, tgt, enroll, label = _joint_setup(d)
    outs = model(wav.to(d), enroll.to(d))
    sisdr, ce = parse_loss(["SISDR", "CE"])
    loss = sum(w * sisdr(e, tgt.to(d)) for w, e in zip((0.8, 0.1, 0.1), outs[:3])) + 0.5 * ce(outs[3], label.to(d))
    loss.backward()
    p = {k: (v.clone() if CT.is_buffer(k) else v.clone().requires_grad_(True)) for k, v in params.items()}
    CT.spexplus_loss(CT.convtasnet_forward(p, cfg, wav, enroll), tgt, label).backward()
    floor = _check_grads(model, p, 3e-3, CT)
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
    assert abs(loss.item() - float(g["loss"])) < DB_TOL
    for k, prm in model.named_parameters():
        assert abs(float(prm.grad.norm()) - float(g["gnorm/" + k])) <= 3e-3 * float(g["gnorm/" + k]) + floor, k


@pytest.mark.parametrize("R,T", [(1, 1234), (3, 167), (2, 20)])
def test_ragged_lengths_and_single_row(R, T):
    """Edge cases of the reference's framing: T not a multiple of the stride (the tail that does not fill a
    short-window frame is dropped, the middle / long windows see zero extension, encoder.py:104-111), the shortest
    legal input (one frame), a single row and an odd row count -- forward and input-independent gradients vs the
    oracle."""
    from oracle import convtasnet_oracle as CT
    d = _cuda()
    kw = dict(N=16, L=20, B=16, H=24, P=3, X=3, R=1)
    cfg, params, model = _build(kw, 41, d)
    g = torch.Generator().manual_seed(R * 1000 + T)
    wav, emb = torch.randn(R, T, generator=g) * 0.1, torch.randn(R, 256, generator=g)
    ests = model(wav.to(d), emb.to(d))
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    ref = CT.convtasnet_forward(p, cfg, wav, emb)
    Tp = (T - 20) // 10 + 1
    probes = [torch.randn(R, (Tp - 1) * 10 + 20, generator=g) for _ in range(3)]
    for e, r in zip(ests, ref):
        assert e.shape == r.shape == (R, (Tp - 1) * 10 + 20)
        assert rel(e, r) < WAV_TOL
    sum((e * q.to(d)).sum() for e, q in zip(ests, probes)).backward()
    sum((r * q).sum() for r, q in zip(ref, probes)).backward()
    gn = max(float(v.grad.norm()) for v in p.values())
    for k, prm in model.named_parameters():
        err = float((prm.grad.detach().cpu().double() - p[k].grad.double()).norm())
        assert err <= 5e-3 * float(p[k].grad.norm()) + 1e-4 * gn, (k, err)


def test_input_shorter_than_the_window_raises():
    d = _cuda()
    _, _, model = _build(dict(N=16, L=20, B=16, H=24, P=3, X=2, R=1), 42, d)
    with pytest.raises(RuntimeError):
        model(torch.randn(2, 12, device=d), torch.randn(2, 256, device=d))
    with pytest.raises(RuntimeError):
        model(torch.randn(2, 3, 100, device=d), torch.randn(2, 256, device=d))
