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
        norm=cfg.norm, multi_fuse=cfg.multi_fuse, use_spk_transform=cfg.use_spk_transform, joint_training=False)
    model.load_state_dict(params, strict=True)
    return cfg, params, model.to(d)


def _gpu_loss(ests, tgt):
    from wesep_amd.functional import SISDRFn
    return sum(w * SISDRFn.apply(e, tgt, 1e-8) for w, e in zip((0.8, 0.1, 0.1), ests))


def _cases():
    from oracle.make_golden import TASNET_CASES
    return sorted(TASNET_CASES)


@pytest.mark.parametrize("name", ["convtasnet_gln_r2_t1600", "convtasnet_cln_xform_r4_t2000",
                                  "convtasnet_gln_l16_r2_t1200"])
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
        assert err <= GRAD_TOL * float(p[k].grad.norm()) + floor, (k, err, float(p[k].grad.norm()))
        assert abs(float(prm.grad.norm()) - float(g["gnorm/" + k])) <= GRAD_TOL * float(g["gnorm/" + k]) + floor, k


def test_unbuilt_variants_fail_loudly():
    from wesep_amd.models import get_model
    cls = get_model("ConvTasNet")
    for kw in (dict(joint_training=True), dict(joint_training=False, encoder_type="Deep"),
               dict(joint_training=False, skip_con=True), dict(joint_training=False, norm="BN"),
               dict(joint_training=False, spk_fuse_type="FiLM"), dict(joint_training=False, causal=True)):
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
