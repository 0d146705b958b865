"""GPU parity of the DPCCN path: every new conv2d.hip kernel against its torch statement
(tests/emu_dev.py, the same functions that stand in for the device in the CPU host-logic test), and the assembled
model against the fixtures generated from the real reference (waveform <= 1e-3, loss <= 1e-2 dB, gradient norms)."""
import os

import numpy as np
import pytest
import torch

from tests import emu_dev as E

pytestmark = pytest.mark.gpu


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


@pytest.mark.parametrize("name", ["dpccn_multiply_r2_t4480", "dpccn_additive_xform_r2_t4608"])
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
    floor = 1e-3 * max(float(g["gnorm/" + k]) for k, _ in model.named_parameters())
    bad = []
    for k, prm in model.named_parameters():
        gn = float(g["gnorm/" + k])
        if prm.grad is None or abs(float(prm.grad.norm()) - gn) > 3e-2 * gn + floor:
            bad.append((k, None if prm.grad is None else float(prm.grad.norm()), gn))
    assert not bad, bad[:8]


def test_dpccn_unbuilt_variants_fail_loudly():
    from wesep_amd.models import get_model
    for kw in (dict(joint_training=False, spk_fuse_type="concat"), dict(joint_training=False, causal=True),
               dict(joint_training=False, stride2=(1, 1))):
        with pytest.raises(NotImplementedError):
            get_model("DPCCN")(**kw)
    m = get_model("DPCCN")(joint_training=False, tcn_blocks=1, tcn_layers=1)
    with pytest.raises(Exception):
        m(torch.randn(2, 4480), torch.randn(2, 256))            # CPU tensors: no CPU path
