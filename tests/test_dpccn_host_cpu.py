"""CPU: host logic of the DPCCN path -- shapes, row addressing, weight re-layouts and the
autograd wiring of wesep_amd/functional_dpccn.py + models/dpccn.py -- with the device entry points replaced by the
torch emulation of tests/emu_dev.py, against the reference fixtures and the oracle.  The HIP kernels themselves are
checked on the GPU (tests/test_dpccn_gpu.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import bsrnn_oracle as O
from oracle import dpccn_oracle as DP
from oracle.make_golden import DPCCN_CASES
from tests import emu_dev


def _run(name, monkeypatch, golden_dir):
    from wesep_amd.models import get_model
    emu_dev.install(monkeypatch)
    kw, R, T, seed = DPCCN_CASES[name]
    cfg = DP.DPCCNConfig(**kw)
    params = DP.synth_params(cfg, seed)
    model = get_model("DPCCN")(**kw, joint_training=False)
    model.load_state_dict(params, strict=True)
    model.train()
    wav, tgt, emb = O.synth_batch(R, T, seed)
    est, dummy = model(wav, emb)
    loss = O.sisdr_loss(est, tgt)
    loss.backward()
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    return model, est, loss, g


@pytest.mark.parametrize("name", ["dpccn_multiply_r2_t4480", "dpccn_additive_xform_r2_t4608", "dpccn_film_r2_t4352",
                                  "dpccn_concat_xform_r2_t4352", "dpccn_causal_r2_t4480"])
def test_dpccn_host_logic_matches_reference_fixture(name, monkeypatch, golden_dir):
    model, est, loss, g = _run(name, monkeypatch, golden_dir)
    ref = g["est"]
    assert est.shape == ref.shape
    assert np.linalg.norm(est.detach().numpy() - ref) / np.linalg.norm(ref) < 1e-3
    assert abs(loss.item() - float(g["loss"])) < 1e-2
    floor = 1e-4 * max(float(g["gnorm/" + k]) for k, _ in model.named_parameters())
    for k, prm in model.named_parameters():
        gn = float(g["gnorm/" + k])
        assert prm.grad is not None, k
        assert abs(float(prm.grad.norm()) - gn) <= 2e-2 * gn + floor, (k, float(prm.grad.norm()), gn)


@pytest.mark.parametrize("C0,g,co5,x_grad", [(16, 16, 16, True), (64, 32, 64, True), (32, 16, 32, False), (8, 4, 12, True)])
def test_dense_block_function_matches_the_concatenating_composition(C0, g, co5, x_grad, monkeypatch):
    """functional_dpccn.DenseBlockFn (one feature map, halo convolutions, input gradient per channel block) against the
    torch composition it replaces -- conv2d / ELU / InstanceNorm2d with torch.cat between the layers -- in fp32: output,
    every weight / bias gradient and the input gradient; with an input that needs no gradient the last block
    convolution is skipped."""
    from wesep_amd import functional_dpccn as FD
    emu_dev.install(monkeypatch)
    monkeypatch.setattr(torch.Tensor, "is_cuda", property(lambda self: True))
    gen = torch.Generator().manual_seed(C0 + g)
    B, H, W = 2, 5, 7
    M = B * H * W
    x = torch.randn(M, C0, generator=gen, dtype=torch.float32).requires_grad_(x_grad)
    params = []
    for i in range(5):
        ci, co = C0 + i * g, (g if i < 4 else co5)
        params += [(0.2 * torch.randn(co, ci, 3, 3, generator=gen, dtype=torch.float32)).requires_grad_(True),
                   (0.1 * torch.randn(co, generator=gen, dtype=torch.float32)).requires_grad_(True)]
    dout = torch.randn(M, co5, generator=gen, dtype=torch.float32)
    out = FD.DenseBlockFn.apply(x, (B, H, W), *params)
    out.backward(dout)
    got = [out.detach().clone(), x.grad.clone() if x_grad else None] + [p.grad.clone() for p in params]
    for p in params:
        p.grad = None
    x2 = x.detach().clone().requires_grad_(x_grad)
    feats = [x2.view(B, H, W, C0).permute(0, 3, 1, 2)]
    for i in range(5):
        y = torch.nn.functional.conv2d(torch.cat(feats, 1), params[2 * i], params[2 * i + 1], padding=1)
        feats.append(torch.nn.functional.instance_norm(torch.nn.functional.elu(y), eps=1e-5))
    ref = feats[-1].permute(0, 2, 3, 1).reshape(M, co5)
    ref.backward(dout)
    want = [ref.detach(), x2.grad if x_grad else None] + [p.grad for p in params]
    for a, b in zip(got, want):
        if b is None:
            assert a is None
            continue
        assert float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30)) < 2e-5
