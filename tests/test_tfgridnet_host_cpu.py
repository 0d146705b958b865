"""CPU: host logic of the TF-GridNet path with the device entry points replaced by the torch
emulation of tests/emu_dev.py, against the reference fixtures: zero-padded hidden-256 recurrences, window row views and
their overlap-add adjoints, the flattened layer norms and the per-(row, head) attention products."""
import os

import numpy as np
import pytest
import torch

from oracle import bsrnn_oracle as O
from oracle import tfgridnet_oracle as TG
from oracle.make_golden import TFGRIDNET_CASES, tfgridnet_batch
from tests import emu_dev


@pytest.mark.parametrize("name", sorted(TFGRIDNET_CASES))
def test_tfgridnet_host_logic_matches_reference_fixture(name, monkeypatch, golden_dir):
    from wesep_amd.models import get_model
    emu_dev.install(monkeypatch)
    kw, R, T, seed = TFGRIDNET_CASES[name]
    cfg = TG.TFGridNetConfig(**kw)
    params = TG.synth_params(cfg, seed)
    model = get_model("TFGridNet")(**kw, joint_training=False)
    model.load_state_dict(params, strict=True)
    model.train()
    wav, tgt, emb = tfgridnet_batch(cfg, R, T, seed)
    est, dummy = model(wav, emb)
    loss = O.sisdr_loss(est, tgt)
    loss.backward()
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    ref = g["est"]
    assert est.shape == ref.shape
    assert np.linalg.norm(est.detach().numpy() - ref) / np.linalg.norm(ref) < 1e-3
    assert abs(loss.item() - float(g["loss"])) < 1e-2
    floor = 1e-4 * max(float(g["gnorm/" + k]) for k, _ in model.named_parameters())
    for k, prm in model.named_parameters():
        gn = float(g["gnorm/" + k])
        assert prm.grad is not None, k
        assert abs(float(prm.grad.norm()) - gn) <= 2e-2 * gn + floor, (k, float(prm.grad.norm()), gn)


@pytest.mark.parametrize("B,T,Q,nh,E,cp", [(2, 5, 9, 4, 8, 12), (1, 6, 5, 2, 4, 4)])
def test_qkv_heads_function_matches_the_per_head_composition(B, T, Q, nh, E, cp, monkeypatch):
    """functional_tfgridnet.QKVHeadsFn (one projection output, one kernel per projection) against the composition it
    replaces -- per head PReLU, LayerNorm over (channel, bin) with the head's affine, heads stacked along the batch, keys /
    values zero-padded to a multiple of four frames -- in torch autograd: the three outputs and every gradient."""
    from wesep_amd import functional_tfgridnet as FG
    emu_dev.install(monkeypatch)
    monkeypatch.setattr(torch.Tensor, "is_cuda", property(lambda self: True))
    gen = torch.Generator().manual_seed(B * 10 + Q)
    Tp = -(-T // 4) * 4
    M, ld = B * T * Q, nh * (2 * E + cp)
    qkv = torch.randn(M, ld, generator=gen).requires_grad_(True)
    par = []
    for ch in (E, E, cp):
        par += [(0.1 + 0.4 * torch.rand(nh, generator=gen)).requires_grad_(True),
                (1 + 0.3 * torch.randn(nh, Q * ch, generator=gen)).requires_grad_(True),
                (0.3 * torch.randn(nh, Q * ch, generator=gen)).requires_grad_(True)]
    douts = [torch.randn(nh * B, tp, Q * ch, generator=gen) for ch, tp in ((E, T), (E, Tp), (cp, Tp))]
    outs = FG.QKVHeadsFn.apply(qkv, (B, T, Tp, Q, nh, E, cp), *par)
    sum((o * d).sum() for o, d in zip(outs, douts)).backward()
    got = [o.detach().clone() for o in outs] + [qkv.grad.clone()] + [p.grad.clone() for p in par]
    qkv.grad = None
    for p in par:
        p.grad = None
    refs, off = [], 0
    for j, (ch, tp) in enumerate(((E, T), (E, Tp), (cp, Tp))):
        s, gm, bt = par[3 * j:3 * j + 3]
        x = qkv[:, off:off + nh * ch].reshape(B, T, Q, nh, ch)
        heads = []
        for h in range(nh):
            u = torch.nn.functional.prelu(x[:, :, :, h, :], s[h:h + 1])                       # [B, T, Q, ch]
            mu = u.mean((2, 3), keepdim=True)
            var = ((u - mu) ** 2).mean((2, 3), keepdim=True)
            n = (u - mu) / torch.sqrt(var + 1e-5)
            y = n * gm[h].view(1, 1, Q, ch) + bt[h].view(1, 1, Q, ch)
            heads.append(torch.nn.functional.pad(y.reshape(B, T, Q * ch), (0, 0, 0, tp - T)))
        refs.append(torch.stack(heads, 0).reshape(nh * B, tp, Q * ch))
        off += nh * ch
    sum((o * d).sum() for o, d in zip(refs, douts)).backward()
    want = [r.detach() for r in refs] + [qkv.grad] + [p.grad for p in par]
    for a, b in zip(got, want):
        assert float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30)) < 2e-5
