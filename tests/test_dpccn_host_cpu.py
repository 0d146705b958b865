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
