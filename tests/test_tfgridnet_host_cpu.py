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
