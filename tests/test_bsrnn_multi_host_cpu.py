"""CPU: host logic of `BSRNN_Multi` (wesep_amd/models/bsrnn_multi_optim.py; reference bsrnn_multi_optim.py:300-470)
and of the pBSRNN module tree, on the coarse emulation of the pBSRNN autograd functions (tests/emu_bsrnn.py) plus
the entry-point emulation (tests/emu_dev.py) for the speaker encoder / front-end.  Test-only: the product has no CPU
path.  Fixtures come from the real reference (oracle/make_golden.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import bsrnn_oracle as O
from oracle.make_golden import CASES, MULTI_CASES, MULTI_LOSS_WEIGHT, synth_multi_params
from tests import emu_bsrnn, emu_dev


@pytest.fixture
def emu(monkeypatch):
    emu_dev.install(monkeypatch)
    emu_bsrnn.install(monkeypatch)
    monkeypatch.setattr(torch.Tensor, "is_cuda", property(lambda self: True))


@pytest.fixture
def emu_real_resrnn(monkeypatch):
    from tests import emu_blk
    emu_dev.install(monkeypatch)
    emu_blk.install(monkeypatch)
    emu_bsrnn.install(monkeypatch, real_resrnn=True)
    monkeypatch.setattr(torch.Tensor, "is_cuda", property(lambda self: True))


@pytest.mark.parametrize("name", ["bsrnn_multiply_r2_t4000", "bsrnn_film_multi_r2_t3000"])
def test_bsrnn_with_production_resrnn_matches_reference_fixture(name, golden_dir, emu_real_resrnn):
    """As below, but the separator runs the PRODUCT's blocked-layout ResRNN host code (functional.ResRNNBlkFn: Z-layout
    sequence maps of both views, packs, gradient routing) on the blocked-layout emulation -- the real-reference fixture
    is reproduced end to end through it, gradients included."""
    _check_bsrnn_fixture(name, golden_dir, tol=5e-5, gtol=2e-3)


@pytest.mark.parametrize("name", sorted(CASES))
def test_bsrnn_module_tree_on_emulation_matches_reference_fixture(name, golden_dir, emu):
    _check_bsrnn_fixture(name, golden_dir, tol=1e-5, gtol=1e-3)


def _check_bsrnn_fixture(name, golden_dir, tol, gtol):
    """Validates the harness itself and the BSRNN host code (module tree, parameter routing, fusion variants): the
    real-reference fixtures must be reproduced through the product's modules."""
    from wesep_amd.models import get_model
    from wesep_amd.utils.losses import parse_loss
    kw, R, T, seed = CASES[name]
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    cfg = O.BSRNNConfig(**kw)
    model = get_model("BSRNN")(spk_emb_dim=cfg.spk_emb_dim, num_repeat=cfg.num_repeat,
                               use_spk_transform=cfg.use_spk_transform, spk_fuse_type=cfg.spk_fuse_type,
                               multi_fuse=cfg.multi_fuse, joint_training=False)
    model.load_state_dict(O.synth_params(cfg, seed), strict=True)
    model.train()
    wav, tgt, emb = O.synth_batch(R, T, seed)
    est, dummy = model(wav, emb)
    assert dummy.dim() == 0
    assert np.linalg.norm(est.detach().numpy() - g["est"]) / np.linalg.norm(g["est"]) < tol
    loss = parse_loss("SISDR")[0](est, tgt)
    assert abs(loss.item() - float(g["loss"])) < 1e-3
    loss.backward()
    for k, prm in model.named_parameters():
        gn = float(g["gnorm/" + k])
        assert abs(float(prm.grad.norm()) - gn) <= gtol * gn + 1e-7, k


@pytest.mark.parametrize("name", sorted(MULTI_CASES))
def test_bsrnn_multi_matches_reference_fixture(name, golden_dir, emu):
    from wesep_amd.models import get_model
    from wesep_amd.utils.executor import Executor
    from wesep_amd.utils.losses import parse_loss
    kw, spk_model, R, T, Tw, seed = MULTI_CASES[name]
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    cfg = O.BSRNNConfig(**kw)
    model = get_model("BSRNN_Multi")(
        spk_emb_dim=cfg.spk_emb_dim, num_repeat=cfg.num_repeat, use_spk_transform=cfg.use_spk_transform,
        spk_fuse_type=cfg.spk_fuse_type, multi_fuse=cfg.multi_fuse, joint_training=True, multi_task=False,
        spk_model=spk_model, spk_model_init=False, spk_feat=False, feat_type="consistent",
        spk_args=dict(feat_dim=80, embed_dim=cfg.spk_emb_dim, pooling_func="TSTP", two_emb_layer=False))
    params = synth_multi_params(cfg, spk_model, seed)
    missing, unexpected = model.load_state_dict(params, strict=False)
    assert not unexpected
    assert set(missing) == {"preEmphasis.flipped_filter", "spk_encoder.spectrogram.window", "spk_encoder.mel_scale.fb"}
    model.train()
    wav, tgt, enroll = torch.from_numpy(g["wav"]), torch.from_numpy(g["tgt"]), torch.from_numpy(g["enroll"])
    outs = model(wav, enroll)
    assert len(outs) == 4
    # fp32 reassociation in the GEMM-shaped front-end / encoder moves the embedding by ~1e-5 relative, the estimates
    # follow; the second pass re-encodes the first estimate through 2-row BatchNorm (amplifies, see the oracle test)
    for got, key, tol in ((outs[0], "est", 1e-4), (outs[1], "self_est", 1e-3), (outs[2], "emb1", 1e-4),
                          (outs[3], "emb2", 1e-3)):
        assert np.linalg.norm(got.detach().numpy() - g[key]) / np.linalg.norm(g[key]) < tol, key
    # the recipe's loss through the Executor's composition (loss_posi [[0, 1]], loss_weight [[0.4, 0.6]])
    loss = Executor._loss(outs, tgt, None, parse_loss("SISDR"), ([[0, 1]], [list(MULTI_LOSS_WEIGHT)]), False)
    assert abs(loss.item() - float(g["loss"])) < 1e-2            # -50 dB SI-SDR: see tests/test_oracle_golden.py
    loss.backward()
    for k, prm in model.named_parameters():
        gn = float(g["gnorm/" + k])
        assert prm.grad is not None and abs(float(prm.grad.norm()) - gn) <= 2e-2 * gn + 1e-6, k
    with torch.no_grad():                                        # no-grad mode: the plain BSRNN pair
        pair = model(wav, enroll)
    assert len(pair) == 2 and tuple(pair[1].shape) == (R, cfg.spk_emb_dim)


def test_bsrnn_multi_constructor_contract():
    from wesep_amd.models import get_model
    spk = dict(spk_model="ResNet18", spk_args=dict(feat_dim=80, embed_dim=256, pooling_func="TSTP",
                                                   two_emb_layer=False))
    with pytest.raises(NotImplementedError):
        get_model("BSRNN_Multi")(num_repeat=1, joint_training=False)
    with pytest.raises(NotImplementedError):                      # fbank enrollment: the estimate cannot be re-encoded
        get_model("BSRNN_Multi")(num_repeat=1, joint_training=True, spk_feat=True, **spk)
    m = get_model("BSRNN_Multi")(num_repeat=1, joint_training=True, spk_feat=False, **spk)
    ref = get_model("BSRNN")(num_repeat=1, joint_training=True, spk_feat=False, **spk)
    assert list(m.state_dict().keys()) == list(ref.state_dict().keys())
