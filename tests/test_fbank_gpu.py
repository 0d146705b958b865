"""GPU parity of the device kaldi fbank + CMN (`wesep_amd/utils/funcs.py`) and of the
Executor's SSA self-enrollment pass (executor.py:89-102), through the C ABI.  Checked against the fixtures that hold
outputs of the reference's own C++ kaldi front-end (runtime/frontend/fbank.h via oracle/build_ref.py) and against
oracle/fbank_oracle.py at BASELINE size.  Tolerance: absolute, on log-mel energies of magnitude 10..25; the C++
reference itself (fp32 table FFT, logf) sits within 1e-4 of the float64 restatement."""
import os

import numpy as np
import pytest
import torch

from oracle import fbank_oracle as FB
from oracle.make_golden import FBANK_CASES, synth_fbank_wave

pytestmark = pytest.mark.gpu


def _close(got, want, max_tol=2e-3, mean_tol=3e-5):
    """Log-mel energies: the mean error is what a defect moves (CPU fp32 restatement: 5e-7 .. 5e-6 against float64 /
    the C++ reference); the maximum has a heavy tail from the lowest filters, which span one or two FFT bins -- a
    spectral null there turns fp32 summation-order differences into 1e-4-sized log errors (measured 1.4e-4 over 1 M
    values with the CPU emulation), so it gets a loose bound only."""
    err = np.abs(np.asarray(got, dtype=np.float64) - np.asarray(want, dtype=np.float64))
    assert err.mean() < mean_tol and err.max() < max_tol, (err.mean(), err.max())


def _cuda():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.mark.parametrize("name", sorted(FBANK_CASES))
def test_fbank_matches_reference_cpp_fixture(name, golden_dir):
    from wesep_amd.utils.funcs import apply_cmvn, compute_fbank
    d = _cuda()
    R, T, sr, nb, seed = FBANK_CASES[name]
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    wav = torch.from_numpy(synth_fbank_wave(R, T, sr, seed)).to(d)
    feats = compute_fbank(wav, num_mel_bins=nb, dither=0.0, sample_rate=sr)
    _close(feats.cpu().numpy(), g["fbank"])
    _close(apply_cmvn(feats).cpu().numpy(), g["fbank_cmn"])


def test_fbank_full_size_and_ragged_lengths():
    """BASELINE enrollment shape (32 rows x 4 s -> [32, 398, 80]) and a length with T % 4 != 0, against the oracle."""
    from wesep_amd.utils.funcs import apply_cmvn, compute_fbank
    d = _cuda()
    rng = np.random.default_rng(21)
    for R, T in ((32, 64000), (3, 12345), (2, 400)):
        wav = (0.1 * rng.standard_normal((R, T)) + 0.01).astype(np.float32)
        got = apply_cmvn(compute_fbank(torch.from_numpy(wav).to(d), dither=0.0)).cpu().numpy()
        want = FB.apply_cmvn(FB.compute_fbank(wav, dither=0.0))
        assert got.shape == want.shape == (R, 1 + (T - 400) // 160, 80)
        _close(got, want)
    sil = compute_fbank(torch.zeros(1, 800, device=d), dither=0.0)
    assert torch.allclose(sil, torch.full_like(sil, float(np.log(FB.FLT_EPS))))


def test_fbank_dither_statistics():
    from wesep_amd.utils.funcs import compute_fbank
    d = _cuda()
    torch.manual_seed(3)
    got = compute_fbank(torch.zeros(4, 16000, device=d), dither=1.0).cpu().numpy()
    ref = FB.compute_fbank(np.zeros((4, 16000), np.float32), dither=1.0, rng=np.random.default_rng(9))
    assert np.abs(got.mean((0, 1)) - ref.mean((0, 1))).max() < 0.4     # > 4 sigma of the per-bin mean difference
    assert abs(got.mean() - ref.mean()) < 0.03


def test_executor_ssa_step_on_joint_model():
    """One SSA step of the jointly trained pBSRNN + ResNet18 (spk_feat True): the step must equal the manual
    two-pass computation (dither 0), and every parameter must receive a gradient."""
    import random
    from wesep_amd.models import get_model
    from wesep_amd.utils.executor import Executor
    from wesep_amd.utils.funcs import apply_cmvn, compute_fbank
    from wesep_amd.utils.losses import parse_loss
    from wesep_amd.utils.schedulers import ExponentialDecrease
    d = _cuda()
    torch.manual_seed(0)
    model = get_model("BSRNN")(num_repeat=1, spk_fuse_type="multiply", multi_fuse=False, use_spk_transform=False,
                               joint_training=True, spk_model="ResNet18", spk_feat=True,
                               spk_args=dict(feat_dim=80, embed_dim=256, pooling_func="TSTP",
                                             two_emb_layer=False)).to(d)
    model.train()
    g = torch.Generator().manual_seed(4)
    wav = 0.1 * torch.randn(2, 16000, generator=g)
    tgt = 0.1 * torch.randn(2, 16000, generator=g)
    enroll = torch.randn(2, 98, 80, generator=g)
    args = dict(num_mel_bins=80, frame_length=25, frame_shift=10, dither=0.0)
    state = {k: v.clone() for k, v in model.state_dict().items()}
    crit = parse_loss("SISDR")
    # manual two-pass
    with torch.no_grad():
        est0 = model(wav.to(d), enroll.to(d))[0]
        fb = apply_cmvn(compute_fbank(est0, **args, sample_rate=16000))
    assert tuple(fb.shape) == (2, 98, 80)
    want = FB.apply_cmvn(FB.compute_fbank(est0.cpu().numpy(), dither=0.0))
    _close(fb.cpu().numpy(), want)
    loss_manual = float(crit[0](model(wav.to(d), fb)[0], tgt.to(d)).mean())
    # executor (BatchNorm running statistics were advanced by the passes above: restore)
    model.load_state_dict(state)
    opt = torch.optim.SGD(model.parameters(), lr=0.0)
    sched = ExponentialDecrease(opt, num_epochs=1, epoch_iter=1, initial_lr=1e-9, final_lr=1e-9, warm_up_epoch=0)
    batch = {"wav_mix": wav, "wav_targets": tgt, "spk_embeds": enroll, "spk_label": torch.zeros(0)}
    random.seed(0)
    loss, _ = Executor().train([batch], [model], 1, [opt], crit, [sched], scaler=None, epoch=1, enable_amp=False,
                               logger=None, device=d, se_loss_weight=([[0]], [[1.0]]), SSA_enroll_prob=1.0,
                               fbank_args=args, sample_rate=16000, speaker_feat=True)
    assert np.isfinite(loss) and abs(loss - loss_manual) < 1e-2, (loss, loss_manual)
    assert all(p.grad is not None for p in model.parameters())
