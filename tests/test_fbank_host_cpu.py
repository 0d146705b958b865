"""CPU: host logic of the device kaldi fbank / CMN (`wesep_amd/utils/funcs.py`) and of
the Executor's SSA self-enrollment pass.

Two layers:
  * the oracle (oracle/fbank_oracle.py) against the fixtures that hold outputs of the reference's own C++ kaldi
    front-end (runtime/frontend/fbank.h, compiled by oracle/build_ref.py), and against that library live when it
    was built;
  * the product's folded basis / overlapping-row-view composition with the HIP entry points replaced by the torch
    emulation of tests/emu_dev.py (test-only; the product has no CPU path) against the same fixtures."""
import os

import numpy as np
import pytest
import torch

from oracle import fbank_oracle as FB
from oracle.make_golden import FBANK_CASES, synth_fbank_wave
from tests import emu_dev

# the C++ reference evaluates its FFT and logf in fp32; observed oracle-vs-reference differences are <= 1e-4 on
# log-energies of magnitude 10..25
REF_ATOL = 3e-4


@pytest.fixture
def emu(monkeypatch):
    emu_dev.install(monkeypatch)
    monkeypatch.setattr(torch.Tensor, "is_cuda", property(lambda self: True))


def _case(name, golden_dir):
    R, T, sr, nb, seed = FBANK_CASES[name]
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    wav = synth_fbank_wave(R, T, sr, seed)
    assert np.array_equal(wav, g["wav"])
    return wav, sr, nb, g


@pytest.mark.parametrize("name", sorted(FBANK_CASES))
def test_oracle_matches_reference_cpp_fixture(name, golden_dir):
    wav, sr, nb, g = _case(name, golden_dir)
    got = FB.compute_fbank(wav, nb, 25, 10, 0.0, sr)
    assert got.shape == g["fbank"].shape
    assert np.abs(got - g["fbank"]).max() < REF_ATOL
    assert np.abs(FB.apply_cmvn(got) - g["fbank_cmn"]).max() < REF_ATOL
    got32 = FB.compute_fbank(wav, nb, 25, 10, 0.0, sr, dtype=np.float32)
    assert np.abs(got32 - g["fbank"]).max() < REF_ATOL


@pytest.mark.skipif(not FB.ref_available(), reason="oracle/_ref/libref_fbank.so not built")
def test_oracle_matches_reference_cpp_live():
    rng = np.random.default_rng(5)
    for T, sr, nb in ((16000, 16000, 80), (400, 16000, 80), (719, 16000, 40), (8000, 8000, 40)):
        w = (0.2 * rng.standard_normal(T) + 0.05).astype(np.float32) * np.float32(1 << 15)
        ref = FB.ref_fbank(w, nb, sr)
        got = FB.kaldi_fbank(w, nb, 25, 10, 0.0, sr)
        assert ref.shape == got.shape == (1 + (T - sr // 40) // (sr // 100), nb)
        assert np.abs(ref - got).max() < REF_ATOL
    assert FB.ref_fbank(np.zeros(399, np.float32)).shape == (0, 80)          # shorter than a frame: no frames
    z = FB.ref_fbank(np.zeros(1600, np.float32))                            # digital silence: the log floor
    assert np.allclose(z, np.log(FB.FLT_EPS)) and np.allclose(FB.kaldi_fbank(np.zeros(1600)), np.log(FB.FLT_EPS))


def test_mel_bank_shape_and_partition():
    bank = FB.mel_banks(80, 512, 16000)
    assert bank.shape == (80, 257) and bank[:, 256].max() == 0.0 and bank.min() >= 0.0
    # neighbouring triangles sum to one between the first and last centre frequencies
    centres = bank.argmax(1)
    inner = bank[:, centres[0] + 1:centres[-1]].sum(0)
    assert np.abs(inner - 1.0).max() < 1e-9


@pytest.mark.parametrize("name", sorted(FBANK_CASES))
def test_device_composition_matches_reference_fixture(name, golden_dir, emu):
    from wesep_amd.utils.funcs import apply_cmvn, compute_fbank
    wav, sr, nb, g = _case(name, golden_dir)
    feats = compute_fbank(torch.from_numpy(wav), num_mel_bins=nb, dither=0.0, sample_rate=sr)
    assert tuple(feats.shape) == g["fbank"].shape
    assert np.abs(feats.numpy() - g["fbank"]).max() < REF_ATOL
    cmn = apply_cmvn(feats)
    assert np.abs(cmn.numpy() - g["fbank_cmn"]).max() < REF_ATOL
    assert cmn.mean(1).abs().max() < 1e-4
    assert apply_cmvn(feats, norm_mean=False) is feats


def test_device_composition_unaligned_lengths_and_silence(emu):
    from wesep_amd.utils.funcs import compute_fbank
    rng = np.random.default_rng(11)
    wav = (0.1 * rng.standard_normal((2, 1203))).astype(np.float32)         # T % 4 != 0: scalar A loads
    got = compute_fbank(torch.from_numpy(wav), dither=0.0).numpy()
    ref = FB.compute_fbank(wav, dither=0.0)
    assert got.shape == ref.shape == (2, 6, 80)
    assert np.abs(got - ref).max() < REF_ATOL
    sil = compute_fbank(torch.zeros(1, 800), dither=0.0)
    assert torch.allclose(sil, torch.full_like(sil, float(np.log(FB.FLT_EPS))))


def test_dither_statistics(emu):
    """dither=1.0 on digital silence: every frame is unit Gaussian noise through the kaldi chain; the mean log-mel
    energy per bin must agree with the oracle's under its own generator (different streams, same distribution)."""
    from wesep_amd.utils.funcs import compute_fbank
    torch.manual_seed(3)
    got = compute_fbank(torch.zeros(4, 16000), dither=1.0).numpy()
    ref = FB.compute_fbank(np.zeros((4, 16000), np.float32), dither=1.0, rng=np.random.default_rng(9))
    assert got.shape == ref.shape
    # 392 frames per estimate; the lowest filters span one FFT bin (log of a chi-square(2): std 1.28), so the
    # difference of two per-bin means has std <= 0.09 -- 0.4 is > 4 sigma; the all-bin mean is far tighter
    assert np.abs(got.mean((0, 1)) - ref.mean((0, 1))).max() < 0.4
    assert abs(got.mean() - ref.mean()) < 0.03
    assert abs(got.std(1).mean() - ref.std(1).mean()) < 0.1
    # and dither is not applied when it is 0
    a = compute_fbank(torch.zeros(1, 1600), dither=0.0)
    assert float(a.std()) == 0.0


def test_argument_errors(emu):
    from wesep_amd.utils.funcs import apply_cmvn, compute_fbank
    with pytest.raises(ValueError):
        compute_fbank(torch.zeros(1, 399))
    with pytest.raises(ValueError):
        compute_fbank(torch.zeros(1, 2, 1600))
    with pytest.raises(NotImplementedError):
        compute_fbank(torch.zeros(1, 1600), num_mel_bins=23)
    with pytest.raises(NotImplementedError):
        apply_cmvn(torch.zeros(1, 8, 80), norm_var=True)


def test_no_cpu_path():
    from wesep_amd._lib import WesepHipError
    from wesep_amd.utils.funcs import apply_cmvn, compute_fbank
    with pytest.raises(WesepHipError):
        compute_fbank(torch.zeros(1, 1600))
    with pytest.raises(WesepHipError):
        apply_cmvn(torch.zeros(1, 8, 80))


class _Recorder(torch.nn.Module):
    """Stand-in separator: est = gain * mix (+ a dependence on the enrollment's mean so that gradients flow)."""

    def __init__(self):
        super().__init__()
        self.gain = torch.nn.Parameter(torch.tensor(0.5))
        self.calls = []

    def forward(self, mix, enroll):
        self.calls.append((enroll.detach().clone(), torch.is_grad_enabled()))
        return self.gain * mix + 0.0 * enroll.mean(), torch.zeros(())


def _run_ssa(prob, speaker_feat, fbank_args, steps=3):
    import random
    import wesep_amd.utils.executor as ex
    from wesep_amd.utils.executor import Executor
    from wesep_amd.utils.losses import parse_loss
    from wesep_amd.utils.schedulers import ExponentialDecrease
    random.seed(0)
    g = torch.Generator().manual_seed(2)
    wav = 0.1 * torch.randn(2, 1600, generator=g)
    batch = {"wav_mix": wav, "wav_targets": wav.clone(), "spk_embeds": torch.randn(2, 8, 80, generator=g),
             "spk_label": torch.zeros(0)}
    model = _Recorder()
    opt = torch.optim.SGD(model.parameters(), lr=0.0)
    sched = ExponentialDecrease(opt, num_epochs=1, epoch_iter=steps, initial_lr=1e-9, final_lr=1e-9, warm_up_epoch=0)
    crit = [lambda est, ref: ((est - ref) ** 2).mean(1)]
    ex_clip, ex.clip_gradients = ex.clip_gradients, (lambda model, clip: None)   # the HIP clip is not under test
    try:
        return _train(Executor, batch, steps, model, opt, crit, sched, prob, fbank_args, speaker_feat), batch
    finally:
        ex.clip_gradients = ex_clip


def _train(Executor, batch, steps, model, opt, crit, sched, prob, fbank_args, speaker_feat):
    Executor().train([batch] * steps, [model], steps, [opt], crit, [sched], scaler=None, epoch=1, enable_amp=False,
                     logger=None, device=torch.device("cpu"), se_loss_weight=([[0]], [[1.0]]),
                     SSA_enroll_prob=prob, fbank_args=fbank_args, sample_rate=16000, speaker_feat=speaker_feat)
    return model


def test_executor_ssa_second_pass(emu):
    """executor.py:89-102: with probability SSA_enroll_prob the step is a no-grad pass on the given enrollment, then
    the real pass on CMN(fbank(estimate)); otherwise one pass on the given enrollment."""
    args = dict(num_mel_bins=80, frame_length=25, frame_shift=10, dither=0.0)
    model, batch = _run_ssa(1.0, True, args)
    assert len(model.calls) == 6
    for first, second in zip(model.calls[0::2], model.calls[1::2]):
        assert first[1] is False and second[1] is True
        assert torch.equal(first[0], batch["spk_embeds"])
        est = (0.5 * batch["wav_mix"]).numpy()
        want = FB.apply_cmvn(FB.compute_fbank(est, dither=0.0))
        assert tuple(second[0].shape) == want.shape == (2, 8, 80)
        assert np.abs(second[0].numpy() - want).max() < REF_ATOL
    assert model.gain.grad is not None
    # raw-audio enrollment models (speaker_feat False): the estimate itself is the new enrollment
    model, batch = _run_ssa(1.0, False, args, steps=1)
    assert torch.equal(model.calls[1][0], 0.5 * batch["wav_mix"])
    # probability 0: the reference's plain step; in between: a Bernoulli choice per step
    model, batch = _run_ssa(0, True, args)
    assert len(model.calls) == 3 and all(torch.equal(c[0], batch["spk_embeds"]) and c[1] for c in model.calls)
    model, _ = _run_ssa(0.5, True, args, steps=40)
    assert 40 < len(model.calls) < 80


def test_executor_accepts_enable_amp_and_a_grad_scaler(emu):
    """executor.py:88,130-134: the AMP switch and the GradScaler protocol are part of the reference's Executor.train
    signature.  This path has nothing for autocast to downcast, so `enable_amp=True` must run and leave the step
    unchanged; a scaler object that is passed in is driven through scale -> unscale_ -> step -> update."""
    import wesep_amd.utils.executor as ex
    from wesep_amd.utils.executor import Executor
    from wesep_amd.utils.schedulers import ExponentialDecrease

    class Scaler:
        def __init__(self):
            self.calls = []

        def scale(self, loss):
            self.calls.append("scale")
            return loss * 8.0

        def unscale_(self, optimizer):
            self.calls.append("unscale_")
            for group in optimizer.param_groups:
                for p in group["params"]:
                    if p.grad is not None:
                        p.grad.div_(8.0)

        def step(self, optimizer):
            self.calls.append("step")
            optimizer.step()

        def update(self):
            self.calls.append("update")

    def run(enable_amp, scaler):
        torch.manual_seed(3)
        lin = torch.nn.Linear(16, 16)

        class M(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.lin = lin

            def forward(self, wav, enroll):
                return [self.lin(wav.view(-1, 16)).view_as(wav)]

        model = M()
        opt = torch.optim.SGD(model.parameters(), lr=0.1)
        sched = ExponentialDecrease(opt, num_epochs=1, epoch_iter=2, initial_lr=0.1, final_lr=0.1, warm_up_epoch=0)
        g = torch.Generator().manual_seed(4)
        wav = torch.randn(2, 160, generator=g)
        batch = {"wav_mix": wav, "wav_targets": wav.flip(1).contiguous(), "spk_embeds": torch.randn(2, 8, generator=g),
                 "spk_label": torch.zeros(0)}
        crit = [lambda est, ref: ((est - ref) ** 2).mean(1)]
        keep, ex.clip_gradients = ex.clip_gradients, (lambda model, clip: None)
        try:
            Executor().train([batch] * 2, [model], 2, [opt], crit, [sched], scaler=scaler, epoch=1, enable_amp=enable_amp,
                             logger=None, device=torch.device("cpu"), se_loss_weight=([[0]], [[1.0]]))
        finally:
            ex.clip_gradients = keep
        return torch.cat([p.detach().flatten() for p in model.parameters()])

    base = run(False, None)
    assert torch.equal(run(True, None), base)
    sc = Scaler()
    scaled = run(True, sc)
    assert sc.calls == ["scale", "unscale_", "step", "update"] * 2
    assert torch.allclose(scaled, base, rtol=1e-6, atol=1e-7)
