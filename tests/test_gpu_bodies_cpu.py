"""CPU: the bodies of the GPU tests that have not run on hardware yet (tests/test_*_gpu.py), executed on
the CPU emulations (tests/emu_dev.py entry points, tests/emu_bsrnn.py pBSRNN functions) with `cuda:0` replaced by the
CPU.  This checks the tests themselves -- shapes, fixture keys, tolerances against emulated fp32 numerics, the manual
two-pass compositions they compare with -- so that their first run on a GPU tests the kernels and not the test code.
(The native-runtime GPU tests drive a C++ library and have no emulation; their plan is covered by
tests/test_engine_cpu.py.)"""
import pytest
import torch

from tests import emu_bsrnn, emu_dev

if torch.cuda.is_available():
    pytest.skip("CPU rehearsal of GPU tests: pointless where the real ones run", allow_module_level=True)


@pytest.fixture
def emu(monkeypatch):
    import wesep_amd.utils.executor as ex
    emu_dev.install(monkeypatch)
    emu_bsrnn.install(monkeypatch)
    monkeypatch.setattr(torch.Tensor, "is_cuda", property(lambda self: True))
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    monkeypatch.setattr(ex, "clip_gradients", lambda model, clip: None)      # the HIP clip is not under test


def test_fbank_gpu_test_bodies(emu, monkeypatch, golden_dir):
    import tests.test_fbank_gpu as t
    monkeypatch.setattr(t, "_cuda", lambda: torch.device("cpu"))
    for name in sorted(t.FBANK_CASES):
        t.test_fbank_matches_reference_cpp_fixture(name, golden_dir)
    t.test_fbank_full_size_and_ragged_lengths()
    t.test_fbank_dither_statistics()
    t.test_executor_ssa_step_on_joint_model()


def test_bsrnn_multi_gpu_test_bodies(emu, monkeypatch, golden_dir):
    import tests.test_bsrnn_multi_gpu as t
    monkeypatch.setattr(t, "_cuda", lambda: torch.device("cpu"))
    for name in sorted(t.MULTI_CASES):
        t.test_bsrnn_multi_two_pass_forward_and_gradients(name, golden_dir)


def test_campplus_gpu_test_bodies(emu, monkeypatch):
    """tests/test_campplus_gpu.py (CAM++ speaker encoder) on the emulation, as written (a subset of the parametrisations)."""
    import tests.test_campplus_gpu as t
    monkeypatch.setattr(t, "_cuda", lambda: torch.device("cpu"))
    for args in ((3, 230, 32, 100), (2, 100, 8, 100), (2, 301, 12, 7)):
        t.test_segment_kernels_match_torch(*args)
    for args in ((320, 128, 5, 1, 2, False), (128, 32, 3, 2, 1, False), (64, 32, 1, 1, 1, True), (32, 16, 3, 1, 2, True)):
        t.test_conv1d_matches_torch(*args)
    t.test_bn_act_and_mel_strided_conv_block_match_torch()
    t.test_campplus_matches_oracle(monkeypatch)
    t.test_bsrnn_joint_training_with_campplus_runs_and_matches_oracle()


def test_convtasnet_variant_gpu_test_bodies(emu, monkeypatch, golden_dir):
    """The causal depthwise convolution and reference-fixture variant tests of tests/test_convtasnet_gpu.py."""
    import tests.test_convtasnet_gpu as t
    monkeypatch.setattr(t, "_cuda", lambda: torch.device("cpu"))
    for dil, P in ((1, 3), (4, 3), (2, 5)):
        t.test_causal_dwconv_fwd_bwd(dil, P)
    for name in ("convtasnet_plain_skip_r2_t1600", "convtasnet_multi_bn_skip_r4_t1600"):
        t.test_variants_match_reference_fixture(name, golden_dir)
    t.test_joint_training_with_a_wespeaker_encoder_on_fbank(golden_dir)


def test_resnet_pooling_gpu_test_bodies(emu, monkeypatch):
    import tests.test_resnet_gpu as t
    monkeypatch.setattr(t, "_cuda", lambda: torch.device("cpu"))
    for pooling in ("TAP", "TSDP", "ASTP"):
        t.test_resnet_pooling_variants_match_oracle(pooling)


def test_pair_bptt_gpu_test_body(emu, monkeypatch):
    """tests/test_kernels_gpu.py::test_lstm_pair_bwd_vs_torch on the blocked-layout emulation (small cases)."""
    import tests.test_kernels_gpu as t
    from tests import emu_blk
    import wesep_amd.dev as dev
    emu_blk.install(monkeypatch)
    monkeypatch.setattr(dev, "bls_unpack", lambda x: x)      # the emulation keeps plain fp32 where the kernels keep BLS
    monkeypatch.setattr(t, "_cuda", lambda: torch.device("cpu"))
    t.test_lstm_pair_bwd_vs_torch("time", (3, 7, 37))
    t.test_lstm_pair_bwd_vs_torch("band", (4, 9, 16))


def test_engine_separator_gpu_test_bodies(emu, monkeypatch, tmp_path):
    """tests/test_zzz_engine_separators_gpu.py has not run on hardware: its bodies -- model construction, parameter
    randomisation, export, the Python forward of every variant -- run here on the emulations, with the engine in its dry
    run (argument validation of every launch; it computes nothing, so the comparison itself is stubbed out)."""
    import tests.test_engine_gpu as tg
    import tests.test_zzz_engine_separators_gpu as z
    from tests import emu_blk
    from wesep_amd import engine as E
    emu_blk.install(monkeypatch)
    monkeypatch.setattr(tg, "_cuda", lambda: torch.device("cpu"))
    monkeypatch.setattr(tg, "rel", lambda a, b: 0.0)
    real = E.Engine

    class DryEngine(real):
        def __init__(self, path):
            super().__init__(path, dry_run=True)
    monkeypatch.setattr(E, "Engine", DryEngine)
    for i, v in enumerate(("joint-resnet18-multiply", "fixed-additive", "fixed-film-causal")):
        d = tmp_path / f"d{i}"
        d.mkdir()
        z.test_dpccn_engine_matches_python_model(d, v)
    for i, v in enumerate(("joint-resnet18-multiply", "fixed-additive", "fixed-film-hidden64")):
        d = tmp_path / f"g{i}"
        d.mkdir()
        z.test_tfgridnet_engine_matches_python_model(d, v)
