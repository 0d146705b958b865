"""CPU: argument-contract dry run of whole host paths into the real libwesep_hip.so (tests/abi_dryrun.py).

Each case runs the product's forward + backward host code on CPU tensors; every `ws_*` call must pass its entry
point's argument validation (shapes, leading dimensions, alignment flags, split counts) and fail only at the launch,
because this machine has no GPU.  This catches the "bad args" class of defects (a width that is not a multiple of
4, a split count over a limit, a leading dimension smaller than the row) for configurations the `-m gpu` tests do
not reach -- the shipped recipe sizes in particular -- without computing anything."""
import os
import random

import pytest
import torch

from tests import abi_dryrun

if torch.cuda.is_available():        # on the GPU box the launches would succeed: covered by the -m gpu tests
    pytest.skip("argument-contract dry run is for GPU-less machines", allow_module_level=True)

# entry points that size themselves from the device and are expected to refuse a machine without CUs
DEVICE_DEPENDENT = ("ws_lstm_fwd_cluster", "ws_lstm_fwd_cluster2", "ws_lstm_bwd_cluster", "ws_lstm_bwd_pair")

SPK = dict(joint_training=True, spk_feat=True,
           spk_args=dict(feat_dim=80, embed_dim=256, pooling_func="TSTP", two_emb_layer=False))


def _check(calls, at_least):
    kept = [c for c in calls if not (c[0] in DEVICE_DEPENDENT and "CUs" in c[2])]
    abi_dryrun.assert_contracts_hold(kept, at_least)
    return {w for w, _, _ in kept}


def _fwd_bwd(model, *inputs):
    model.train()
    out = model(*inputs)
    outs = out if isinstance(out, (list, tuple)) else [out]
    sum(o.sum() for o in outs if o.dim() > 0).backward()
    assert all(p.grad is not None for p in model.parameters())
    return outs


@pytest.mark.parametrize("R,T", [(2, 16000), (6, 24000), (2, 12345)])
@pytest.mark.parametrize("fuse,multi", [("multiply", False), ("FiLM", True), ("concat", True), ("additive", False)])
def test_bsrnn_contracts(monkeypatch, R, T, fuse, multi):
    from wesep_amd.models import get_model
    calls = abi_dryrun.install(monkeypatch)
    model = get_model("BSRNN")(num_repeat=2, spk_fuse_type=fuse, multi_fuse=multi, joint_training=False)
    est, _ = _fwd_bwd(model, torch.randn(R, T), torch.randn(R, 256))
    assert tuple(est.shape) == (R, T)
    used = _check(calls, 50)
    assert {"ws_stft_bandsplit", "ws_gemm_p2b", "ws_gemm_tnb", "ws_mask_istft_bwd"} <= used


def test_bsrnn_headline_row_count_contracts(monkeypatch):
    """R = 32 rows (BASELINE configs[1] per-GPU batch) at a short length: 1024 time-view / 32*Tf band-view sequences
    take the cluster and fused band-view branches of the host code."""
    from wesep_amd.models import get_model
    calls = abi_dryrun.install(monkeypatch)
    model = get_model("BSRNN")(num_repeat=1, spk_fuse_type="multiply", multi_fuse=False, joint_training=False)
    _fwd_bwd(model, torch.randn(32, 8192), torch.randn(32, 256))
    used = _check(calls, 30)
    assert "ws_lstm_fwd_fused" in used or "ws_lstm_fwd" in used


@pytest.mark.parametrize("spk_model", ["ResNet18", "ResNet34"])
def test_joint_bsrnn_and_ssa_step_contracts(monkeypatch, spk_model):
    """The jointly trained speaker encoder on fbank enrollment, then one Executor step with the SSA second pass and
    the fused clip + Adam."""
    from wesep_amd.models import get_model
    from wesep_amd.optim import FusedClipAdam
    from wesep_amd.utils.executor import Executor
    from wesep_amd.utils.losses import parse_loss
    from wesep_amd.utils.schedulers import ExponentialDecrease
    calls = abi_dryrun.install(monkeypatch)
    model = get_model("BSRNN")(num_repeat=1, spk_fuse_type="multiply", multi_fuse=False, use_spk_transform=False,
                               spk_model=spk_model, multi_task=True, spksInTrain=251, **SPK)
    opt = FusedClipAdam(model.parameters(), lr=1e-3, weight_decay=1e-4)
    sched = ExponentialDecrease(opt, num_epochs=1, epoch_iter=2, initial_lr=1e-3, final_lr=1e-4, warm_up_epoch=0)
    batch = {"wav_mix": torch.randn(2, 16000), "wav_targets": torch.randn(2, 16000),
             "spk_embeds": torch.randn(2, 98, 80), "spk_label": torch.tensor([3, 250])}
    random.seed(0)
    Executor().train([batch, batch], [model], 2, [opt], parse_loss(["SISDR", "CE"]), [sched], scaler=None, epoch=1,
                     enable_amp=False, logger=None, device=torch.device("cpu"),
                     se_loss_weight=([[0], [1]], [[1.0], [0.1]]), multi_task=True, SSA_enroll_prob=1.0,
                     fbank_args=dict(num_mel_bins=80, frame_length=25, frame_shift=10, dither=1.0),
                     sample_rate=16000, speaker_feat=True)
    used = _check(calls, 200)
    assert {"ws_power_spec", "ws_log_eps", "ws_chan_sums", "ws_bn_stats", "ws_tstp_fwd", "ws_cross_entropy",
            "ws_sisdr_fwd", "ws_grad_norms", "ws_clip_adam_step"} <= used


def test_bsrnn_multi_recipe_contracts(monkeypatch):
    """examples/librimix/tse/v2/confs/bsrnn_multi_optim.yaml model_args (num_repeat reduced to 2): both passes, the
    raw-audio front-end twice, ResNet34, and the recipe's two-output loss through one Executor step."""
    from wesep_amd.models import get_model
    from wesep_amd.optim import FusedClipAdam
    from wesep_amd.utils.executor import Executor
    from wesep_amd.utils.losses import parse_loss
    from wesep_amd.utils.schedulers import ExponentialDecrease
    calls = abi_dryrun.install(monkeypatch)
    model = get_model("BSRNN_Multi")(sr=16000, win=512, stride=128, feature_dim=128, num_repeat=2,
                                     spk_fuse_type="multiply", use_spk_transform=False, multi_fuse=False,
                                     joint_training=True, spk_model="ResNet34", spk_model_init=False,
                                     spk_args=dict(feat_dim=80, embed_dim=256, pooling_func="TSTP",
                                                   two_emb_layer=False),
                                     spk_emb_dim=256, spk_model_freeze=False, spk_feat=False, feat_type="consistent",
                                     multi_task=False, spksInTrain=251)
    opt = FusedClipAdam(model.parameters(), lr=1e-3, weight_decay=1e-4)
    sched = ExponentialDecrease(opt, num_epochs=1, epoch_iter=1, initial_lr=1e-3, final_lr=1e-4, warm_up_epoch=0)
    batch = {"wav_mix": torch.randn(4, 48000), "wav_targets": torch.randn(4, 48000),
             "spk_embeds": torch.randn(4, 40000), "spk_label": torch.zeros(0)}
    Executor().train([batch], [model], 1, [opt], parse_loss("SISDR"), [sched], scaler=None, epoch=1, enable_amp=False,
                     logger=None, device=torch.device("cpu"), se_loss_weight=([[0, 1]], [[0.4, 0.6]]),
                     speaker_feat=False)
    used = _check(calls, 400)
    assert {"ws_preemph_pad", "ws_power_spec", "ws_sisdr_fwd", "ws_clip_adam_step"} <= used
    assert all(p.grad is not None for p in model.parameters())


def test_fbank_contracts(monkeypatch):
    from wesep_amd.utils.funcs import apply_cmvn, compute_fbank
    calls = abi_dryrun.install(monkeypatch)
    for R, T, sr, nb in ((32, 64000, 16000, 80), (2, 1203, 16000, 80), (3, 400, 16000, 40), (2, 4000, 8000, 40)):
        f = compute_fbank(torch.zeros(R, T), num_mel_bins=nb, dither=1.0, sample_rate=sr)
        win, shift = sr // 40, sr // 100
        assert tuple(apply_cmvn(f).shape) == (R, 1 + (T - win) // shift, nb)
    _check(calls, 40)


@pytest.mark.parametrize("kw,R,T,Te", [
    (dict(N=32, L=20, B=32, H=64, P=3, X=3, R=2, joint_training=False), 2, 1600, 0),
    (dict(N=512, L=20, B=256, H=512, P=3, X=8, R=4, joint_training=False), 2, 8000, 0),       # confs/ size
    (dict(N=16, L=16, B=24, H=40, P=3, X=2, R=1, norm="cLN", joint_training=False), 4, 2000, 0),
    (dict(N=256, L=20, B=256, H=512, P=3, X=8, R=4, joint_training=True, multi_task=True, spksInTrain=251), 4, 8000,
     24000),                                                                                  # SpEx+ recipe
])
def test_convtasnet_contracts(monkeypatch, kw, R, T, Te):
    from wesep_amd.models import get_model
    calls = abi_dryrun.install(monkeypatch)
    model = get_model("ConvTasNet")(**kw)
    enroll = torch.randn(R, Te) if Te else torch.randn(R, 256)
    outs = _fwd_bwd(model, torch.randn(R, T), enroll)
    assert tuple(outs[0].shape) == (R, T)
    _check(calls, 40)


def test_dpccn_recipe_contracts(monkeypatch):
    """examples/librimix/tse/v2/confs/dpccn.yaml model_args with the ResNet34 encoder."""
    from wesep_amd.models import get_model
    calls = abi_dryrun.install(monkeypatch)
    model = get_model("DPCCN")(win=512, stride=128, feature_dim=257, tcn_blocks=10, tcn_layers=2, causal=False,
                               spk_fuse_type="multiply", use_spk_transform=False, spk_model="ResNet34", **SPK)
    est, _ = _fwd_bwd(model, torch.randn(2, 16384), torch.randn(2, 100, 80))
    assert tuple(est.shape) == (2, 16384)
    _check(calls, 300)


@pytest.mark.parametrize("ks,hs,T", [(1, 1, 8000), (4, 1, 6400), (4, 2, 6400)])
def test_tfgridnet_recipe_contracts(monkeypatch, ks, hs, T):
    """examples/librimix/tse/v2/confs/tfgridnet.yaml model_args (n_layers reduced to 2) with the ResNet34 encoder,
    at the recipe's emb_ks/emb_hs = 1 and at unfold sizes > 1."""
    from wesep_amd.models import get_model
    calls = abi_dryrun.install(monkeypatch)
    model = get_model("TFGridNet")(n_fft=128, stride=64, n_layers=2, lstm_hidden_units=192, attn_n_head=4,
                                   attn_approx_qk_dim=512, emb_dim=128, emb_ks=ks, emb_hs=hs,
                                   use_spk_transform=False, spk_fuse_type="multiply", spk_model="ResNet34", **SPK)
    est, _ = _fwd_bwd(model, torch.randn(2, T), torch.randn(2, 100, 80))
    assert tuple(est.shape) == (2, T)
    _check(calls, 200)


def test_train_entry_with_real_models_contracts(monkeypatch, tmp_path):
    """wesep_amd.bin.train end to end (config -> synthetic batches -> model -> Executor epochs -> checkpoints) with the
    real pBSRNN module trees; every launch of the two epochs passes its argument validation."""
    import wesep_amd.bin.train as T
    calls = abi_dryrun.install(monkeypatch)
    base = {"exp_dir": str(tmp_path / "exp"), "seed": 1, "num_epochs": 2, "num_avg": 2, "save_epoch_interval": 1,
            "log_batch_interval": 1, "clip_grad": 5.0, "enable_amp": False, "loss": "SISDR",
            "loss_args": {"loss_posi": [[0]], "loss_weight": [[1.0]]}, "model": {"tse_model": "BSRNN"},
            "model_init": {"tse_model": None},
            "model_args": {"tse_model": dict(num_repeat=1, spk_fuse_type="multiply", multi_fuse=False,
                                             use_spk_transform=False, spk_model="ResNet18", **SPK)},
            "optimizer": {"tse_model": "Adam"}, "optimizer_args": {"tse_model": {"lr": 1e-3, "weight_decay": 1e-4}},
            "scheduler": {"tse_model": "ExponentialDecrease"},
            "scheduler_args": {"tse_model": {"initial_lr": 1e-3, "final_lr": 2.5e-5, "warm_up_epoch": 0}},
            "dataloader_args": {"batch_size": 1}, "dataset_args": {"chunk_len": 8000, "resample_rate": 16000}}
    assert T.train(base, 2).step == 4
    assert sorted(os.listdir(tmp_path / "exp" / "models"))[:2] == ["checkpoint_1.pt", "checkpoint_2.pt"]
    multi = {**base, "exp_dir": str(tmp_path / "exp2"), "model": {"tse_model": "BSRNN_Multi"},
             "loss_args": {"loss_posi": [[0, 1]], "loss_weight": [[0.4, 0.6]]}}
    multi["model_args"] = {"tse_model": {**base["model_args"]["tse_model"], "spk_feat": False, "feat_type": "consistent"}}
    assert T.train(multi, 1).step == 2
    _check(calls, 500)


@pytest.mark.parametrize("B,T", [(2, 8000), (8, 6400)])
def test_tfgridnet_blocked_recurrence_path_contracts(monkeypatch, B, T):
    """WESEP_TFGRID_BLOCKED=1: the recipe geometry (emb_dim 128, emb_ks = emb_hs = 1) on the pBSRNN blocked-layout
    recurrence machinery, with the sequence count zero-padded to the cluster kernels' multiple of 64."""
    from wesep_amd.models import get_model
    monkeypatch.setenv("WESEP_TFGRID_BLOCKED", "1")
    calls = abi_dryrun.install(monkeypatch)
    model = get_model("TFGridNet")(n_fft=128, stride=64, n_layers=2, lstm_hidden_units=192, attn_n_head=4,
                                   attn_approx_qk_dim=512, emb_dim=128, emb_ks=1, emb_hs=1, use_spk_transform=False,
                                   spk_fuse_type="multiply", joint_training=False)
    est, _ = _fwd_bwd(model, torch.randn(B, T), torch.randn(B, 256))
    assert tuple(est.shape) == (B, T)
    used = _check(calls, 100)
    assert {"ws_gemm_p2b", "ws_gemm_b2p", "ws_gemm_tnb"} <= used
    # few long sequences: the pair BPTT (it sizes itself from the device, so its CU check refuses this machine)
    assert "ws_lstm_bwd" in used or any(w == "ws_lstm_bwd_pair" for w, _, _ in calls)


@pytest.mark.parametrize("spk_model,spk_args,E", [
    ("ECAPA_TDNN_GLOB_c512", dict(feat_dim=80, embed_dim=192, pooling_func="ASTP"), 192),       # bsrnn.yaml:66-71
    ("ECAPA_TDNN_c1024", dict(feat_dim=80, embed_dim=256, pooling_func="ASTP", emb_bn=True), 256),
    ("CAMPPlus", dict(feat_dim=80, embed_dim=512, pooling_func="TSTP"), 512),                    # bsrnn.yaml:72-74
    ("ResNet50", dict(feat_dim=80, embed_dim=256, pooling_func="ASTP", two_emb_layer=True), 256)])
def test_joint_bsrnn_with_the_other_recipe_encoders_contracts(monkeypatch, spk_model, spk_args, E):
    """The recipe's alternative speaker encoders at their full size (ECAPA-TDNN 6.2 M, CAM++ 7.2 M parameters, a
    Bottleneck ResNet with attentive pooling and two embedding layers) inside a jointly trained pBSRNN: every launch of
    forward + backward passes the real library's argument validation (250 enrollment frames: three mask segments after
    CAM++'s stride-2 layer)."""
    from wesep_amd.models import get_model
    calls = abi_dryrun.install(monkeypatch)
    model = get_model("BSRNN")(num_repeat=1, spk_fuse_type="multiply", multi_fuse=False, use_spk_transform=False,
                               joint_training=True, spk_feat=True, spk_emb_dim=E, spk_model=spk_model, spk_args=spk_args)
    est, _ = _fwd_bwd(model, torch.randn(4, 8000), torch.randn(4, 250, 80))
    assert tuple(est.shape) == (4, 8000)
    seen = _check(calls, 300)
    if spk_model == "CAMPPlus":
        assert {"ws_seg_sums", "ws_seg_scale"} <= seen
    if "ECAPA" in spk_model:
        assert {"ws_astp_fwd", "ws_astp_bwd"} <= seen


@pytest.mark.parametrize("kw", [
    dict(N=512, L=16, B=128, H=512, P=3, X=8, R=3, encoder_type=None, decoder_type=None, skip_con=True),   # classic sizes
    dict(N=256, L=20, B=256, H=512, P=3, X=4, R=2, encoder_type="Deep", decoder_type="Deep", causal=True, norm="cLN",
         activate="sigmoid", spk_fuse_type="FiLM"),
    dict(N=256, L=20, B=256, H=512, P=3, X=4, R=2, norm="BN", skip_con=True, causal=True)])
def test_convtasnet_variant_contracts(monkeypatch, kw):
    """The non-recipe halves of the ConvTasNet constructor (plain / Deep ends, skip, causal, BN) at production widths."""
    from wesep_amd.models import get_model
    calls = abi_dryrun.install(monkeypatch)
    model = get_model("ConvTasNet")(joint_training=False, use_spk_transform=False, **kw)
    model.train()
    out = model(torch.randn(2, 8000), torch.randn(2, 256))
    outs = out if isinstance(out, (list, tuple)) else [out]
    sum(o.sum() for o in outs).backward()
    unused = {n for n, p in model.named_parameters() if p.grad is None}
    assert all(".Output." in n for n in unused), unused       # skip_con: only the last blocks' residual Output convs
    seen = _check(calls, 100)
    if kw.get("causal"):
        assert "ws_dwconv_ex_fwd" in seen


def test_tfgridnet_multi_source_and_dpccn_causal_contracts(monkeypatch):
    from wesep_amd.models import get_model
    calls = abi_dryrun.install(monkeypatch)
    model = get_model("TFGridNet")(n_fft=128, stride=64, n_layers=1, lstm_hidden_units=192, attn_n_head=4,
                                   attn_approx_qk_dim=512, emb_dim=48, emb_ks=4, emb_hs=1, n_srcs=3, n_imics=4,
                                   use_spk_transform=False, spk_fuse_type="multiply", joint_training=False)
    est, _ = _fwd_bwd(model, torch.randn(2, 6400, 4), torch.randn(2, 256))
    assert tuple(est.shape) == (2, 3, 6400)
    _check(calls, 100)
    calls.clear()
    model = get_model("DPCCN")(win=512, stride=128, feature_dim=257, tcn_blocks=10, tcn_layers=2, causal=True,
                               spk_fuse_type="FiLM", use_spk_transform=False, joint_training=False)
    est, _ = _fwd_bwd(model, torch.randn(2, 16384), torch.randn(2, 256))
    assert tuple(est.shape) == (2, 16384)
    _check(calls, 300)
