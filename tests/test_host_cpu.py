"""CPU: the C-ABI library loads and exports every declared symbol, ctypes mirrors the C structs,
the Python surface mirrors the reference (state_dict keys, factory, schedulers, checkpoints),
and the product path refuses to run without the GPU (no fallback)."""
import ctypes
import os
import re
import subprocess
import sys
import tempfile

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "wesep_hip.h")


@pytest.fixture(scope="session")
def built_lib():
    from wesep_amd.build import build
    build(verbose=False)
    from wesep_amd import _lib
    return _lib


def test_library_exports_every_declared_symbol(built_lib):
    src = open(HEADER).read()
    declared = set(re.findall(r"^(?:int|const char\*)\s+(ws_\w+)\s*\(", src, flags=re.M))
    assert declared, "header parse failed"
    assert declared == set(built_lib.EXPORTED_SYMBOLS)
    lib = built_lib.lib()
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.ws_abi_version() == built_lib.ABI_VERSION == 20


def test_built_library_has_no_packed_fp32_arithmetic(built_lib):
    """The fence of profiles/r03_kernel_race.md: packed FP32 instructions with an operand selection compute wrong low
    halves beside gemm_b2p on the MI355X, so the library is built without packed FP32 (build.py NO_PACKED_FP32).
    Disassembles the gfx950 code objects of the built .so; no GPU needed."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_no_packed_fp32.py"), built_lib.LIB_PATH],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert "v_mfma_f32_32x32x16_bf16" in r.stdout                  # the walk did see the kernels


def test_ctypes_structs_match_c_layout(built_lib):
    prog = r'''
#include <stdio.h>
#include <stddef.h>
#include "wesep_hip.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(ws_gemm_nt_args), sizeof(ws_gemm_tn_args),
         sizeof(ws_groups_geom), sizeof(ws_lstm_args), sizeof(ws_bands), sizeof(ws_group_nt),
         sizeof(ws_group_tn), sizeof(ws_tensor_ref), offsetof(ws_gemm_tn_args, g_div));
  printf("%zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(ws_gemm_p2b_args), sizeof(ws_gemm_b2p_args),
         sizeof(ws_gemm_tnb_args), sizeof(ws_lstm_cluster_args), sizeof(ws_lstm_fused_args), sizeof(ws_seqmap),
         offsetof(ws_lstm_args, run_if), offsetof(ws_gemm_p2b_args, run_if));
  printf("%zu %zu\n", sizeof(ws_lstm_pair_args), offsetof(ws_lstm_pair_args, nseq));
  return 0;
}'''
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "t.c")
        open(c, "w").write(prog)
        exe = os.path.join(td, "t")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        sizes = [int(x) for x in subprocess.check_output([exe]).split()]
    L = built_lib
    assert sizes[:8] == [ctypes.sizeof(L.GemmNTArgs), ctypes.sizeof(L.GemmTNArgs), ctypes.sizeof(L.GroupsGeom),
                         ctypes.sizeof(L.LstmArgs), ctypes.sizeof(L.Bands), L.GROUP_NT_DTYPE.itemsize,
                         L.GROUP_TN_DTYPE.itemsize, L.TENSOR_REF_DTYPE.itemsize]
    assert sizes[8] == L.GemmTNArgs.g_div.offset
    assert sizes[17:] == [ctypes.sizeof(L.LstmPairArgs), L.LstmPairArgs.nseq.offset]
    assert sizes[9:17] == [ctypes.sizeof(L.GemmP2BArgs), ctypes.sizeof(L.GemmB2PArgs), ctypes.sizeof(L.GemmTNBArgs),
                         ctypes.sizeof(L.LstmClusterArgs), ctypes.sizeof(L.LstmFusedArgs), ctypes.sizeof(L.SeqMapC),
                         L.LstmArgs.run_if.offset, L.GemmP2BArgs.run_if.offset]


@pytest.mark.parametrize("kw", [
    dict(num_repeat=1, spk_fuse_type="multiply", multi_fuse=False),
    dict(num_repeat=2, spk_fuse_type="FiLM", multi_fuse=True),
    dict(num_repeat=1, spk_fuse_type="additive", multi_fuse=False, use_spk_transform=True),
    dict(num_repeat=1, spk_fuse_type="concat", multi_fuse=True),
    dict(num_repeat=6, spk_fuse_type="multiply", multi_fuse=False),
])
def test_state_dict_keys_match_reference(kw):
    from oracle import bsrnn_oracle as O
    from wesep_amd.models import get_model
    cfg = O.BSRNNConfig(**kw)
    m = get_model("BSRNN")(joint_training=False, use_spk_transform=cfg.use_spk_transform,
                           **{k: v for k, v in kw.items() if k != "use_spk_transform"})
    shapes = O.param_shapes(cfg)     # pinned to the real reference by oracle/make_golden.py
    sd = m.state_dict()
    assert list(sd.keys()) == list(shapes.keys())
    assert all(tuple(sd[k].shape) == shapes[k] for k in sd)
    assert not list(m.buffers())
    if kw["num_repeat"] == 6:        # 21.435 M parameters, 530 tensors
        assert len(sd) == 530
        assert abs(sum(v.numel() for v in sd.values()) / 1e6 - 21.435) < 1e-3


def test_film_zero_init_and_factory_errors():
    from wesep_amd.models import get_model
    m = get_model("BSRNN")(num_repeat=1, spk_fuse_type="FiLM", multi_fuse=True, joint_training=False)
    fc = m.separator.separation[0].fc
    assert float(fc.gamma_fcs[0].weight.abs().sum()) == 0 and float(fc.beta_fcs[0].bias.abs().sum()) == 0
    with pytest.raises(NotImplementedError):
        get_model("BSRNN")(joint_training=True)
    with pytest.raises(NotImplementedError):
        get_model("TFGridNet")(window="hamming", joint_training=False)     # only the hann window is built
    with pytest.raises(NotImplementedError):
        get_model("BSRNN_Feats")
    with pytest.raises(NotImplementedError):                        # the self-enrollment pass needs raw-audio joint training
        get_model("BSRNN_Multi")(num_repeat=1, joint_training=False)


def test_no_cpu_fallback():
    from wesep_amd._lib import WesepHipError
    from wesep_amd.models import get_model
    from wesep_amd.utils.losses import parse_loss
    m = get_model("BSRNN")(num_repeat=1, spk_fuse_type="multiply", multi_fuse=False, joint_training=False,
                           use_spk_transform=False)
    with pytest.raises(WesepHipError):
        m(torch.randn(2, 4000), torch.randn(2, 256))
    with pytest.raises(WesepHipError):
        parse_loss("SISDR")[0](torch.randn(2, 100), torch.randn(2, 100))
    with pytest.raises(NotImplementedError):
        parse_loss("STFT")


def test_product_package_never_imports_oracle():
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "wesep_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M):
                    bad.append(f)
    assert not bad, bad


def test_schedulers_match_oracle_formula():
    from oracle import bsrnn_oracle as O
    from wesep_amd.utils.schedulers import ExponentialDecrease
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=0.1)
    s = ExponentialDecrease(opt, num_epochs=150, epoch_iter=100, initial_lr=1e-3, final_lr=2.5e-5,
                            warm_up_epoch=0, scale_ratio=1.0)
    for it in (0, 1, 777, 14999):
        s.step(it)
        assert abs(opt.param_groups[0]["lr"] - O.exponential_decrease_lr(it, 15000, 1e-3, 2.5e-5)) < 1e-15
    s2 = ExponentialDecrease(opt, 10, 10, 1e-3, 1e-4, warm_up_epoch=2, scale_ratio=4.0)
    s2.step(5)
    assert abs(opt.param_groups[0]["lr"] - O.exponential_decrease_lr(5, 100, 1e-3, 1e-4, 20, 4.0)) < 1e-15
    sd = s2.state_dict()
    assert "optimizer" not in sd and sd["current_iter"] == 6


def test_checkpoint_roundtrip_reference_format(tmp_path):
    from wesep_amd.models import get_model
    from wesep_amd.optim import FusedClipAdam
    from wesep_amd.utils.checkpoint import load_checkpoint, load_pretrained_model, save_checkpoint
    from wesep_amd.utils.schedulers import ExponentialDecrease
    mk = lambda: get_model("BSRNN")(num_repeat=1, spk_fuse_type="multiply", multi_fuse=False,
                                    joint_training=False, use_spk_transform=False)
    m1, m2 = mk(), mk()
    o1 = FusedClipAdam(m1.parameters(), lr=1e-3, weight_decay=1e-4)
    s1 = ExponentialDecrease(o1, 2, 10, 1e-3, 1e-4)
    s1.step(3)
    path = str(tmp_path / "model_1.pt")
    save_checkpoint([m1], [o1], [s1], None, path)
    st = torch.load(path, map_location="cpu")
    assert set(st) == {"models", "optimizers", "schedulers", "scaler"} and len(st["models"]) == 1
    o2 = FusedClipAdam(m2.parameters(), lr=5e-4)
    s2 = ExponentialDecrease(o2, 2, 10, 1e-3, 1e-4)
    load_checkpoint([m2], [o2], [s2], None, path)
    assert all(torch.equal(a, b) for a, b in zip(m1.state_dict().values(), m2.state_dict().values()))
    assert s2.current_iter == 4
    load_pretrained_model(mk(), path)


def test_tn_splits_cover_rows():
    from wesep_amd.dev import tn_splits
    for M in (1, 31, 32, 33, 2048, 2049, 4100, 513024, 16032):
        n, rps = tn_splits(M)
        assert n >= 1 and rps % 32 == 0 and n * rps >= M and (n - 1) * rps < M


def test_band_plan_offsets():
    from oracle import bsrnn_oracle as O
    from wesep_amd.functional import BandPlan
    bw = O.band_widths(16000, 512)
    plan = BandPlan(bw, 128, torch.device("cpu"))
    assert plan.K == 32 and sum(bw) == 257
    assert int(plan.bn_woff[-1]) == 128 * 2 * 257
    assert int(plan.m_w3off[-1]) == 4 * 257 * 512 and int(plan.m_b3off[-1]) == 4 * 257
    assert plan.bands.band_of_bin.tolist()[:4] == [0, 0, 0, 1]


def test_product_and_oracle_synthetic_rows_agree():
    from oracle.bsrnn_oracle import synth_batch as a
    from wesep_amd.utils.synthetic import synth_batch as b
    for x, y in zip(a(4, 1000, 7), b(4, 1000, 7)):
        assert torch.equal(x, y)


def test_score_matches_training_loss_definition():
    """cal_SISNR (score.py:7-21) against the oracle's SI-SDR on well-conditioned signals: the two eps
    placements agree to 1e-3 dB, and SI-SNRi is the difference to the unprocessed mixture."""
    import numpy as np
    import torch
    from oracle import bsrnn_oracle as O
    from wesep_amd.utils.score import cal_SISNR, cal_SISNRi
    g = np.random.default_rng(3)
    ref = g.standard_normal(16000).astype(np.float32)
    for snr in (-5.0, 0.0, 12.0, 30.0):
        est = ref + g.standard_normal(16000).astype(np.float32) * 10 ** (-snr / 20)
        want = -float(O.sisdr_loss(torch.from_numpy(est)[None], torch.from_numpy(ref)[None]))
        assert abs(cal_SISNR(est, ref) - want) < 1e-3
    mix = ref + g.standard_normal(16000).astype(np.float32)
    s, si = cal_SISNRi(est, ref, mix)
    assert abs(si - (s - cal_SISNR(mix, ref))) < 1e-12


def test_convtasnet_state_dict_keys_match_reference():
    """Module tree / parameter names of wesep.models.convtasnet.ConvTasNet (fixed-embedding SpEx+ configuration)."""
    from oracle import convtasnet_oracle as CT
    from wesep_amd.models import get_model
    for kw in (dict(N=32, L=20, B=32, H=64, P=3, X=3, R=2),
               dict(N=16, L=20, B=24, H=40, P=3, X=2, R=1, norm="cLN", use_spk_transform=True)):
        cfg = CT.ConvTasNetConfig(**kw)
        model = get_model("ConvTasNet")(**kw, joint_training=False,
                                        **({} if "use_spk_transform" in kw else {"use_spk_transform": False}))
        shapes = CT.param_shapes(cfg)
        sd = model.state_dict()
        assert list(sd.keys()) == list(shapes.keys())
        assert all(tuple(sd[k].shape) == shapes[k] for k in sd)
    # SpEx+ joint mode (reference default joint_training=True, spk_feat=False): shared encoder + ResNet4SpExplus
    kw = dict(N=256, L=20, B=32, H=48, P=3, X=2, R=2, joint_training=True, multi_task=True, spksInTrain=11)
    model = get_model("ConvTasNet")(**kw, use_spk_transform=False)
    shapes = CT.param_shapes(CT.ConvTasNetConfig(**kw))
    sd = model.state_dict()
    assert list(sd.keys()) == list(shapes.keys())
    assert all(tuple(sd[k].shape) == shapes[k] for k in sd)
    with pytest.raises(NotImplementedError):
        get_model("ConvTasNet")(spk_feat=True)    # wespeaker encoder on fbank features: SURVEY 8 row a12


def test_device_prefetcher_passthrough_on_cpu():
    """Datapipe hand-off: order, dtype conversion and pass-through of non-tensor entries."""
    import torch
    from wesep_amd.utils.prefetch import DevicePrefetcher
    batches = [{"wav_mix": torch.full((2, 5), float(i), dtype=torch.float64), "wav_targets": torch.zeros(2, 5),
                "spk_embeds": torch.ones(2, 3), "spk_label": torch.tensor([i, i + 1]), "key": [f"u{i}"]}
               for i in range(4)]
    got = list(DevicePrefetcher(batches, "cpu"))
    assert len(got) == 4 and len(DevicePrefetcher(batches, "cpu")) == 4
    for i, b in enumerate(got):
        assert b["wav_mix"].dtype == torch.float32 and float(b["wav_mix"][0, 0]) == float(i)
        assert b["spk_label"].dtype == torch.int64 and b["key"] == [f"u{i}"]
    assert list(DevicePrefetcher([], "cpu")) == []


def test_average_model_tool(tmp_path):
    """wesep/bin/average_model.py semantics: last --num numbered checkpoints, true division, {"models": [sd]} out;
    full checkpoints and bare state dicts both accepted; averaged / final / latest files are left out."""
    import subprocess
    import sys
    from wesep_amd.bin.average_model import average_checkpoints, select_checkpoints
    sds = []
    for e in (3, 9, 10, 11):
        sd = {"w": torch.full((2, 3), float(e)), "bn.num_batches_tracked": torch.tensor(e)}
        sds.append(sd)
        torch.save({"models": [sd], "optimizers": [], "schedulers": []} if e != 10 else sd,
                   tmp_path / f"checkpoint_{e}.pt")
    for name in ("avg_model.pt", "final_model.pt", "latest.pt", "checkpoint_avg.pt"):
        torch.save({"models": [sds[0]]}, tmp_path / name)
    picked = select_checkpoints(str(tmp_path), num=2)
    assert [os.path.basename(p) for p in picked] == ["checkpoint_10.pt", "checkpoint_11.pt"]
    assert [os.path.basename(p) for p in select_checkpoints(str(tmp_path), num=5, max_epoch=9)] == \
        ["checkpoint_3.pt", "checkpoint_9.pt"]
    avg = average_checkpoints(picked)
    assert torch.equal(avg["w"], torch.full((2, 3), 10.5)) and float(avg["bn.num_batches_tracked"]) == 10.5
    dst = tmp_path / "out.pt"
    r = subprocess.run([sys.executable, "-m", "wesep_amd.bin.average_model", "--dst_model", str(dst), "--src_path",
                        str(tmp_path), "--mode", "epochs", "--epochs", "3,9"], capture_output=True, text=True,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stderr
    out = torch.load(dst)
    assert list(out.keys()) == ["models"] and torch.equal(out["models"][0]["w"], torch.full((2, 3), 6.0))
    with pytest.raises(ValueError):
        average_checkpoints([])


def test_train_entry_host_flow(tmp_path, monkeypatch):
    """wesep_amd.bin.train on CPU with a stand-in model: config parsing + `--key value` overrides, the synthetic
    collated batches (keys / shapes of tse_collate_fn for fixed-embedding, fbank and raw-audio enrollment), epochs of
    Executor.train / cv, the reference's checkpoint naming and saving rule, resume from a checkpoint."""
    import yaml
    import wesep_amd.bin.train as T
    import wesep_amd.utils.executor as ex

    class Stub(torch.nn.Module):
        def __init__(self, **kw):
            super().__init__()
            self.gain = torch.nn.Parameter(torch.tensor(0.5))

        def forward(self, mix, enroll):
            return self.gain * mix + 0.0 * enroll.float().mean(), torch.zeros(())

    monkeypatch.setattr(T, "get_model", lambda name: Stub)
    monkeypatch.setattr(T, "parse_loss", lambda loss: [lambda est, ref: ((est - ref) ** 2).mean(1)])
    monkeypatch.setattr(ex, "clip_gradients", lambda model, clip: None)
    conf = {"exp_dir": str(tmp_path / "exp"), "seed": 3, "num_epochs": 3, "num_avg": 1, "save_epoch_interval": 2,
            "log_batch_interval": 2, "clip_grad": 5.0, "enable_amp": False, "loss": "SISDR",
            "loss_args": {"loss_posi": [[0]], "loss_weight": [[1.0]]},
            "model": {"tse_model": "BSRNN"}, "model_init": {"tse_model": None},
            "model_args": {"tse_model": {"joint_training": False, "spk_emb_dim": 256}},
            "optimizer": {"tse_model": "Adam"}, "optimizer_args": {"tse_model": {"lr": 1.0, "weight_decay": 0.0}},
            "scheduler": {"tse_model": "ExponentialDecrease"},
            "scheduler_args": {"tse_model": {"initial_lr": 1e-3, "final_lr": 1e-4, "warm_up_epoch": 0}},
            "dataloader_args": {"batch_size": 2}, "dataset_args": {"chunk_len": 1600, "resample_rate": 16000}}
    path = tmp_path / "conf.yaml"
    path.write_text(yaml.dump(conf))
    configs, n = T.parse_config(["--config", str(path), "--synthetic", "4", "--dataloader_args.batch_size", "3",
                                 "--clip_grad", "3.0"])
    assert n == 4 and configs["dataloader_args"]["batch_size"] == 3 and configs["clip_grad"] == 3.0
    batch = next(iter(T.SyntheticTseLoader(configs, 1, 0)))
    assert batch["wav_mix"].shape == batch["wav_targets"].shape == (6, 1600) and batch["spk_embeds"].shape == (6, 256)
    assert batch["spk_label"].dtype == torch.int64 and len(batch["key"]) == 6
    joint = {**configs, "model_args": {"tse_model": {"joint_training": True, "spk_feat": True,
                                                     "spk_args": {"feat_dim": 80}, "spksInTrain": 11}}}
    b = next(iter(T.SyntheticTseLoader(joint, 1, 0)))
    assert b["spk_embeds"].shape == (6, 398, 80) and b["spk_embeds"].mean(1).abs().max() < 1e-5
    assert int(b["spk_label"].max()) < 11
    joint["model_args"]["tse_model"]["spk_feat"] = False
    assert next(iter(T.SyntheticTseLoader(joint, 1, 0)))["spk_embeds"].shape == (6, 64000)
    executor = T.train(configs, n)
    assert executor.step == 3 * 4
    models = sorted(os.listdir(tmp_path / "exp" / "models"))
    assert models == ["checkpoint_2.pt", "checkpoint_3.pt", "final_checkpoint.pt", "latest_checkpoint.pt"]
    assert os.readlink(tmp_path / "exp" / "models" / "final_checkpoint.pt") == "checkpoint_3.pt"
    ck = torch.load(tmp_path / "exp" / "models" / "checkpoint_3.pt", weights_only=False)
    assert set(ck) == {"models", "optimizers", "schedulers", "scaler"} and "gain" in ck["models"][0]
    assert os.path.exists(tmp_path / "exp" / "config.yaml") and os.path.exists(tmp_path / "exp" / "train.log")
    # lr comes from the scheduler's initial_lr, not from optimizer_args (train.py:236-237)
    assert ck["optimizers"][0]["param_groups"][0]["lr"] < 1.1e-3
    # resume: starts after the checkpoint's epoch
    configs["num_epochs"], configs["checkpoint"] = 4, str(tmp_path / "exp" / "models" / "checkpoint_3.pt")
    assert T.train(configs, n).step == 4
    with pytest.raises(SystemExit):
        T.train(configs, 0)
