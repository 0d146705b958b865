"""CPU: the native runtime (include/wesep_engine.h, runtime/engine.cc) without a GPU -- library and symbols, the
weight container written by `wesep_amd.bin.export_engine`, the engine's DRY RUN (every launch of the plan must pass
its entry point's argument validation in the real libwesep_hip.so; nothing is computed), the wav reader / writer and
the `separate_main` command-line tool."""
import ctypes
import os
import struct
import subprocess
import wave

import numpy as np
import pytest
import torch

from wesep_amd import engine as E
from wesep_amd.bin.export_engine import export_engine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
needs_no_gpu = pytest.mark.skipif(torch.cuda.is_available(), reason="the engine's dry run is refused when a GPU is visible")
SPK = dict(joint_training=True, spk_feat=True,
           spk_args=dict(feat_dim=80, embed_dim=256, pooling_func="TSTP", two_emb_layer=False))
FIXED = [dict(num_repeat=2, spk_fuse_type="multiply", multi_fuse=False, use_spk_transform=False),
         dict(num_repeat=2, spk_fuse_type="FiLM", multi_fuse=True, use_spk_transform=True),
         dict(num_repeat=1, spk_fuse_type="concat", multi_fuse=True, use_spk_transform=False),
         dict(num_repeat=1, spk_fuse_type="additive", multi_fuse=False, use_spk_transform=True)]


def _model(**kw):
    from wesep_amd.models import get_model
    return get_model("BSRNN")(**kw)


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(E.LIB_PATH)
    header = open(os.path.join(ROOT, "include", "wesep_engine.h")).read()
    import re
    declared = sorted(set(re.findall(r"\b(ws_engine_\w+)\s*\(", header)))
    assert declared == sorted(E.SYMBOLS)                       # the binding knows every entry point of the header
    for sym in declared:
        assert hasattr(lib, sym), sym
    assert E.lib().ws_engine_abi_version() == E.ENGINE_ABI_VERSION


@needs_no_gpu
def test_container_layout_and_rejection(tmp_path):
    m = _model(joint_training=False, **FIXED[0])
    path = str(tmp_path / "m.wsw")
    n, nf = export_engine(m, path)
    sd = {k: v for k, v in m.state_dict().items()}
    assert n == len(sd)
    raw = open(path, "rb").read()
    assert raw[:8] == b"WSEPW001" and len(raw) > nf * 4
    # the data section holds every tensor bit-exactly at a 16-byte aligned offset: spot-check through the header
    n_meta = struct.unpack_from("<I", raw, 8)[0]
    pos = 12 + n_meta * 40
    n_t = struct.unpack_from("<I", raw, pos)[0]
    pos += 4
    table = {}
    for _ in range(n_t):
        ln = struct.unpack_from("<I", raw, pos)[0]
        name = raw[pos + 4:pos + 4 + ln].decode()
        pos += 4 + ln
        nd = struct.unpack_from("<I", raw, pos)[0]
        dims = struct.unpack_from(f"<{nd}q", raw, pos + 4)
        off = struct.unpack_from("<Q", raw, pos + 4 + 8 * nd)[0]
        pos += 4 + 8 * nd + 8
        table[name] = (dims, off)
    assert struct.unpack_from("<Q", raw, pos)[0] == nf
    data = np.frombuffer(raw, dtype=np.float32, offset=pos + 8)
    for k in ("BN.3.1.weight", "separator.separation.1.band_rnn.rnn.weight_hh_l0_reverse", "mask.31.5.bias"):
        dims, off = table[k]
        assert off % 4 == 0 and tuple(dims) == tuple(sd[k].shape)
        assert np.array_equal(data[off:off + sd[k].numel()], sd[k].numpy().reshape(-1))
    # a truncated file and a non-container are refused with a message, not a crash
    bad = str(tmp_path / "bad.wsw")
    open(bad, "wb").write(raw[:len(raw) // 2])
    with pytest.raises(E.WesepHipError, match="not a valid"):
        E.Engine(bad, dry_run=True)
    open(bad, "wb").write(b"not a model")
    with pytest.raises(E.WesepHipError):
        E.Engine(bad, dry_run=True)
    with pytest.raises(E.WesepHipError, match="cannot open"):
        E.Engine(str(tmp_path / "missing.wsw"), dry_run=True)
    # a container with a tensor missing is refused at load (names the tensor)
    sd.pop("mask.7.3.weight")
    from wesep_amd.bin.export_engine import engine_meta, write_container
    write_container(bad, engine_meta(m), sd)
    with pytest.raises(E.WesepHipError, match="mask.7.3.weight"):
        E.Engine(bad, dry_run=True)


@needs_no_gpu
@pytest.mark.parametrize("kw", FIXED, ids=[k["spk_fuse_type"] for k in FIXED])
def test_dry_run_launch_plan_fixed_embeddings(tmp_path, kw):
    """Every fusion variant, ragged and long lengths, 1 / 2 / 32 rows: the whole plan passes argument validation."""
    path = str(tmp_path / "m.wsw")
    export_engine(_model(joint_training=False, **kw), path)
    eng = E.Engine(path, dry_run=True)
    assert eng.info("num_repeat") == kw["num_repeat"] and eng.info("joint_training") == 0 and eng.info("nband") == 32
    counts = set()
    for R, T in ((2, 16000), (2, 12345), (1, 64000), (32, 8192), (2, 512)):
        est = eng.separate(np.ones((R, T), np.float32), np.zeros((R, 256), np.float32), E.ENROLL_EMBEDDING)
        assert est.shape == (R, T) and not est.any()           # a dry run computes nothing
        counts.add(eng.info("n_launches"))
        assert eng.info("arena_bytes") > 0
    assert len(counts) == 1                                    # the plan does not depend on the geometry
    with pytest.raises(E.WesepHipError, match="T >= 512"):
        eng.separate(np.zeros((2, 300), np.float32), np.zeros((2, 256), np.float32), E.ENROLL_EMBEDDING)
    with pytest.raises(E.WesepHipError, match="does not fit"):
        eng.separate(np.zeros((2, 4000), np.float32), np.zeros((2, 98, 80), np.float32), E.ENROLL_FBANK)
    eng.close()


@needs_no_gpu
@pytest.mark.parametrize("spk_model", ["ResNet18", "ResNet34"])
def test_dry_run_launch_plan_joint_model(tmp_path, spk_model):
    path = str(tmp_path / "j.wsw")
    export_engine(_model(num_repeat=1, spk_fuse_type="multiply", multi_fuse=False, use_spk_transform=False,
                         spk_model=spk_model, **SPK), path)
    eng = E.Engine(path, dry_run=True)
    assert eng.info("joint_training") == 1 and eng.info("feat_dim") == 80
    eng.separate(np.zeros((2, 16000), np.float32), np.zeros((2, 98, 80), np.float32), E.ENROLL_FBANK)
    n_fbank = eng.info("n_launches")
    eng.separate(np.zeros((2, 16000), np.float32), np.zeros((2, 24001), np.float32), E.ENROLL_WAVE)
    assert eng.info("n_launches") > n_fbank                    # + the kaldi fbank / CMN launches
    out = eng.forward_pcm16(np.zeros(32000, np.int16), np.zeros(48000, np.int16), np.zeros(50001, np.int16))
    assert out.shape == (2, 32000)
    with pytest.raises(E.WesepHipError, match="does not fit"):
        eng.separate(np.zeros((2, 16000), np.float32), np.zeros((2, 256), np.float32), E.ENROLL_EMBEDDING)
    with pytest.raises(E.WesepHipError, match="shorter than one"):
        eng.separate(np.zeros((2, 16000), np.float32), np.zeros((2, 300), np.float32), E.ENROLL_WAVE)
    eng.close()


@needs_no_gpu
@pytest.mark.parametrize("spk_model,emb_bn", [("ECAPA_TDNN_GLOB_c512", False), ("ECAPA_TDNN_c512", True),
                                              ("ECAPA_TDNN_GLOB_c1024", False)])
def test_dry_run_launch_plan_ecapa_joint_model(tmp_path, spk_model, emb_bn):
    """The layout of the reference's published `bsrnn_ecapa_vox1` model (wesep/cli/hub.py:86-95): a pBSRNN with the
    wespeaker ECAPA-TDNN speaker encoder exports to the native runtime and every launch of its plan (dilated Conv1d as
    implicit-patch GEMMs, Res2Net branches, SE gates, attentive statistics pooling) passes argument validation."""
    path = str(tmp_path / "e.wsw")
    export_engine(_model(num_repeat=1, spk_fuse_type="multiply", multi_fuse=False, use_spk_transform=False,
                         spk_emb_dim=192, joint_training=True, spk_feat=True, spk_model=spk_model,
                         spk_args=dict(feat_dim=80, embed_dim=192, pooling_func="ASTP", emb_bn=emb_bn)), path)
    eng = E.Engine(path, dry_run=True)
    assert eng.info("spk_kind") == 1 and eng.info("spk_glob") == int("GLOB" in spk_model)
    assert eng.info("spk_channels") == (1024 if "1024" in spk_model else 512) and eng.info("spk_emb_bn") == int(emb_bn)
    counts = set()
    for R, Te in ((2, 98), (1, 301), (4, 37)):
        eng.separate(np.zeros((R, 16000), np.float32), np.zeros((R, Te, 80), np.float32), E.ENROLL_FBANK)
        counts.add(eng.info("n_launches"))
    assert len(counts) == 1
    eng.separate(np.zeros((2, 16000), np.float32), np.zeros((2, 24001), np.float32), E.ENROLL_WAVE)
    assert eng.info("n_launches") > counts.pop()
    with pytest.raises(E.WesepHipError, match="does not fit"):
        eng.separate(np.zeros((2, 16000), np.float32), np.zeros((2, 192), np.float32), E.ENROLL_EMBEDDING)
    eng.close()


@needs_no_gpu
def test_dry_run_launch_plan_campplus(tmp_path):
    """The wespeaker CAM++ encoder in the native runtime (round 5: spk_kind 2): FCM head with mel-axis strides, the three
    CAM-dense-TDNN blocks with segment pooling (a last segment shorter than 100 frames and an exact multiple), TSTP, the
    affine-free embedding BatchNorm -- every launch through the real library's argument validation."""
    path = str(tmp_path / "c.wsw")
    export_engine(_model(num_repeat=1, spk_fuse_type="multiply", multi_fuse=False, use_spk_transform=False,
                         joint_training=True, spk_feat=True, spk_model="CAMPPlus", spk_emb_dim=512,
                         spk_args=dict(feat_dim=80, embed_dim=512, pooling_func="TSTP")), path)
    eng = E.Engine(path, dry_run=True)
    assert eng.info("spk_kind") == 2 and eng.info("feat_dim") == 80
    counts = set()
    for R, Te in ((2, 98), (3, 301), (2, 400)):
        est = eng.separate(np.zeros((R, 16000), np.float32), np.zeros((R, Te, 80), np.float32), E.ENROLL_FBANK)
        assert est.shape == (R, 16000)
        counts.add(eng.info("n_launches"))
    assert len(counts) == 3
    eng.close()


def test_export_refuses_speaker_encoders_without_a_launch_plan(tmp_path):
    for spk_model, args, E_ in (("CAMPPlus", dict(feat_dim=80, embed_dim=512, pooling_func="TSTP", growth_rate=16), 512),
                                ("ResNet18", dict(feat_dim=80, embed_dim=256, pooling_func="ASTP", two_emb_layer=False), 256)):
        with pytest.raises(NotImplementedError, match="no launch plan"):
            export_engine(_model(num_repeat=1, spk_fuse_type="multiply", multi_fuse=False, use_spk_transform=False,
                                 joint_training=True, spk_feat=True, spk_model=spk_model, spk_emb_dim=E_, spk_args=args),
                          str(tmp_path / "x.wsw"))


@needs_no_gpu
@pytest.mark.parametrize("spk_model,two_emb", [("ResNet50", False), ("ResNet18", True), ("ResNet101", True)])
def test_dry_run_launch_plan_bottleneck_and_two_emb_layer(tmp_path, spk_model, two_emb):
    """Bottleneck ResNets (1x1 - 3x3(stride) - 1x1, expansion 4) and `two_emb_layer` (seg_1 -> ReLU -> BatchNorm1d(affine =
    False) -> seg_2, the separator takes the second embedding) in the native runtime: export metadata and the launch
    plan's argument validation for three geometries."""
    path = str(tmp_path / "b.wsw")
    export_engine(_model(num_repeat=1, spk_fuse_type="multiply", multi_fuse=False, use_spk_transform=False,
                         joint_training=True, spk_feat=True, spk_model=spk_model,
                         spk_args=dict(feat_dim=80, embed_dim=256, pooling_func="TSTP", two_emb_layer=two_emb)), path)
    eng = E.Engine(path, dry_run=True)
    assert eng.info("spk_kind") == 0 and eng.info("spk_bottleneck") == int(spk_model != "ResNet18")
    assert eng.info("spk_two_emb") == int(two_emb) and eng.info("feat_dim") == 80
    counts = set()
    for R, Te in ((2, 98), (2, 301), (2, 40)):
        eng.separate(np.zeros((R, 16000), np.float32), np.zeros((R, Te, 80), np.float32), E.ENROLL_FBANK)
        counts.add(eng.info("n_launches"))
    assert len(counts) == 1                                    # (one fbank transpose per row: the plan depends on R only)
    eng.separate(np.zeros((1, 16000), np.float32), np.zeros((1, 77, 80), np.float32), E.ENROLL_FBANK)
    eng.close()


@needs_no_gpu
def test_dry_run_raw_audio_joint_model(tmp_path):
    """spk_feat = False (bsrnn_multi_optim.yaml): the in-model PreEmphasis + MelSpectrogram front-end runs in the
    engine from the model's own window / filterbank buffers; a `BSRNN_Multi` checkpoint exports like a `BSRNN`."""
    from wesep_amd.models import get_model
    path = str(tmp_path / "m.wsw")
    model = get_model("BSRNN_Multi")(num_repeat=1, spk_fuse_type="multiply", multi_fuse=False, use_spk_transform=False,
                                     joint_training=True, spk_feat=False, spk_model="ResNet18",
                                     spk_args=SPK["spk_args"])
    export_engine(model, path)
    eng = E.Engine(path, dry_run=True)
    assert eng.info("spk_feat") == 0 and eng.info("joint_training") == 1
    for Tw in (16000, 12345, 1100):
        eng.separate(np.zeros((2, 8000), np.float32), np.zeros((2, Tw), np.float32), E.ENROLL_WAVE)
    with pytest.raises(E.WesepHipError, match="computes its own features"):
        eng.separate(np.zeros((2, 8000), np.float32), np.zeros((2, 98, 80), np.float32), E.ENROLL_FBANK)
    with pytest.raises(E.WesepHipError, match="reflect padding"):
        eng.separate(np.zeros((2, 8000), np.float32), np.zeros((2, 256), np.float32), E.ENROLL_WAVE)
    eng.close()


@needs_no_gpu
def test_infer_extract_engine_rank_dispatch(tmp_path):
    from wesep_amd.bin.infer import extract_engine
    path = str(tmp_path / "m.wsw")
    export_engine(_model(joint_training=False, **FIXED[0]), path)
    eng = E.Engine(path, dry_run=True)
    out = extract_engine(eng, np.ones((2, 4000)), np.zeros((2, 256)))         # float64 in: converted
    assert out.shape == (2, 4000) and out.dtype == np.float32 and not out.any()   # all-zero rows: not normalised
    with pytest.raises(E.WesepHipError, match="does not fit"):
        extract_engine(eng, np.ones((2, 4000)), np.zeros((2, 98, 80)))       # 3-D enrollment = fbank
    eng.close()


def test_export_refuses_models_the_runtime_does_not_run():
    from wesep_amd.models import get_model
    with pytest.raises(NotImplementedError, match="emb_dim 48, emb_ks 4"):          # the constructor defaults
        export_engine(get_model("TFGridNet")(joint_training=False, n_layers=1), "/dev/null")
    with pytest.raises(NotImplementedError, match="n_imics 2"):
        export_engine(get_model("TFGridNet")(joint_training=False, n_layers=1, emb_dim=128, emb_ks=1, emb_hs=1, n_imics=2),
                      "/dev/null")
    for kw, what in ((dict(norm="cLN"), "gLN only"), (dict(causal=True), "causal"), (dict(skip_con=True), "skip"),
                     (dict(spk_fuse_type="FiLM"), "concatConv only"),
                     (dict(encoder_type="Deep", decoder_type="Deep"), "Multi only")):
        with pytest.raises(NotImplementedError, match=what):
            export_engine(get_model("ConvTasNet")(N=32, L=20, B=32, H=64, P=3, X=2, R=1, joint_training=False, **kw),
                          "/dev/null")


@needs_no_gpu
@pytest.mark.parametrize("joint", [False, True], ids=["fixed-embeddings", "spex-plus"])
def test_dry_run_launch_plan_convtasnet(tmp_path, joint):
    """Conv-TasNet / SpEx+ in the native runtime (arch 1): container, geometry read back, and the whole launch plan
    through the real library's argument validation for several lengths and row counts."""
    from wesep_amd.models import get_model
    kw = dict(N=256, L=20, B=64, H=128, P=3, X=3, R=2, spk_emb_dim=256, joint_training=joint)
    m = get_model("ConvTasNet")(**kw)
    path = str(tmp_path / "t.wsw")
    n, _ = export_engine(m, path)
    assert n == sum(1 for k, v in m.state_dict().items() if torch.is_floating_point(v) and not k.startswith("pred_linear."))
    eng = E.Engine(path, dry_run=True)
    assert eng.info("arch") == 1 and eng.info("N") == 256 and eng.info("X") == 3 and eng.info("R") == 2
    assert eng.info("spk_emb_dim") == 256 and eng.info("joint_training") == int(joint)
    counts = set()
    for R, T in ((2, 16000), (1, 12345), (4, 4000), (2, 160)):
        if joint:
            enroll, kind = np.zeros((R, 9000), np.float32), E.ENROLL_WAVE
        else:
            enroll, kind = np.zeros((R, 256), np.float32), E.ENROLL_EMBEDDING
        est = eng.separate(np.ones((R, T), np.float32), enroll, kind)
        assert est.shape == (R, T) and not est.any()
        counts.add(eng.info("n_launches"))
        assert eng.info("arena_bytes") > 0
    assert len(counts) == 1
    with pytest.raises(E.WesepHipError, match="T >= 160"):
        eng.separate(np.zeros((2, 100), np.float32), np.zeros((2, 256), np.float32), E.ENROLL_EMBEDDING)
    wrong = (np.zeros((2, 256), np.float32), E.ENROLL_EMBEDDING) if joint else (np.zeros((2, 9000), np.float32), E.ENROLL_WAVE)
    with pytest.raises(E.WesepHipError, match="Conv-TasNet engine takes"):
        eng.separate(np.zeros((2, 4000), np.float32), *wrong)
    if joint:
        with pytest.raises(E.WesepHipError, match="too short"):
            eng.separate(np.zeros((2, 4000), np.float32), np.zeros((2, 200), np.float32), E.ENROLL_WAVE)
    eng.close()


@needs_no_gpu
@pytest.mark.parametrize("variant", ["fixed-multiply", "fixed-film-causal", "joint-resnet18-additive", "fixed-concat"])
def test_dry_run_launch_plan_dpccn(tmp_path, variant):
    """DPCCN in the native runtime (arch 2): container, geometry read back, and the whole launch plan -- DFT-basis STFT,
    halo-tile dense blocks, strided / transposed implicit-GEMM convolutions, fused ELU + InstanceNorm, the TCN stack, the
    pooling branches, inverse STFT -- through the real library's argument validation for several lengths and row counts."""
    from wesep_amd.models import get_model
    kw = dict(tcn_blocks=3, tcn_layers=2, spk_emb_dim=256, joint_training=False)
    if variant == "fixed-film-causal":
        kw.update(spk_fuse_type="FiLM", causal=True, use_spk_transform=True)
    elif variant == "fixed-concat":        # round 5: the Linear over the frequency axis of cat[x, e] (ws_freq_linear_fwd)
        kw.update(spk_fuse_type="concat")
    elif variant == "joint-resnet18-additive":
        kw.update(spk_fuse_type="additive", joint_training=True, spk_model="ResNet18", spk_feat=True,
                  spk_args=dict(feat_dim=80, embed_dim=256, pooling_func="TSTP", two_emb_layer=False))
    m = get_model("DPCCN")(**kw)
    path = str(tmp_path / "d.wsw")
    n, _ = export_engine(m, path)
    assert n == sum(1 for k, v in m.state_dict().items() if torch.is_floating_point(v) and not k.startswith("pred_linear.")
                    and not k.endswith("num_batches_tracked"))
    eng = E.Engine(path, dry_run=True)
    assert eng.info("arch") == 2 and eng.info("tcn_blocks") == 3 and eng.info("tcn_layers") == 2
    assert eng.info("causal") == int(variant == "fixed-film-causal") and eng.info("joint_training") == int(kw["joint_training"])
    counts = set()
    for R, T in ((2, 16000), (1, 12345), (3, 4000)):
        if kw["joint_training"]:
            enroll, kind = np.zeros((R, 150, 80), np.float32), E.ENROLL_FBANK
        else:
            enroll, kind = np.zeros((R, 256), np.float32), E.ENROLL_EMBEDDING
        est = eng.separate(np.ones((R, T), np.float32), enroll, kind)
        assert est.shape == (R, T) and not est.any()
        counts.add(eng.info("n_launches"))
        assert eng.info("arena_bytes") > 0
    assert len(counts) == (3 if kw["joint_training"] else 1)      # the ResNet front-end transposes one row per launch
    with pytest.raises(E.WesepHipError, match="32 frames"):
        eng.separate(np.zeros((2, 3000), np.float32), enroll[:2], kind)
    eng.close()


@needs_no_gpu
@pytest.mark.parametrize("variant", ["fixed-multiply", "fixed-film-hidden64", "joint-resnet18-additive", "fixed-concat"])
def test_dry_run_launch_plan_tfgridnet(tmp_path, variant):
    """TF-GridNet in the native runtime (arch 3, the recipe's geometry): container, geometry read back, and the whole
    launch plan -- DFT-basis STFT, GroupNorm, per block the row LayerNorms, both BLSTM paths on the blocked-layout kernels
    (the inter-frame path on a strided sequence map), the projection GEMM, the attention-head kernels, grouped logits /
    value GEMMs, softmax, head merge, projection + PReLU + LayerNorm, residual -- through the real library's argument
    validation for several lengths and row counts."""
    from wesep_amd.models import get_model
    kw = dict(n_layers=2, emb_dim=128, emb_ks=1, emb_hs=1, lstm_hidden_units=192, spk_emb_dim=256, joint_training=False)
    if variant == "fixed-film-hidden64":
        kw.update(spk_fuse_type="FiLM", lstm_hidden_units=64, use_spk_transform=True)
    elif variant == "fixed-concat":
        kw.update(spk_fuse_type="concat")
    elif variant == "joint-resnet18-additive":
        kw.update(spk_fuse_type="additive", joint_training=True, spk_model="ResNet18", spk_feat=True,
                  spk_args=dict(feat_dim=80, embed_dim=256, pooling_func="TSTP", two_emb_layer=False))
    m = get_model("TFGridNet")(**kw)
    path = str(tmp_path / "g.wsw")
    export_engine(m, path)
    eng = E.Engine(path, dry_run=True)
    assert eng.info("arch") == 3 and eng.info("n_layers") == 2 and eng.info("lstm_hidden_units") == kw["lstm_hidden_units"]
    assert eng.info("attn_E") == 8 and eng.info("attn_n_head") == 4 and eng.info("joint_training") == int(kw["joint_training"])
    for R, T in ((2, 16000), (1, 12345), (3, 4001), (64, 2048)):        # any sample count (the Python model wants T % 4 == 0)
        if kw["joint_training"]:
            enroll, kind = np.zeros((R, 150, 80), np.float32), E.ENROLL_FBANK
        else:
            enroll, kind = np.zeros((R, 256), np.float32), E.ENROLL_EMBEDDING
        mix = np.random.default_rng(R).standard_normal((R, T)).astype(np.float32)
        est = eng.separate(mix, enroll, kind)
        assert est.shape == (R, T) and not est.any()
        assert eng.info("n_launches") > 0 and eng.info("arena_bytes") > 0
    with pytest.raises(E.WesepHipError, match="T >= 512"):
        eng.separate(np.zeros((2, 200), np.float32), enroll[:2], kind)
    eng.close()


@needs_no_gpu
@pytest.mark.parametrize("name", ["DPCCN", "TFGridNet"])
def test_shipped_recipe_models_export_and_dry_run(tmp_path, name):
    """model_args of examples/librimix/tse/v2/confs/dpccn.yaml / tfgridnet.yaml (joint ResNet34 on 80-bin fbank; the stray
    `multi_fuse` key of tfgridnet.yaml dropped: the reference's own constructor does not take it either): full-size
    containers, every launch of the plan through the real library's argument validation."""
    from wesep_amd.models import get_model
    spk = dict(joint_training=True, spk_model="ResNet34", spk_model_init=False, spk_emb_dim=256, spk_model_freeze=False,
               spk_feat=True, feat_type="consistent", use_spk_transform=False, spk_fuse_type="multiply",
               spk_args=dict(feat_dim=80, embed_dim=256, pooling_func="TSTP", two_emb_layer=False))
    if name == "DPCCN":
        kw = dict(win=512, stride=128, feature_dim=257, tcn_blocks=10, tcn_layers=2, causal=False, multi_fuse=False, **spk)
    else:
        kw = dict(n_srcs=1, sr=16000, n_fft=128, stride=64, window="hann", n_imics=1, n_layers=6, lstm_hidden_units=192,
                  attn_n_head=4, attn_approx_qk_dim=512, emb_dim=128, emb_ks=1, emb_hs=1, activation="prelu", eps=1.0e-5,
                  multi_task=False, spksInTrain=251, **spk)
    path = str(tmp_path / "r.wsw")
    export_engine(get_model(name)(**kw), path)
    eng = E.Engine(path, dry_run=True)
    assert eng.info("arch") == (2 if name == "DPCCN" else 3) and eng.info("joint_training") == 1 and eng.info("spk_blocks1") == 4
    est = eng.separate(np.random.default_rng(0).standard_normal((2, 48001)).astype(np.float32), np.zeros((2, 298, 80), np.float32),
                       E.ENROLL_FBANK)
    assert est.shape == (2, 48001) and eng.info("n_launches") > 300
    out = eng.forward_pcm16(np.zeros(32000, np.int16) + 3, np.zeros(48000, np.int16), np.zeros(50001, np.int16))
    assert out.shape == (2, 32000)
    eng.close()


def _write_wav(path, x, sr=16000):
    with wave.open(str(path), "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(sr)
        w.writeframes(np.asarray(x, dtype=np.int16).tobytes())


@needs_no_gpu
def test_separate_main_dry_run_and_argument_errors(tmp_path):
    exe = os.path.join(ROOT, "runtime", "separate_main")
    assert os.path.exists(exe), "run python -m wesep_amd.build"
    model = str(tmp_path / "j.wsw")
    export_engine(_model(num_repeat=1, spk_fuse_type="multiply", multi_fuse=False, use_spk_transform=False,
                         spk_model="ResNet18", **SPK), model)
    rng = np.random.default_rng(0)
    for name, n in (("mix", 24000), ("e1", 32000), ("e2", 40000)):
        _write_wav(tmp_path / f"{name}.wav", rng.integers(-3000, 3000, n))
    scp = tmp_path / "wav.scp"
    scp.write_text(f"utt1 {tmp_path}/mix.wav {tmp_path}/e1.wav {tmp_path}/e2.wav\n"
                   f"utt2 {tmp_path}/mix.wav {tmp_path}/e2.wav {tmp_path}/e1.wav\n")
    r = subprocess.run([exe, "--wav_scp", str(scp), f"--model={model}", "--dry_run"], capture_output=True, text=True,
                       timeout=120)
    assert r.returncode == 0, r.stderr
    assert r.stdout.count("process: utt") == 2 and "[dry run]" in r.stdout and "RTF:" in r.stdout
    assert "Total: process 3000ms audio" in r.stdout
    # utterance-level concurrency: worker threads with one engine each
    scp3 = tmp_path / "wav3.scp"
    scp3.write_text("".join(f"u{i} {tmp_path}/mix.wav {tmp_path}/e1.wav {tmp_path}/e2.wav\n" for i in range(5)))
    r = subprocess.run([exe, "--wav_scp", str(scp3), "--model", model, "--dry_run", "--jobs", "3", "--devices", "0,1"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert sorted(l.split()[1] for l in r.stdout.splitlines() if l.startswith("process:")) == [f"u{i}" for i in range(5)]
    assert "Total: process 7500ms audio" in r.stdout
    # single-utterance flags of the reference tool
    r = subprocess.run([exe, "--wav_path", f"{tmp_path}/mix.wav", "--spk1_emb", f"{tmp_path}/e1.wav", "--spk2_emb",
                        f"{tmp_path}/e2.wav", "--model", model, "--dry_run"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "process: test" in r.stdout
    # errors: missing model, bad scp line, wrong sample rate, missing output dir
    assert subprocess.run([exe, "--wav_scp", str(scp)], capture_output=True).returncode == 1
    bad = tmp_path / "bad.scp"
    bad.write_text("utt1 only_two_fields\n")
    r = subprocess.run([exe, "--wav_scp", str(bad), "--model", model, "--dry_run"], capture_output=True, text=True)
    assert r.returncode == 1 and "4 fields" in r.stderr
    _write_wav(tmp_path / "mix8k.wav", rng.integers(-3000, 3000, 8000), sr=8000)
    r = subprocess.run([exe, "--wav_path", f"{tmp_path}/mix8k.wav", "--spk1_emb", f"{tmp_path}/e1.wav", "--spk2_emb",
                        f"{tmp_path}/e2.wav", "--model", model, "--dry_run"], capture_output=True, text=True)
    assert r.returncode == 1 and "sample rate" in r.stderr
    r = subprocess.run([exe, "--wav_scp", str(scp), "--model", model], capture_output=True, text=True)
    assert r.returncode == 1 and "Invalid output path" in r.stderr


@needs_no_gpu
def test_separate_main_dry_run_with_a_spex_plus_container(tmp_path):
    """The reference tool's interface on a Conv-TasNet / SpEx+ model file: mixture + two enrollment utterances, the
    enrollment waveforms go through the shared encoder and the SpEx+ speaker encoder inside the engine."""
    from wesep_amd.models import get_model
    exe = os.path.join(ROOT, "runtime", "separate_main")
    model = str(tmp_path / "spex.wsw")
    export_engine(get_model("ConvTasNet")(N=256, L=20, B=64, H=128, P=3, X=3, R=2, spk_emb_dim=256, joint_training=True),
                  model)
    rng = np.random.default_rng(1)
    for name, n in (("mix", 24000), ("e1", 32000), ("e2", 40000)):
        _write_wav(tmp_path / f"{name}.wav", rng.integers(-3000, 3000, n))
    scp = tmp_path / "wav.scp"
    scp.write_text(f"utt1 {tmp_path}/mix.wav {tmp_path}/e1.wav {tmp_path}/e2.wav\n")
    r = subprocess.run([exe, "--wav_scp", str(scp), "--model", model, "--dry_run"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "process: utt1" in r.stdout and "[dry run]" in r.stdout
