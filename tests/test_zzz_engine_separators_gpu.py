"""GPU: the native runtime's DPCCN and TF-GridNet launch plans (runtime/engine.cc, arch 2 / 3) against the Python module
tree in eval mode on the same device.  The plan's first execution was a Python-free run of `runtime/separate_main` with the round's last GPU
seconds (profiles/r03_engine_dpccn_hw_check.json: joint ResNet18 + multiply fusion, 2.2e-5 from the CPU oracle chain);
the TF-GridNet plan's likewise (profiles/r03_engine_tfgridnet_hw_check.json: 1.1e-5).  These comparisons -- which also cover
fixed embeddings, FiLM fusion with SpeakerTransform, causal TCN blocks and a hidden size other than 192, whose plans have
been through the dry run only (tests/test_engine_cpu.py) -- have not run on hardware yet: the file sorts last so that its
first run cannot hide any other test behind the driver's `-x`."""
import numpy as np
import pytest
import torch

from wesep_amd import engine as E
from wesep_amd.bin.export_engine import export_engine

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("variant", ["joint-resnet18-multiply", "fixed-additive", "fixed-film-causal"])
def test_dpccn_engine_matches_python_model(tmp_path, variant):
    from tests.test_engine_gpu import _cuda, rel
    from wesep_amd.models import get_model
    d = _cuda()
    torch.manual_seed(31)
    kw = dict(tcn_blocks=3, tcn_layers=2, spk_emb_dim=256, joint_training=False)
    if variant == "fixed-additive":
        kw.update(spk_fuse_type="additive")
    elif variant == "fixed-film-causal":
        kw.update(spk_fuse_type="FiLM", causal=True, use_spk_transform=True)
    else:
        kw.update(joint_training=True, spk_model="ResNet18", spk_feat=True,
                  spk_args=dict(feat_dim=80, embed_dim=256, pooling_func="TSTP", two_emb_layer=False))
    model = get_model("DPCCN")(**kw)
    with torch.no_grad():                                   # FiLM layers start at zero; BatchNorm statistics at (0, 1)
        for name, p in model.named_parameters():
            if "gamma_fcs" in name or "beta_fcs" in name:
                p.normal_(0.0, 0.05)
        for name, buf in model.named_buffers():
            if name.endswith("running_mean"):
                buf.normal_(0.0, 0.2)
            elif name.endswith("running_var"):
                buf.uniform_(0.5, 1.5)
    path = str(tmp_path / "d.wsw")
    export_engine(model, path)
    eng = E.Engine(path)
    assert eng.info("arch") == 2
    model = model.to(d).eval()
    g = torch.Generator().manual_seed(4)
    for R, T in ((2, 16000), (1, 12345), (3, 4100)):
        wav = 0.1 * torch.randn(R, T, generator=g)
        if kw["joint_training"]:
            enroll = torch.randn(R, 120 + 7 * R, 80, generator=g)
            enroll, kind = enroll - enroll.mean(1, keepdim=True), E.ENROLL_FBANK
        else:
            enroll, kind = torch.randn(R, 256, generator=g), E.ENROLL_EMBEDDING
        est = eng.separate(wav.numpy(), enroll.numpy(), kind)
        with torch.no_grad():
            ref = model(wav.to(d), enroll.to(d))[0]
        assert est.shape == (R, T) and np.isfinite(est).all()
        assert rel(est, ref) < 1e-4, (variant, R, T, rel(est, ref))
    assert eng.info("n_launches") > 0 and eng.info("arena_bytes") > 0
    eng.close()


@pytest.mark.parametrize("variant", ["joint-resnet18-multiply", "fixed-additive", "fixed-film-hidden64"])
def test_tfgridnet_engine_matches_python_model(tmp_path, variant):
    from tests.test_engine_gpu import _cuda, rel
    from wesep_amd.models import get_model
    d = _cuda()
    torch.manual_seed(41)
    kw = dict(n_layers=2, emb_dim=128, emb_ks=1, emb_hs=1, lstm_hidden_units=192, spk_emb_dim=256, joint_training=False)
    if variant == "fixed-additive":
        kw.update(spk_fuse_type="additive")
    elif variant == "fixed-film-hidden64":
        kw.update(spk_fuse_type="FiLM", lstm_hidden_units=64, use_spk_transform=True)
    else:
        kw.update(joint_training=True, spk_model="ResNet18", spk_feat=True,
                  spk_args=dict(feat_dim=80, embed_dim=256, pooling_func="TSTP", two_emb_layer=False))
    model = get_model("TFGridNet")(**kw)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if "gamma_fcs" in name or "beta_fcs" in name:
                p.normal_(0.0, 0.05)
            elif name.endswith(("gamma", "norm.weight", "conv.1.weight")):
                p.uniform_(0.5, 1.5)
            elif name.endswith(("beta", "norm.bias", "conv.1.bias")):
                p.normal_(0.0, 0.1)
        for name, buf in model.named_buffers():
            if name.endswith("running_mean"):
                buf.normal_(0.0, 0.2)
            elif name.endswith("running_var"):
                buf.uniform_(0.5, 1.5)
    path = str(tmp_path / "g.wsw")
    export_engine(model, path)
    eng = E.Engine(path)
    assert eng.info("arch") == 3
    model = model.to(d).eval()
    g = torch.Generator().manual_seed(6)
    for R, T in ((2, 16000), (1, 12344), (3, 4000)):
        wav = 0.1 * torch.randn(R, T, generator=g)
        if kw["joint_training"]:
            enroll = torch.randn(R, 120 + 7 * R, 80, generator=g)
            enroll, kind = enroll - enroll.mean(1, keepdim=True), E.ENROLL_FBANK
        else:
            enroll, kind = torch.randn(R, 256, generator=g), E.ENROLL_EMBEDDING
        est = eng.separate(wav.numpy(), enroll.numpy(), kind)
        with torch.no_grad():
            ref = model(wav.to(d), enroll.to(d))[0]
        assert est.shape == (R, T) and np.isfinite(est).all()
        assert rel(est, ref) < 1e-4, (variant, R, T, rel(est, ref))
    eng.close()


# ---- against the CPU ORACLE: the comparisons above hold the engine to the HIP Python tree; these
#      hold it to the oracle chain directly, as tests/test_engine_gpu.py does for pBSRNN (round 3 had them only as the one-off
#      checks of tools/make_engine_testdata_{dpccn,tfgridnet}.py --check) ------------------------------------------------------
@pytest.mark.parametrize("fuse", ["multiply", "additive", "concat"])
def test_dpccn_engine_matches_oracle(tmp_path, fuse):
    from oracle import dpccn_oracle as DP
    from tests.test_engine_gpu import _cuda, rel
    from wesep_amd.models import get_model
    _cuda()
    kw = dict(tcn_blocks=3, tcn_layers=2, spk_fuse_type=fuse)
    cfg = DP.DPCCNConfig(**kw)
    params = DP.synth_params(cfg, 61)
    model = get_model("DPCCN")(**kw, joint_training=False)
    model.load_state_dict(params, strict=True)
    path = str(tmp_path / "d.wsw")
    export_engine(model, path)
    eng = E.Engine(path)
    g = torch.Generator().manual_seed(7)
    for R, T in ((2, 8000), (1, 6001)):
        wav, emb = 0.1 * torch.randn(R, T, generator=g), torch.randn(R, 256, generator=g)
        est = eng.separate(wav.numpy(), emb.numpy(), E.ENROLL_EMBEDDING)
        with torch.no_grad():
            ref = DP.dpccn_forward(params, cfg, wav, emb)
        assert rel(est, ref) < 1e-3, (fuse, R, T, rel(est, ref))     # north_star: 1e-3 on the separated waveform
    eng.close()


@pytest.mark.parametrize("fuse", ["multiply", "additive", "concat"])
def test_tfgridnet_engine_matches_oracle(tmp_path, fuse):
    from oracle import tfgridnet_oracle as TG
    from tests.test_engine_gpu import _cuda, rel
    from wesep_amd.models import get_model
    _cuda()
    kw = dict(n_fft=128, stride=64, n_layers=2, lstm_hidden_units=192, attn_n_head=4, attn_approx_qk_dim=512, emb_dim=128,
              emb_ks=1, emb_hs=1, spk_fuse_type=fuse)
    cfg = TG.TFGridNetConfig(**kw)
    params = TG.synth_params(cfg, 62)
    model = get_model("TFGridNet")(**kw, joint_training=False)
    model.load_state_dict(params, strict=True)
    path = str(tmp_path / "g.wsw")
    export_engine(model, path)
    eng = E.Engine(path)
    g = torch.Generator().manual_seed(8)
    for R, T in ((2, 8000), (1, 6016)):
        wav, emb = 0.1 * torch.randn(R, T, generator=g), torch.randn(R, 256, generator=g)
        est = eng.separate(wav.numpy(), emb.numpy(), E.ENROLL_EMBEDDING)
        with torch.no_grad():
            ref = TG.tfgridnet_forward(params, cfg, wav, emb)
        assert rel(est, ref) < 1e-3, (fuse, R, T, rel(est, ref))
    eng.close()
