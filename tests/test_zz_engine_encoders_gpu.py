"""GPU: the native runtime's launch plans for Bottleneck ResNets and `two_emb_layer` (runtime/engine.cc) against the
Python module tree in eval mode on the same device.  Written with the round's last GPU seconds: the case that covers both
features (ResNet50 + two_emb_layer) ran green on an MI355X, the two single-feature cases had their plans validated by the
dry run only (tests/test_engine_cpu.py) -- the file sorts last so that their first run cannot hide any other test behind
the driver's `-x`."""
import numpy as np
import pytest
import torch

from wesep_amd import engine as E

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("spk_model,two_emb", [("ResNet50", False), ("ResNet18", True), ("ResNet50", True)])
def test_engine_bottleneck_and_two_emb_layer_match_python(tmp_path, spk_model, two_emb):
    from tests.test_engine_gpu import _cuda, _joint, rel
    d = _cuda()
    model, eng = _joint(tmp_path, spk_model, d, seed=9,
                        spk_args=dict(feat_dim=80, embed_dim=256, pooling_func="TSTP", two_emb_layer=two_emb))
    assert eng.info("spk_bottleneck") == int(spk_model == "ResNet50") and eng.info("spk_two_emb") == int(two_emb)
    g = torch.Generator().manual_seed(5)
    for R, Te in ((2, 120), (3, 77)):
        wav = 0.1 * torch.randn(R, 12000, generator=g)
        fbank = torch.randn(R, Te, 80, generator=g)
        fbank = fbank - fbank.mean(1, keepdim=True)
        with torch.no_grad():
            ref = model(wav.to(d), fbank.to(d))[0]
        assert rel(eng.separate(wav.numpy(), fbank.numpy(), E.ENROLL_FBANK), ref) < 1e-4, (R, Te)
    eng.close()


def test_engine_campplus_matches_python(tmp_path):
    """CAM++ (spk_kind 2, round 5): the native runtime's launch plan against the Python module tree in eval mode, fbank
    enrollment of three lengths (segment pooling with a short last segment, one segment only, an exact multiple of 100 after
    the stride-2 TDNN layer)."""
    from tests.test_engine_gpu import _cuda, _joint, rel
    d = _cuda()
    model, eng = _joint(tmp_path, "CAMPPlus", d, seed=11, spk_emb_dim=512,
                        spk_args=dict(feat_dim=80, embed_dim=512, pooling_func="TSTP"))
    assert eng.info("spk_kind") == 2
    g = torch.Generator().manual_seed(6)
    for R, Te in ((2, 301), (3, 120), (2, 400)):
        wav = 0.1 * torch.randn(R, 12000, generator=g)
        fbank = torch.randn(R, Te, 80, generator=g)
        fbank = fbank - fbank.mean(1, keepdim=True)
        with torch.no_grad():
            ref = model(wav.to(d), fbank.to(d))[0]
        err = rel(eng.separate(wav.numpy(), fbank.numpy(), E.ENROLL_FBANK), ref)
        print(f"engine CAM++ R={R} Te={Te}: rel {err:.2e}")
        assert err < 1e-4, (R, Te, err)
    eng.close()
