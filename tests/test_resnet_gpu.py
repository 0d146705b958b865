"""GPU parity of the wespeaker ResNet speaker encoder path against plain torch and the
restatement in oracle/resnet_oracle.py.  The upstream package is absent: parity is UNPINNED (see the oracle's header)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _cuda():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _cl(x):        # [R, C, H, W] -> channels-last rows [R*H*W, C]
    R, C, H, W = x.shape
    return x.permute(0, 2, 3, 1).reshape(R * H * W, C).contiguous()


def _nchw(y, R, H, W):
    return y.view(R, H, W, -1).permute(0, 3, 1, 2)


@pytest.mark.parametrize("Cin,Cout,k,s,relu,with_res", [(1, 8, 3, 1, True, False), (8, 16, 3, 2, True, False),
                                                        (8, 12, 1, 2, False, False), (16, 16, 3, 1, True, True)])
def test_conv_bn_act_matches_torch(Cin, Cout, k, s, relu, with_res):
    from wesep_amd import functional_resnet as FR
    d = _cuda()
    torch.manual_seed(Cin * 100 + Cout)
    R, H, W = 2, 11, 14
    x = torch.randn(R, Cin, H, W, device=d)
    w = (torch.randn(Cout, Cin, k, k, device=d) * 0.3).requires_grad_(True)
    gamma = (torch.rand(Cout, device=d) + 0.5).requires_grad_(True)
    beta = (torch.randn(Cout, device=d) * 0.1).requires_grad_(True)
    Ho, Wo = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
    res = torch.randn(R, Cout, Ho, Wo, device=d) if with_res else None
    rm, rv = torch.zeros(Cout, device=d), torch.ones(Cout, device=d)
    xin = _cl(x).requires_grad_(Cin > 1)
    rin = _cl(res).requires_grad_(True) if with_res else None
    y = FR.ConvBnActFn.apply(xin, rin, (R, H, W, s, relu, True), w, gamma, beta, rm, rv)
    # torch reference
    xr = x.clone().requires_grad_(True)
    wr, gr, br = (t.detach().clone().requires_grad_(True) for t in (w, gamma, beta))
    rr = res.clone().requires_grad_(True) if with_res else None
    rm2, rv2 = torch.zeros(Cout, device=d), torch.ones(Cout, device=d)
    o = F.batch_norm(F.conv2d(xr, wr, stride=s, padding=k // 2), rm2, rv2, gr, br, True, 0.1, 1e-5)
    if with_res:
        o = o + rr
    if relu:
        o = F.relu(o)
    assert rel(_nchw(y, R, Ho, Wo), o) < 2e-4
    assert torch.allclose(rm, rm2, rtol=1e-3, atol=1e-5) and torch.allclose(rv, rv2, rtol=1e-3, atol=1e-5)
    g = torch.randn_like(o)
    o.backward(g)
    y.backward(_cl(g))
    assert rel(w.grad, wr.grad) < 2e-3
    assert rel(gamma.grad, gr.grad) < 2e-3 and rel(beta.grad, br.grad) < 2e-3
    if Cin > 1:
        assert rel(_nchw(xin.grad, R, H, W), xr.grad) < 2e-3
    if with_res:
        assert rel(_nchw(rin.grad, R, Ho, Wo), rr.grad) < 1e-5


def test_tstp_matches_torch():
    from wesep_amd import functional_resnet as FR
    d = _cuda()
    torch.manual_seed(1)
    R, C, Fq, T = 3, 8, 5, 17
    x = torch.randn(R, C, Fq, T, device=d) * 2 + 1
    xin = _cl(x).requires_grad_(True)
    s = FR.TstpFn.apply(xin, (R, Fq, T))
    xr = x.clone().requires_grad_(True)
    ref = torch.cat((xr.mean(-1).flatten(1), torch.sqrt(torch.var(xr, dim=-1) + 1e-7).flatten(1)), 1)
    assert rel(s, ref) < 1e-5
    g = torch.randn_like(ref)
    ref.backward(g)
    s.backward(g)
    assert rel(_nchw(xin.grad, R, Fq, T), xr.grad) < 1e-4


@pytest.mark.parametrize("gemm,tol", [("f32", 2e-3), ("bf16x3", 2e-3)])
def test_resnet18_matches_oracle(monkeypatch, gemm, tol):
    """ResNet18 (BasicBlock, m_channels 32), 16 mel bins, 64 frames: embedding, every parameter gradient and the
    BatchNorm running statistics against the restatement, under a fixed random linear functional of the embedding.
    Tolerance 2e-3 per tensor in both product modes, on the SAME linear region: the restatement's ReLUs take the
    device's masks (round 2 compared across regions at 3e-2: with the exact-fp32 kernels all but ONE element of every
    activation gradient agreed to 5e-6 -- that element sat on a ReLU kink, pre-activation within 1e-6 of zero, mask
    flipped, and alone carried 1.1 % of the gradient norm of everything below the last stage)."""
    from oracle import resnet_oracle as RO
    from wesep_amd.models.resnet import get_speaker_model
    monkeypatch.setenv("WESEP_GEMM", gemm)
    d = _cuda()
    kw = dict(num_blocks=RO.NUM_BLOCKS["ResNet18"], m=32, feat_dim=16, embed_dim=64)
    params = RO.synth_params(5, **kw)
    model = get_speaker_model("ResNet18")(feat_dim=16, embed_dim=64, pooling_func="TSTP", two_emb_layer=False)
    assert list(model.state_dict().keys()) == list(params.keys())
    model.load_state_dict(params, strict=True)
    model = model.to(d).train()
    g = torch.Generator().manual_seed(9)
    x = torch.randn(3, 64, 16, generator=g)
    probe = torch.randn(3, 64, generator=g)
    masks = _record_relu_masks(monkeypatch)
    zero, emb = model(x.to(d))
    assert float(zero) == 0.0
    (emb * probe.to(d)).sum().backward()
    p = {k: (v.clone() if RO.is_buffer(k) else v.clone().requires_grad_(True)) for k, v in params.items()}
    bufs = {}
    ref = RO.resnet_forward(p, x, num_blocks=kw["num_blocks"], m=32, new_buffers=bufs, relu_masks=masks)
    (ref * probe).sum().backward()
    assert rel(emb, ref) < 1e-3
    sd = model.state_dict()
    for k, v in bufs.items():
        assert torch.allclose(sd[k].cpu(), v, rtol=2e-3, atol=1e-5), k
    assert int(sd["bn1.num_batches_tracked"]) == 1
    gn = max(float(v.grad.norm()) for k, v in p.items() if not RO.is_buffer(k))
    bad = []
    for k, prm in model.named_parameters():
        err = float((prm.grad.detach().cpu().double() - p[k].grad.double()).norm())
        if err > tol * float(p[k].grad.norm()) + 1e-6 * gn:
            bad.append((k, err / float(p[k].grad.norm())))
    assert not bad, bad[:10]


def test_bottleneck_resnet_with_two_emb_layers_matches_oracle():
    """Bottleneck blocks (ResNet50 / 101 / 152 geometry: 1x1 - 3x3(stride) - 1x1, expansion 4; block counts (1, 2, 1, 1)
    keep the oracle short) and two_emb_layer=True: both embeddings, parameter gradients and running statistics against
    the restatement.  Same tolerances as ResNet18 above (ReLU kinks between fp32 and split-bf16 products)."""
    from oracle import resnet_oracle as RO
    from wesep_amd.models import resnet as MR
    d = _cuda()
    nb = (1, 2, 1, 1)
    kw = dict(num_blocks=nb, m=32, feat_dim=16, embed_dim=64, bottleneck=True, two_emb_layer=True)
    params = RO.synth_params(15, **kw)
    model = MR.ResNet(MR.Bottleneck, list(nb), feat_dim=16, embed_dim=64, pooling_func="TSTP", two_emb_layer=True)
    model.load_state_dict(params, strict=True)
    model = model.to(d).train()
    g = torch.Generator().manual_seed(19)
    x, probe = torch.randn(8, 64, 16, generator=g), torch.randn(8, 64, generator=g)
    ea, eb = model(x.to(d))
    ((ea + eb) * probe.to(d)).sum().backward()
    p = {k: (v.clone() if RO.is_buffer(k) else v.clone().requires_grad_(True)) for k, v in params.items()}
    bufs = {}
    ra, rb = RO.resnet_forward(p, x, num_blocks=nb, m=32, new_buffers=bufs, bottleneck=True, two_emb_layer=True)
    ((ra + rb) * probe).sum().backward()
    assert rel(ea, ra) < 1e-3 and rel(eb, rb) < 1e-3
    sd = model.state_dict()
    for k, v in bufs.items():
        assert torch.allclose(sd[k].cpu(), v, rtol=2e-3, atol=1e-5), k
    gn = max(float(v.grad.norm()) for k, v in p.items() if not RO.is_buffer(k))
    bad = []
    for k, prm in model.named_parameters():
        err = float((prm.grad.detach().cpu().double() - p[k].grad.double()).norm())
        if err > 3e-2 * float(p[k].grad.norm()) + 1e-3 * gn:
            bad.append((k, err / float(p[k].grad.norm())))
    assert not bad, bad[:10]


def test_bsrnn_joint_training_with_resnet34_runs_and_matches_oracle():
    """The shipped configuration (confs/bsrnn.yaml: joint_training, ResNet34 on 80-d fbank, multiply fusion):
    separated waveform against oracle(ResNet restatement -> BSRNN oracle), and gradients reach the speaker encoder."""
    from oracle import bsrnn_oracle as O
    from oracle import resnet_oracle as RO
    from wesep_amd.models import get_model
    d = _cuda()
    cfg = O.BSRNNConfig(num_repeat=1, spk_fuse_type="multiply", multi_fuse=False)
    model = get_model("BSRNN")(num_repeat=1, spk_fuse_type="multiply", multi_fuse=False, use_spk_transform=False,
                               joint_training=True, spk_model="ResNet34", spk_feat=True,
                               spk_args=dict(feat_dim=80, embed_dim=256, pooling_func="TSTP", two_emb_layer=False))
    sep = O.synth_params(cfg, 3)
    spk = RO.synth_params(4, prefix="spk_model.")
    model.load_state_dict({**spk, **sep}, strict=True)
    model = model.to(d).train()
    wav, tgt, _ = O.synth_batch(2, 3000, 3)
    fbank = torch.randn(2, 40, 80, generator=torch.Generator().manual_seed(8))
    est, second = model(wav.to(d), fbank.to(d))
    assert tuple(second.shape) == (2, 256)          # pred_linear = Identity without multi_task (bsrnn.py:357)
    emb = RO.resnet_forward({k: v.clone() for k, v in spk.items()}, fbank, prefix="spk_model.")
    ref = O.bsrnn_forward(sep, cfg, wav, emb)
    assert rel(est, ref) < 1e-3
    from wesep_amd.utils.losses import parse_loss
    parse_loss("SISDR")[0](est, tgt.to(d)).backward()
    g = model.spk_model.conv1.weight.grad
    assert g is not None and torch.isfinite(g).all() and float(g.norm()) > 0
    with pytest.raises(NotImplementedError):
        get_model("BSRNN")(joint_training=True, spk_model="ResNet34", spk_feat=False, feat_type="other",
                           spk_args=dict(feat_dim=80, embed_dim=256, pooling_func="TSTP", two_emb_layer=False))
    with pytest.raises(NotImplementedError):
        get_model("BSRNN")(joint_training=True, spk_model="XVEC", spk_feat=True,
                           spk_args=dict(feat_dim=80, embed_dim=192, pooling_func="TSTP"))


@pytest.mark.parametrize("pooling", ["TAP", "TSDP", "ASTP"])
def test_resnet_pooling_variants_match_oracle(pooling):
    """`pooling_func` other than TSTP (wespeaker pooling_layers: temporal average / standard deviation / attentive
    statistics on the [R, C * F', T] view): embedding and parameter gradients against the restatement."""
    from oracle import resnet_oracle as RO
    from wesep_amd.models import resnet as MR
    d = _cuda()
    kw = dict(num_blocks=(1, 1, 1, 1), m=32, feat_dim=16, embed_dim=64, pooling=pooling)
    params = RO.synth_params(8, **kw)
    model = MR.ResNet(MR.BasicBlock, [1, 1, 1, 1], feat_dim=16, embed_dim=64, pooling_func=pooling, two_emb_layer=False)
    model.load_state_dict(params, strict=True)
    model = model.to(d).train()
    g = torch.Generator().manual_seed(14)
    x, probe = torch.randn(6, 48, 16, generator=g), torch.randn(6, 64, generator=g)
    _, emb = model(x.to(d))
    (emb * probe.to(d)).sum().backward()
    p = {k: (v.clone() if RO.is_buffer(k) else v.clone().requires_grad_(True)) for k, v in params.items()}
    ref = RO.resnet_forward(p, x, num_blocks=kw["num_blocks"], m=32, pooling=pooling)
    (ref * probe).sum().backward()
    assert rel(emb, ref) < 1e-3
    gn = max(float(v.grad.norm()) for k, v in p.items() if not RO.is_buffer(k))
    for k, prm in model.named_parameters():
        if k == "pool.linear2.bias":       # softmax over T ignores a per-channel shift: the true gradient is zero
            continue
        err = float((prm.grad.detach().cpu().double() - p[k].grad.double()).norm())
        assert err <= 3e-2 * float(p[k].grad.norm()) + 1e-3 * gn, (k, err)      # end to end: ReLU kinks, as above


def test_fbank_frontend_matches_restatement_and_joint_raw_audio_path():
    """PreEmphasis + MelSpectrogram + log + CMN on the device against the torch.stft restatement
    (torchaudio absent: unpinned for the MelSpectrogram half), then the spk_feat=False joint model end to end."""
    from oracle import resnet_oracle as RO
    from wesep_amd.models import get_model
    from wesep_amd.modules.common.frontend import MelSpectrogram, PreEmphasis, fbank_frontend
    d = _cuda()
    g = torch.Generator().manual_seed(2)
    wav = torch.randn(3, 16000, generator=g) * 0.1
    wav[1, 4000:9000] = 0.0                                     # a silent stretch: log floor
    pre, mel = PreEmphasis().to(d), MelSpectrogram().to(d)
    feat = fbank_frontend(wav.to(d), pre, mel)
    ref = RO.fbank_frontend(wav)
    assert feat.shape == ref.shape == (3, 126, 80)
    assert float((feat.cpu() - ref).abs().max()) < 2e-3 and rel(feat, ref) < 1e-4
    model = get_model("BSRNN")(num_repeat=1, spk_fuse_type="multiply", multi_fuse=False, use_spk_transform=False,
                               joint_training=True, spk_model="ResNet18", spk_feat=False,
                               spk_args=dict(feat_dim=80, embed_dim=256, pooling_func="TSTP", two_emb_layer=False))
    keys = list(model.state_dict().keys())
    for k in ("preEmphasis.flipped_filter", "spk_encoder.spectrogram.window", "spk_encoder.mel_scale.fb"):
        assert k in keys, k
    model = model.to(d).train()
    est, _ = model(torch.randn(2, 3000, device=d) * 0.1, wav[:2].to(d))
    assert est.shape == (2, 3000) and torch.isfinite(est).all()


def _record_relu_masks(monkeypatch):
    """Wraps models.resnet._cba: collects the ReLU masks (output > 0) of the device forward in evaluation order, as
    [R, C, F', T'] boolean tensors on the CPU."""
    import wesep_amd.models.resnet as MR
    masks = []
    real = MR._cba

    def cba(x, res, R, H, W, stride, relu, conv, bn, training):
        y = real(x, res, R, H, W, stride, relu, conv, bn, training)
        if relu:
            Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
            masks.append((y.detach() > 0).view(R, Ho, Wo, -1).permute(0, 3, 1, 2).cpu())
        return y

    monkeypatch.setattr(MR, "_cba", cba)
    return masks


@pytest.mark.parametrize("name,Tf,Fq,E", [("ResNet18", 64, 16, 64), ("ResNet34", 96, 80, 256)])
def test_resnet_gradients_on_the_same_linear_region(monkeypatch, name, Tf, Fq, E):
    """Every parameter gradient of the speaker encoder against the restatement at 2e-3 -- ResNet34 at the recipe's
    geometry (80 mel bins, 256-d embedding, confs/bsrnn.yaml:58-64).  The restatement is differentiated on the SAME
    linear region as the device forward (its ReLUs use the device's masks, oracle.resnet_oracle.resnet_forward
    relu_masks): what remains is arithmetic, not the handful of pre-activations within rounding distance of zero whose
    flipped masks dominate test_resnet18_matches_oracle's 3e-2."""
    from oracle import resnet_oracle as RO
    from wesep_amd.models.resnet import get_speaker_model
    d = _cuda()
    kw = dict(num_blocks=RO.NUM_BLOCKS[name], m=32, feat_dim=Fq, embed_dim=E)
    params = RO.synth_params(7, **kw)
    model = get_speaker_model(name)(feat_dim=Fq, embed_dim=E, pooling_func="TSTP", two_emb_layer=False)
    model.load_state_dict(params, strict=True)
    model = model.to(d).train()
    g = torch.Generator().manual_seed(13)
    x = torch.randn(3, Tf, Fq, generator=g)
    probe = torch.randn(3, E, generator=g)
    masks = _record_relu_masks(monkeypatch)
    _, emb = model(x.to(d))
    (emb * probe.to(d)).sum().backward()
    torch.cuda.synchronize()
    p = {k: (v.clone() if RO.is_buffer(k) else v.clone().requires_grad_(True)) for k, v in params.items()}
    ref = RO.resnet_forward(p, x, num_blocks=kw["num_blocks"], m=32, relu_masks=masks)
    (ref * probe).sum().backward()
    assert rel(emb, ref) < 1e-3
    per = {k: rel(prm.grad, p[k].grad) for k, prm in model.named_parameters()}
    worst = max(per, key=per.get)
    print(f"{name}: emb rel {rel(emb, ref):.2e}; worst gradient {per[worst]:.2e} ({worst}); "
          f"median {sorted(per.values())[len(per) // 2]:.2e}")
    assert per[worst] < 2e-3, (worst, per[worst])
