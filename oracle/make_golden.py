"""TEST INFRASTRUCTURE ONLY -- generate `tests/golden/*.npz` from the REAL reference.

Run in the authoring container (where `/root/reference` exists):

    python -m oracle.make_golden

For every case it builds the reference `wesep.models.bsrnn.BSRNN`
(`joint_training=False`), loads the deterministic parameter set
`oracle.bsrnn_oracle.synth_params(cfg, seed)` into it with a strict
`load_state_dict`, runs forward + restated SI-SDR loss + backward on the
synthetic batch `synth_batch(R, T, seed)`, and stores: the inputs, the
estimated waveform, the loss, and for every parameter gradient its L2 norm
plus its first 16 values (full gradient for tensors <= 4096 elements).
The fixtures pin `oracle/bsrnn_oracle.py` (tests/test_oracle_golden.py) and,
through it, the HIP path.  The reference cannot travel to the GPU box; the
fixtures can.
"""
import os
import sys

import numpy as np
import torch

from oracle import bsrnn_oracle as O
from oracle import convtasnet_oracle as CT
from oracle import dpccn_oracle as DP
from oracle import tfgridnet_oracle as TG
from oracle.ref_import import import_reference

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")

CASES = {
    # name: (cfg kwargs, R, T, seed)
    "bsrnn_multiply_r2_t4000": (dict(num_repeat=1, spk_fuse_type="multiply", multi_fuse=False), 2, 4000, 11),
    "bsrnn_film_multi_r2_t3000": (dict(num_repeat=2, spk_fuse_type="FiLM", multi_fuse=True), 2, 3000, 12),
    "bsrnn_additive_xform_r4_t2048": (dict(num_repeat=1, spk_fuse_type="additive", multi_fuse=False,
                                           use_spk_transform=True), 4, 2048, 13),
    "bsrnn_concat_r2_t2500": (dict(num_repeat=1, spk_fuse_type="concat", multi_fuse=True), 2, 2500, 14),
}


def run_case(name, kw, R, T, seed):
    get_model = import_reference()
    cfg = O.BSRNNConfig(**kw)
    ref = get_model("BSRNN")(
        spk_emb_dim=cfg.spk_emb_dim, sr=cfg.sr, win=cfg.win, stride=cfg.stride,
        feature_dim=cfg.feature_dim, num_repeat=cfg.num_repeat,
        use_spk_transform=cfg.use_spk_transform, spk_fuse_type=cfg.spk_fuse_type,
        multi_fuse=cfg.multi_fuse, joint_training=False)
    params = O.synth_params(cfg, seed)
    ref_sd = ref.state_dict()
    assert list(ref_sd.keys()) == list(params.keys()), "oracle param_shapes() order != reference state_dict"
    for k in ref_sd:
        assert tuple(ref_sd[k].shape) == tuple(params[k].shape), k
    ref.load_state_dict(params, strict=True)
    ref.train()
    wav, tgt, emb = O.synth_batch(R, T, seed)
    est, _ = ref(wav, emb)
    loss = O.sisdr_loss(est, tgt)
    loss.backward()
    # independent cross-check of the loss restatement: reference's numpy SI-SNR metric
    sisnr_np = np.mean([O.cal_sisnr_np(tgt[r].numpy(), est[r].detach().numpy()) for r in range(R)])
    assert abs(-float(loss) - sisnr_np) < 1e-3, (float(loss), sisnr_np)
    out = {
        "wav": wav.numpy(), "tgt": tgt.numpy(), "emb": emb.numpy(),
        "est": est.detach().numpy(), "loss": np.float64(loss.item()),
        "param_checksum": np.float64(sum(float(v.double().abs().sum()) for v in params.values())),
    }
    names = []
    for k, prm in ref.named_parameters():
        g = prm.grad.detach().reshape(-1)
        names.append(k)
        out["gnorm/" + k] = np.float64(g.double().norm().item())
        out["ghead/" + k] = g[:16].numpy().copy()
        if g.numel() <= 4096:
            out["gfull/" + k] = g.numpy().copy()
    out["names"] = np.array(names)
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: loss={loss.item():.6f} sisnr_np={sisnr_np:.6f} est_rms={est.pow(2).mean().sqrt().item():.4e}")


TASNET_CASES = {
    # name: (ConvTasNetConfig kwargs, rows, T (multiple of the stride), seed)
    "convtasnet_gln_r2_t1600": (dict(N=32, L=20, B=32, H=64, P=3, X=3, R=2), 2, 1600, 21),
    "convtasnet_cln_xform_r4_t2000": (dict(N=16, L=20, B=24, H=40, P=3, X=2, R=1, norm="cLN",
                                           use_spk_transform=True), 4, 2000, 22),
    "convtasnet_gln_l16_r2_t1200": (dict(N=24, L=16, B=16, H=32, P=3, X=4, R=1), 2, 1200, 23),
    # the other speaker-fusion types of FuseSeparation (separation.py:116-135): SpeakerFuseLayer - PReLU - norm - blocks
    "convtasnet_multiply_r2_t1600": (dict(N=32, L=20, B=32, H=64, P=3, X=3, R=2, spk_fuse_type="multiply"), 2, 1600, 25),
    "convtasnet_additive_cln_r2_t1600": (dict(N=16, L=20, B=24, H=40, P=3, X=2, R=2, norm="cLN",
                                              spk_fuse_type="additive"), 2, 1600, 26),
    "convtasnet_film_r2_t1600": (dict(N=32, L=20, B=32, H=64, P=3, X=2, R=2, spk_fuse_type="FiLM"), 2, 1600, 27),
    "convtasnet_concat_r2_t1600": (dict(N=32, L=20, B=32, H=64, P=3, X=2, R=1, spk_fuse_type="concat"), 2, 1600, 28),
    # SpEx+ joint mode: enrollment waveform through the shared encoder + ResNet4SpExplus (N must be 256),
    # multi-task speaker head; loss = .8/.1/.1 SI-SDR + .5 CE; BatchNorm running statistics are pinned too
    "spexplus_joint_r4_t1600": (dict(N=256, L=20, B=32, H=48, P=3, X=2, R=2, joint_training=True,
                                     multi_task=True, spksInTrain=11), 4, 1600, 24),
}
ENROLL_LEN = 2400


def run_tasnet_case(name, kw, R, T, seed):
    """Conv-TasNet / SpEx+ with fixed embeddings (`joint_training=False`), multi-scale SI-SDR loss."""
    get_model = import_reference()
    cfg = CT.ConvTasNetConfig(**kw)
    ref = get_model("ConvTasNet")(
        N=cfg.N, L=cfg.L, B=cfg.B, H=cfg.H, P=cfg.P, X=cfg.X, R=cfg.R, spk_emb_dim=cfg.spk_emb_dim,
        norm=cfg.norm, activate="relu", causal=False, skip_con=False, spk_fuse_type=cfg.spk_fuse_type,
        multi_fuse=cfg.multi_fuse, use_spk_transform=cfg.use_spk_transform, encoder_type="Multi",
        decoder_type="Multi", joint_training=cfg.joint_training, multi_task=cfg.multi_task,
        spksInTrain=cfg.spksInTrain, spk_feat=False, feat_type="consistent")
    params = CT.synth_params(cfg, seed)
    ref_sd = ref.state_dict()
    assert list(ref_sd.keys()) == list(params.keys()), "oracle param_shapes() order != reference state_dict"
    for k in ref_sd:
        assert tuple(ref_sd[k].shape) == tuple(params[k].shape), k
    ref.load_state_dict(params, strict=True)
    ref.train()
    wav, tgt, emb = O.synth_batch(R, T, seed)
    label = None
    if cfg.joint_training:
        emb, label = CT.synth_enrollment(R, ENROLL_LEN, cfg.spksInTrain, seed)
    ests = ref(wav, emb)
    assert all(e.shape == tgt.shape for e in ests[:3]), [tuple(e.shape) for e in ests]
    loss = CT.spexplus_loss(ests, tgt, label) if cfg.multi_task else CT.multiscale_sisdr_loss(ests, tgt)
    loss.backward()
    out = {
        "wav": wav.numpy(), "tgt": tgt.numpy(), "emb": emb.numpy(), "loss": np.float64(loss.item()),
        "param_checksum": np.float64(sum(float(v.double().abs().sum()) for v in params.values())),
    }
    for i, e in enumerate(ests[:3]):
        out[f"est{i + 1}"] = e.detach().numpy()
    if cfg.multi_task:
        out["logits"] = ests[3].detach().numpy()
        out["label"] = label.numpy()
    for k, v in ref.state_dict().items():          # BatchNorm running statistics after this one training step
        if k.endswith(("running_mean", "running_var")):
            out["buf/" + k] = v.numpy().copy()
    names = []
    for k, prm in ref.named_parameters():
        g = prm.grad.detach().reshape(-1)
        names.append(k)
        out["gnorm/" + k] = np.float64(g.double().norm().item())
        out["ghead/" + k] = g[:16].numpy().copy()
        if g.numel() <= 4096:
            out["gfull/" + k] = g.numpy().copy()
    out["names"] = np.array(names)
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: loss={loss.item():.6f} est1_rms={ests[0].pow(2).mean().sqrt().item():.4e}")


TASNET_VARIANT_CASES = {
    # name: (reference ConvTasNet constructor kwargs, rows, T, seed).  The constructor options outside the oracle's
    # restatement (convtasnet.py:16-46: causal, skip_con, norm='BN', Deep / plain encoder-decoder pairs, sigmoid masks):
    # the fixture carries the PARAMETERS too (the reference's own initialisation under the seed, norm gains / PReLU slopes /
    # BatchNorm buffers perturbed to non-trivial values), so the HIP path is held to the reference directly.
    "convtasnet_plain_skip_r2_t1600": (dict(N=32, L=20, B=24, H=48, P=3, X=3, R=2, skip_con=True, encoder_type=None,
                                            decoder_type=None), 2, 1600, 51),
    "convtasnet_deep_causal_cln_r2_t1600": (dict(N=32, L=20, B=24, H=48, P=3, X=3, R=1, norm="cLN", causal=True,
                                                 encoder_type="Deep", decoder_type="Deep", activate="sigmoid",
                                                 spk_fuse_type="multiply"), 2, 1600, 52),
    "convtasnet_multi_bn_skip_r4_t1600": (dict(N=32, L=20, B=32, H=64, P=3, X=2, R=2, norm="BN", skip_con=True), 4, 1600, 53),
    "convtasnet_multi_causal_gln_r2_t1600": (dict(N=32, L=20, B=32, H=64, P=3, X=3, R=1, causal=True), 2, 1600, 54),
    "convtasnet_plain_bn_film_r4_t1200": (dict(N=32, L=16, B=24, H=40, P=3, X=2, R=2, norm="BN", encoder_type=None,
                                               decoder_type=None, spk_fuse_type="FiLM"), 4, 1200, 55),
}


def variant_loss(out, tgt):
    """Loss of a variant case: multi-scale SI-SDR for the Multi decoder's three estimates, plain SI-SDR otherwise
    (the plain ConvTrans1D decoder returns [R, 1, T], convs.py:27-41)."""
    if isinstance(out, (list, tuple)):
        return CT.multiscale_sisdr_loss(list(out), tgt)
    est = out.reshape(out.shape[0], -1)
    n = min(est.shape[-1], tgt.shape[-1])
    return O.sisdr_loss(est[:, :n], tgt[:, :n])


def run_tasnet_variant_case(name, kw, R, T, seed):
    get_model = import_reference()
    torch.manual_seed(seed)
    ref = get_model("ConvTasNet")(**{**dict(use_spk_transform=False, joint_training=False), **kw})
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for k, v in ref.state_dict().items():
            last = k.split(".")[-1]
            if k.endswith("num_batches_tracked"):
                continue
            if last == "running_mean":
                v.copy_(0.1 * torch.randn(v.shape, generator=g))
            elif last == "running_var":
                v.copy_(1.0 + 0.2 * torch.rand(v.shape, generator=g))
            elif v.numel() == 1 and last == "weight":                   # PReLU slope
                v.copy_(0.25 + 0.05 * torch.randn(v.shape, generator=g))
            elif ("norm" in k.lower() or "LayerN" in k or ".ln." in k) and v.dim() <= 2 and v.shape[-1] in (1, v.numel()):
                v.copy_((1.0 + 0.1 * torch.randn(v.shape, generator=g)) if last == "weight"
                        else 0.1 * torch.randn(v.shape, generator=g))
            elif last == "bias":
                v.copy_(0.05 * torch.randn(v.shape, generator=g))
    ref.train()
    params = {k: v.detach().clone() for k, v in ref.state_dict().items()}
    wav, tgt, emb = O.synth_batch(R, T, seed)
    res = ref(wav, emb)
    loss = variant_loss(res, tgt)
    loss.backward()
    ests = list(res) if isinstance(res, (list, tuple)) else [res]
    out = {"wav": wav.numpy(), "tgt": tgt.numpy(), "emb": emb.numpy(), "loss": np.float64(loss.item())}
    for i, e in enumerate(ests):
        out[f"est{i + 1}"] = e.detach().numpy()
    for k, v in params.items():
        out["param/" + k] = v.numpy()
    for k, v in ref.state_dict().items():          # BatchNorm running statistics after this one training step
        if k.endswith(("running_mean", "running_var")):
            out["buf/" + k] = v.numpy().copy()
    for k, prm in ref.named_parameters():
        if prm.grad is None:       # skip_con: the last block's residual `Output` conv is never used (separation.py:43-49)
            out["gnorm/" + k] = np.float64(-1.0)
            continue
        gr = prm.grad.detach().reshape(-1)
        out["gnorm/" + k] = np.float64(gr.double().norm().item())
        if gr.numel() <= 4096:
            out["gfull/" + k] = gr.numpy().copy()
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: loss={loss.item():.6f} est1_rms={ests[0].pow(2).mean().sqrt().item():.4e} "
          f"params={sum(v.numel() for v in params.values())}")


DPCCN_CASES = {
    # name: (DPCCNConfig kwargs, rows, T, seed) -- T >= 4352: the pyramid pooling needs >= 32 frames
    "dpccn_multiply_r2_t4480": (dict(tcn_blocks=3, tcn_layers=1), 2, 4480, 31),
    "dpccn_concat_xform_r2_t4352": (dict(tcn_blocks=2, tcn_layers=2, spk_fuse_type="concat", use_spk_transform=True),
                                    2, 4352, 32),
    "dpccn_additive_xform_r2_t4608": (dict(tcn_blocks=2, tcn_layers=2, spk_fuse_type="additive",
                                           use_spk_transform=True), 2, 4608, 33),
    "dpccn_film_r2_t4352": (dict(tcn_blocks=2, tcn_layers=1, spk_fuse_type="FiLM"), 2, 4352, 34),
    "dpccn_causal_r2_t4480": (dict(tcn_blocks=3, tcn_layers=1, causal=True), 2, 4480, 35),
}


def run_dpccn_case(name, kw, R, T, seed):
    """DPCCN with fixed embeddings (`joint_training=False`), SI-SDR loss on the estimate."""
    get_model = import_reference()
    cfg = DP.DPCCNConfig(**kw)
    ref = get_model("DPCCN")(win=cfg.win, stride=cfg.stride, spk_emb_dim=cfg.spk_emb_dim,
                             use_spk_transform=cfg.use_spk_transform, spk_fuse_type=cfg.spk_fuse_type,
                             feature_dim=cfg.feature_dim, tcn_dims=cfg.tcn_dims, tcn_blocks=cfg.tcn_blocks,
                             tcn_layers=cfg.tcn_layers, pool_size=cfg.pool_size, causal=cfg.causal, joint_training=False)
    params = DP.synth_params(cfg, seed)
    ref_sd = ref.state_dict()
    assert list(ref_sd.keys()) == list(params.keys()), "oracle param_shapes() order != reference state_dict"
    for k in ref_sd:
        assert tuple(ref_sd[k].shape) == tuple(params[k].shape), k
    ref.load_state_dict(params, strict=True)
    ref.train()
    wav, tgt, emb = O.synth_batch(R, T, seed)
    est, _ = ref(wav, emb)
    loss = O.sisdr_loss(est, tgt)
    loss.backward()
    out = {"wav": wav.numpy(), "tgt": tgt.numpy(), "emb": emb.numpy(), "est": est.detach().numpy(),
           "loss": np.float64(loss.item()),
           "param_checksum": np.float64(sum(float(v.double().abs().sum()) for v in params.values()))}
    names = []
    for k, prm in ref.named_parameters():
        g = prm.grad.detach().reshape(-1)
        names.append(k)
        out["gnorm/" + k] = np.float64(g.double().norm().item())
        out["ghead/" + k] = g[:16].numpy().copy()
    out["names"] = np.array(names)
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: loss={loss.item():.6f} est_rms={est.pow(2).mean().sqrt().item():.4e}")


TFGRIDNET_CASES = {
    # name: (TFGridNetConfig kwargs, rows, T, seed)
    "tfgridnet_ks4_r2_t1600": (dict(n_layers=2, lstm_hidden_units=24, emb_dim=16, attn_approx_qk_dim=260), 2, 1600, 41),
    "tfgridnet_ks1_additive_r2_t1280": (dict(n_layers=1, lstm_hidden_units=16, emb_dim=8, emb_ks=1, emb_hs=1,
                                             attn_n_head=2, attn_approx_qk_dim=260, spk_fuse_type="additive",
                                             use_spk_transform=True), 2, 1280, 42),
    "tfgridnet_ks1_film_r2_t1280": (dict(n_layers=2, lstm_hidden_units=16, emb_dim=8, emb_ks=1, emb_hs=1,
                                         attn_n_head=2, attn_approx_qk_dim=260, spk_fuse_type="FiLM"), 2, 1280, 43),
    "tfgridnet_ks1_concat_r2_t1280": (dict(n_layers=2, lstm_hidden_units=16, emb_dim=8, emb_ks=1, emb_hs=1,
                                           attn_n_head=2, attn_approx_qk_dim=260, spk_fuse_type="concat"), 2, 1280, 44),
    # two output sources from a three-microphone mixture [R, T, 3] (tfgridnet.py:173,192-194,216-244,280-300)
    "tfgridnet_ks1_srcs2_mics3_r2_t1280": (dict(n_layers=1, lstm_hidden_units=16, emb_dim=8, emb_ks=1, emb_hs=1,
                                                attn_n_head=2, attn_approx_qk_dim=260, n_srcs=2, n_imics=3), 2, 1280, 45),
}


def tfgridnet_batch(cfg, R, T, seed):
    """(mixture, target, embedding) of a TF-GridNet case: [R, T] / [R, T]; with n_imics = M > 1 the mixture is [R, T, M]
    (microphone m = the synthetic mixture plus a delayed, scaled copy), with n_srcs = S > 1 the target is [R, S, T]."""
    wav, tgt, emb = O.synth_batch(R, T, seed)
    if cfg.n_imics > 1:
        wav = torch.stack([wav if m == 0 else 0.8 ** m * torch.roll(wav, 3 * m, 1) + 0.05 * torch.roll(tgt, 7 * m, 1)
                           for m in range(cfg.n_imics)], 2).contiguous()
    if cfg.n_srcs > 1:
        tgt = torch.stack([tgt if s == 0 else torch.roll(tgt, 11 * s, 1) * (1.0 - 0.2 * s) for s in range(cfg.n_srcs)], 1)
    return wav, tgt, emb


def run_tfgridnet_case(name, kw, R, T, seed):
    """TF-GridNet with fixed embeddings (`joint_training=False`), SI-SDR loss on the estimate."""
    get_model = import_reference()
    cfg = TG.TFGridNetConfig(**kw)
    ref = get_model("TFGridNet")(n_fft=cfg.n_fft, stride=cfg.stride, n_layers=cfg.n_layers,
                                 lstm_hidden_units=cfg.lstm_hidden_units, attn_n_head=cfg.attn_n_head,
                                 attn_approx_qk_dim=cfg.attn_approx_qk_dim, emb_dim=cfg.emb_dim, emb_ks=cfg.emb_ks,
                                 emb_hs=cfg.emb_hs, eps=cfg.eps, spk_emb_dim=cfg.spk_emb_dim,
                                 use_spk_transform=cfg.use_spk_transform, spk_fuse_type=cfg.spk_fuse_type,
                                 n_srcs=cfg.n_srcs, n_imics=cfg.n_imics, joint_training=False)
    params = TG.synth_params(cfg, seed)
    ref_sd = ref.state_dict()
    assert list(ref_sd.keys()) == list(params.keys()), "oracle param_shapes() order != reference state_dict"
    for k in ref_sd:
        assert tuple(ref_sd[k].shape) == tuple(params[k].shape), k
    ref.load_state_dict(params, strict=True)
    ref.train()
    wav, tgt, emb = tfgridnet_batch(cfg, R, T, seed)
    est, _ = ref(wav, emb)
    loss = O.sisdr_loss(est, tgt)
    loss.backward()
    out = {"wav": wav.numpy(), "tgt": tgt.numpy(), "emb": emb.numpy(), "est": est.detach().numpy(),
           "loss": np.float64(loss.item()),
           "param_checksum": np.float64(sum(float(v.double().abs().sum()) for v in params.values()))}
    names = []
    for k, prm in ref.named_parameters():
        g = prm.grad.detach().reshape(-1)
        names.append(k)
        out["gnorm/" + k] = np.float64(g.double().norm().item())
    out["names"] = np.array(names)
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: loss={loss.item():.6f} est_rms={est.pow(2).mean().sqrt().item():.4e}")


# ---- kaldi-style enrollment fbank: outputs of the reference's own C++ front-end --------
# name -> (R, T, sample_rate, num_mel_bins, seed)
FBANK_CASES = {
    "fbank_r3_t6400_16k_80": (3, 6400, 16000, 80, 31),
    "fbank_r2_t4000_8k_40": (2, 4000, 8000, 40, 32),
}


def synth_fbank_wave(R, T, sample_rate, seed):
    """Seeded speech-like rows in [-1, 1]: a few harmonics under a slow envelope, noise and a small DC offset."""
    rng = np.random.default_rng(seed)
    t = np.arange(T) / sample_rate
    rows = []
    for r in range(R):
        f0 = 110.0 + 40.0 * r
        x = sum(0.2 / (h + 1) * np.sin(2 * np.pi * f0 * (h + 1) * t + rng.uniform(0, 6.28)) for h in range(6))
        x = x * (0.6 + 0.4 * np.sin(2 * np.pi * 3.0 * t)) + 0.02 * rng.standard_normal(T) + 0.01 * (r - 1)
        rows.append(x)
    return np.stack(rows, 0).astype(np.float32)


def run_fbank_case(name, R, T, sample_rate, nb, seed):
    from oracle import build_ref, fbank_oracle as FB
    assert build_ref.build(), "reference runtime sources not found"
    wav = synth_fbank_wave(R, T, sample_rate, seed)
    feats = np.stack([FB.ref_fbank(row * np.float32(1 << 15), nb, sample_rate) for row in wav], 0)
    cmn = np.stack([FB.ref_fbank(row * np.float32(1 << 15), nb, sample_rate, apply_mean=True) for row in wav], 0)
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), wav=wav, fbank=feats, fbank_cmn=cmn)
    print(f"{name}: feats {feats.shape} mean {feats.mean():.4f}")


# ---- BSRNN_Multi (SSA multi-optimisation): the REAL reference module tree and two-pass
# forward; its two third-party halves (wespeaker ResNet, torchaudio MelSpectrogram -- both absent from this image) are
# stood in for by the restatements of oracle/resnet_oracle.py, so the fixture pins the BSRNN_Multi structure, the
# separator, PreEmphasis and the loss composition, and leaves those two halves unpinned as everywhere else.
# name -> (BSRNNConfig kwargs, spk model, rows, T, enrollment samples, seed)
MULTI_CASES = {
    "bsrnn_multi_r2_t3000": (dict(num_repeat=1, spk_fuse_type="multiply", multi_fuse=False), "ResNet18", 2, 3000,
                             4000, 41),
}
MULTI_LOSS_WEIGHT = (0.4, 0.6)          # bsrnn_multi_optim.yaml: loss_posi [[0, 1]], loss_weight [[0.4, 0.6]]


def multi_embed_fn(params, spk_model):
    """enrollment waveform -> embedding through the restated front-end + speaker encoder (prefix spk_model.)."""
    from oracle import resnet_oracle as RO
    return lambda w: RO.resnet_forward(params, RO.fbank_frontend(w).detach(), num_blocks=RO.NUM_BLOCKS[spk_model],
                                       prefix="spk_model.")


def synth_multi_params(cfg, spk_model, seed):
    from oracle import resnet_oracle as RO
    p = dict(O.synth_params(cfg, seed))
    p.update(RO.synth_params(seed + 1, num_blocks=RO.NUM_BLOCKS[spk_model], prefix="spk_model."))
    return p


def _install_third_party_stand_ins():
    import torch.nn as nn
    from oracle import resnet_oracle as RO

    class StandInResNet(nn.Module):
        """wespeaker ResNet stand-in: parameters under mangled upstream names, forward = the restatement."""

        def __init__(self, num_blocks, feat_dim=80, embed_dim=256, pooling_func="TSTP", two_emb_layer=False):
            super().__init__()
            assert pooling_func == "TSTP" and not two_emb_layer
            self.num_blocks = num_blocks
            shapes = RO.param_shapes(num_blocks=num_blocks, feat_dim=feat_dim, embed_dim=embed_dim)
            self.keys = list(shapes)
            for k, shp in shapes.items():
                if RO.is_buffer(k):
                    self.register_buffer(k.replace(".", "__"), torch.zeros(shp))
                else:
                    self.register_parameter(k.replace(".", "__"), nn.Parameter(torch.zeros(shp)))

        def forward(self, x):
            p = {k: getattr(self, k.replace(".", "__")) for k in self.keys}
            return torch.tensor(0.0), RO.resnet_forward(p, x, num_blocks=self.num_blocks)

    class StandInMelSpectrogram(nn.Module):
        """torchaudio.transforms.MelSpectrogram stand-in (centre, reflect, power 2, HTK, no norm)."""

        def __init__(self, sample_rate, n_fft, win_length, hop_length, f_min, window_fn, n_mels):
            super().__init__()
            assert win_length == n_fft
            self.n_fft, self.hop = n_fft, hop_length
            self.register_buffer("window", window_fn(n_fft))
            self.register_buffer("fb", RO.melscale_fbanks(n_fft // 2 + 1, f_min, float(sample_rate // 2), n_mels,
                                                          sample_rate))

        def forward(self, y):
            spec = torch.stft(y, self.n_fft, self.hop, self.n_fft, window=self.window, center=True,
                              pad_mode="reflect", normalized=False, onesided=True, return_complex=True).abs().pow(2.0)
            return torch.matmul(spec.transpose(-1, -2), self.fb).transpose(-1, -2)

    import_reference()
    import wesep.models.bsrnn_multi_optim as ref_multi
    ref_multi.get_speaker_model = lambda name: (lambda **kw: StandInResNet(RO.NUM_BLOCKS[name], **kw))
    ref_multi.torchaudio.transforms.MelSpectrogram = StandInMelSpectrogram
    return ref_multi


def run_multi_case(name, kw, spk_model, R, T, Tw, seed):
    ref_multi = _install_third_party_stand_ins()
    cfg = O.BSRNNConfig(**kw)
    ref = ref_multi.BSRNN_Multi(
        spk_emb_dim=cfg.spk_emb_dim, sr=cfg.sr, win=cfg.win, stride=cfg.stride, feature_dim=cfg.feature_dim,
        num_repeat=cfg.num_repeat, use_spk_transform=cfg.use_spk_transform, spk_fuse_type=cfg.spk_fuse_type,
        multi_fuse=cfg.multi_fuse, joint_training=True, multi_task=False, spk_model=spk_model, spk_model_init=False,
        spk_model_freeze=False, spk_args=dict(feat_dim=80, embed_dim=cfg.spk_emb_dim, pooling_func="TSTP",
                                              two_emb_layer=False), spk_feat=False, feat_type="consistent")
    params = synth_multi_params(cfg, spk_model, seed)
    sd = ref.state_dict()
    for k in sd:                                    # separator / BN / mask by name; speaker stand-in by mangled name
        if k.startswith("spk_model."):
            src = "spk_model." + k[len("spk_model."):].replace("__", ".")
            sd[k] = params[src].to(sd[k].dtype)
        elif k in params:
            sd[k] = params[k]
    ref.load_state_dict(sd, strict=True)
    loaded = {k for k in sd if k.startswith("spk_model.") or k in params}
    assert len(loaded) == len(params), (len(loaded), len(params))
    ref.train()
    wav, tgt, _ = O.synth_batch(R, T, seed)
    g = torch.Generator().manual_seed(seed + 2)
    enroll = 0.1 * torch.randn(R, Tw, generator=g)
    s, self_s, e1, e2 = ref(wav, enroll)
    loss = MULTI_LOSS_WEIGHT[0] * O.sisdr_loss(s, tgt) + MULTI_LOSS_WEIGHT[1] * O.sisdr_loss(self_s, tgt)
    loss.backward()
    with torch.no_grad():
        assert len(ref(wav, enroll)) == 2           # no-grad mode: the plain BSRNN outputs
    out = {"wav": wav.numpy(), "tgt": tgt.numpy(), "enroll": enroll.numpy(), "est": s.detach().numpy(),
           "self_est": self_s.detach().numpy(), "emb1": e1.detach().numpy(), "emb2": e2.detach().numpy(),
           "loss": np.float64(loss.item()),
           "param_checksum": np.float64(sum(float(v.double().abs().sum()) for v in params.values()))}
    names = []
    for k, prm in ref.named_parameters():
        key = ("spk_model." + k[len("spk_model."):].replace("__", ".")) if k.startswith("spk_model.") else k
        names.append(key)
        out["gnorm/" + key] = np.float64(prm.grad.detach().double().norm().item())
    out["names"] = np.array(names)
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: loss={loss.item():.6f} |s - self_s| / |s| = {float((s - self_s).norm() / s.norm()):.3e}")


if __name__ == "__main__":
    torch.set_num_threads(8)
    only = sys.argv[1:]
    for name, (kw, R, T, seed) in CASES.items():
        if only and name not in only:
            continue
        run_case(name, kw, R, T, seed)
    for name, (kw, R, T, seed) in TASNET_CASES.items():
        if only and name not in only:
            continue
        run_tasnet_case(name, kw, R, T, seed)
    for name, (kw, R, T, seed) in TASNET_VARIANT_CASES.items():
        if only and name not in only:
            continue
        run_tasnet_variant_case(name, kw, R, T, seed)
    for name, (kw, R, T, seed) in DPCCN_CASES.items():
        if only and name not in only:
            continue
        run_dpccn_case(name, kw, R, T, seed)
    for name, (kw, R, T, seed) in TFGRIDNET_CASES.items():
        if only and name not in only:
            continue
        run_tfgridnet_case(name, kw, R, T, seed)
    for name, (R, T, sr, nb, seed) in FBANK_CASES.items():
        if only and name not in only:
            continue
        run_fbank_case(name, R, T, sr, nb, seed)
    for name, (kw, spk_model, R, T, Tw, seed) in MULTI_CASES.items():
        if only and name not in only:
            continue
        run_multi_case(name, kw, spk_model, R, T, Tw, seed)
