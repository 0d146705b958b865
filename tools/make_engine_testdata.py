"""Writes runtime/_testdata/: a small deterministic joint pBSRNN (1 repeat, multiply fusion, ResNet18 on kaldi fbank)
as a weight container, three wav files and a wav.scp, plus the CPU oracle's expected outputs (expected.npz) -- the
inputs of a Python-free hardware check of the native runtime:

    runtime/separate_main --wav_scp runtime/_testdata/wav.scp --model runtime/_testdata/m.wsw \
                          --output_dir <dir> --raw_out
    python tools/make_engine_testdata.py --check <dir>

The expectation chains oracle/fbank_oracle.py (kaldi fbank + CMN, pinned to the reference's C++ front-end),
oracle/resnet_oracle.py (eval mode) and oracle/bsrnn_oracle.py (pinned to the reference)."""
import argparse
import os
import sys
import wave

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "runtime", "_testdata")
KW = dict(num_repeat=1, spk_fuse_type="multiply", multi_fuse=False)
SEED, N_MIX, N_E1, N_E2 = 77, 24000, 32000, 36000


def params():
    from oracle import bsrnn_oracle as O
    from oracle import resnet_oracle as RO
    cfg = O.BSRNNConfig(**KW)
    p = dict(O.synth_params(cfg, SEED))
    spk = RO.synth_params(SEED + 1, num_blocks=RO.NUM_BLOCKS["ResNet18"], prefix="spk_model.")
    g = torch.Generator().manual_seed(SEED + 2)
    for k in spk:                                   # non-trivial running statistics for the eval-mode BatchNorm
        if k.endswith("running_mean"):
            spk[k] = 0.2 * torch.randn(spk[k].shape, generator=g)
        elif k.endswith("running_var"):
            spk[k] = 0.5 + torch.rand(spk[k].shape, generator=g)
    p.update(spk)
    return cfg, p


def signals():
    rng = np.random.default_rng(SEED)
    t = np.arange(N_E2) / 16000.0
    def voice(f0, n):
        x = sum(0.25 / (h + 1) * np.sin(2 * np.pi * f0 * (h + 1) * t[:n] + rng.uniform(0, 6.28)) for h in range(5))
        return x * (0.6 + 0.4 * np.sin(2 * np.pi * 2.5 * t[:n])) + 0.02 * rng.standard_normal(n)
    s1, s2 = voice(120.0, N_E2), voice(190.0, N_E2)
    mix = 0.5 * (s1[:N_MIX] + s2[:N_MIX])
    to16 = lambda x: np.clip(np.round(x * 20000), -32768, 32767).astype(np.int16)
    return to16(mix), to16(s1[:N_E1]), to16(s2)


def write_wav(path, x):
    with wave.open(path, "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(16000)
        w.writeframes(x.tobytes())


def expected(cfg, p, mix16, e1, e2):
    from oracle import bsrnn_oracle as O
    from oracle import fbank_oracle as FB
    from oracle import resnet_oracle as RO
    n_enroll = min(len(e1), len(e2))
    enroll = np.stack([e1[:n_enroll], e2[:n_enroll]]).astype(np.float32) / 32768.0
    fb = FB.apply_cmvn(FB.compute_fbank(enroll, dither=0.0)).astype(np.float32)
    with torch.no_grad():
        emb = RO.resnet_forward(p, torch.from_numpy(fb), num_blocks=RO.NUM_BLOCKS["ResNet18"], prefix="spk_model.",
                                training=False)
        wav = torch.from_numpy(mix16.astype(np.float32) / 32768.0).repeat(2, 1)
        est = O.bsrnn_forward(p, cfg, wav, emb)
    return est.numpy(), emb.numpy(), fb


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", help="directory with separate_main --raw_out outputs: compare with expected.npz")
    args = ap.parse_args()
    if args.check:
        g = np.load(os.path.join(OUT, "expected.npz"))
        for k in (1, 2):
            got = np.fromfile(os.path.join(args.check, f"utt1-spk{k}.f32"), dtype=np.float32)
            ref = g["est"][k - 1]
            rel = np.linalg.norm(got - ref) / np.linalg.norm(ref)
            print(f"spk{k}: n={got.size} finite={bool(np.isfinite(got).all())} rel error vs CPU oracle {rel:.3e} "
                  f"(|ref| rms {np.sqrt((ref ** 2).mean()):.3e})")
        return
    from wesep_amd.bin.export_engine import export_engine
    from wesep_amd.models import get_model
    os.makedirs(OUT, exist_ok=True)
    cfg, p = params()
    model = get_model("BSRNN")(num_repeat=1, spk_fuse_type="multiply", multi_fuse=False, use_spk_transform=False,
                               joint_training=True, spk_model="ResNet18", spk_feat=True,
                               spk_args=dict(feat_dim=80, embed_dim=256, pooling_func="TSTP", two_emb_layer=False))
    model.load_state_dict(p, strict=True)
    export_engine(model, os.path.join(OUT, "m.wsw"))
    mix16, e1, e2 = signals()
    for name, x in (("mix", mix16), ("e1", e1), ("e2", e2)):
        write_wav(os.path.join(OUT, name + ".wav"), x)
    with open(os.path.join(OUT, "wav.scp"), "w") as f:
        f.write("utt1 runtime/_testdata/mix.wav runtime/_testdata/e1.wav runtime/_testdata/e2.wav\n")
    est, emb, fb = expected(cfg, p, mix16, e1, e2)
    np.savez_compressed(os.path.join(OUT, "expected.npz"), est=est, emb=emb, fbank=fb)
    print("wrote", OUT, "est rms", float(np.sqrt((est ** 2).mean())), "fbank", fb.shape)


if __name__ == "__main__":
    main()
