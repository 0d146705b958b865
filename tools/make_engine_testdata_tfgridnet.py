"""Writes runtime/_testdata/tfgridnet/: a deterministic joint TF-GridNet (recipe geometry, 2 blocks, multiply fusion,
ResNet18 on kaldi fbank) as a weight container, three wav files and a wav.scp, plus the CPU oracle's expected outputs
(expected.npz) -- the inputs of a Python-free hardware check of the native runtime's TF-GridNet plan (runtime/engine.cc,
arch 3):

    runtime/separate_main --wav_scp runtime/_testdata/tfgridnet/wav.scp --model runtime/_testdata/tfgridnet/m.wsw \
                          --output_dir <dir> --raw_out
    python tools/make_engine_testdata_tfgridnet.py --check <dir>

The expectation chains oracle/fbank_oracle.py (kaldi fbank + CMN, pinned to the reference's C++ front-end),
oracle/resnet_oracle.py (eval mode) and oracle/tfgridnet_oracle.py (pinned to the reference)."""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
OUT = os.path.join(ROOT, "runtime", "_testdata", "tfgridnet")
SEED = 57
KW = dict(n_layers=2, emb_dim=128, emb_ks=1, emb_hs=1, lstm_hidden_units=192)


def params():
    from oracle import resnet_oracle as RO
    from oracle import tfgridnet_oracle as O
    cfg = O.TFGridNetConfig(**KW)
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


def expected(cfg, p, mix16, e1, e2):
    from oracle import fbank_oracle as FB
    from oracle import resnet_oracle as RO
    from oracle import tfgridnet_oracle as O
    n_enroll = min(len(e1), len(e2))
    enroll = np.stack([e1[:n_enroll], e2[:n_enroll]]).astype(np.float32) / 32768.0
    fb = FB.apply_cmvn(FB.compute_fbank(enroll, dither=0.0)).astype(np.float32)
    with torch.no_grad():
        emb = RO.resnet_forward(p, torch.from_numpy(fb), num_blocks=RO.NUM_BLOCKS["ResNet18"], prefix="spk_model.",
                                training=False)
        wav = torch.from_numpy(mix16.astype(np.float32) / 32768.0).repeat(2, 1)
        est = O.tfgridnet_forward(p, cfg, wav, emb)
    est = est[0] if isinstance(est, tuple) else est
    return est.numpy(), emb.numpy(), fb


def main():
    import make_engine_testdata as B
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", help="directory with separate_main --raw_out outputs: compare with expected.npz")
    ap.add_argument("--json", help="with --check: also write the comparison here")
    args = ap.parse_args()
    if args.check:
        g = np.load(os.path.join(OUT, "expected.npz"))
        res = {}
        for k in (1, 2):
            got = np.fromfile(os.path.join(args.check, f"utt1-spk{k}.f32"), dtype=np.float32)
            ref = g["est"][k - 1]
            rel = float(np.linalg.norm(got - ref) / np.linalg.norm(ref)) if got.size == ref.size else float("nan")
            res[f"spk{k}"] = {"n": int(got.size), "finite": bool(np.isfinite(got).all()), "rel_error_vs_cpu_oracle": rel,
                              "ref_rms": float(np.sqrt((ref ** 2).mean()))}
            print(f"spk{k}: n={got.size} finite={res[f'spk{k}']['finite']} rel error vs CPU oracle {rel:.3e} "
                  f"(|ref| rms {res[f'spk{k}']['ref_rms']:.3e})")
        if args.json:
            json.dump(res, open(args.json, "w"), indent=1)
        return
    from wesep_amd.bin.export_engine import export_engine
    from wesep_amd.models import get_model
    os.makedirs(OUT, exist_ok=True)
    cfg, p = params()
    model = get_model("TFGridNet")(**KW, spk_fuse_type="multiply", use_spk_transform=False, joint_training=True,
                                   spk_model="ResNet18", spk_feat=True,
                                   spk_args=dict(feat_dim=80, embed_dim=256, pooling_func="TSTP", two_emb_layer=False))
    model.load_state_dict(p, strict=True)
    n, nf = export_engine(model, os.path.join(OUT, "m.wsw"))
    mix16, e1, e2 = B.signals()
    for name, x in (("mix", mix16), ("e1", e1), ("e2", e2)):
        B.write_wav(os.path.join(OUT, name + ".wav"), x)
    with open(os.path.join(OUT, "wav.scp"), "w") as f:
        f.write("utt1 runtime/_testdata/tfgridnet/mix.wav runtime/_testdata/tfgridnet/e1.wav runtime/_testdata/tfgridnet/e2.wav\n")
    est, emb, fb = expected(cfg, p, mix16, e1, e2)
    np.savez_compressed(os.path.join(OUT, "expected.npz"), est=est, emb=emb, fbank=fb)
    print("wrote", OUT, n, "tensors", nf * 4 >> 20, "MiB; est", est.shape, "rms", float(np.sqrt((est ** 2).mean())))


if __name__ == "__main__":
    main()
