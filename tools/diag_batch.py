"""Diagnostic (GPU): is a row of a big batch bit-identical to the same row run in a small batch?
Usage: WESEP_GEMM=f32|bf16x3 WESEP_LSTM=f32|bf16x3 python tools/diag_batch.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import bsrnn_oracle as O  # noqa: E402
from wesep_amd.models import get_model  # noqa: E402

d = torch.device("cuda:0")
kw = dict(num_repeat=2, spk_fuse_type="FiLM", multi_fuse=True)
cfg = O.BSRNNConfig(**kw)
params = O.synth_params(cfg, 6)
model = get_model("BSRNN")(use_spk_transform=False, joint_training=False, **kw)
model.load_state_dict(params)
model.to(d).eval()
wav, tgt, emb = O.synth_batch(32, 64000, 6)
for gm, lm in (("f32", "f32"), ("bf16x3", "f32"), ("f32", "bf16x3"), ("bf16x3", "bf16x3")):
    os.environ["WESEP_GEMM"], os.environ["WESEP_LSTM"] = gm, lm
    with torch.no_grad():
        big, _ = model(wav.to(d), emb.to(d))
        big2, _ = model(wav.to(d), emb.to(d))
        print(gm, lm, "rerun big max diff:", float((big - big2).abs().max()), flush=True)
        for r0 in (0, 30):
            small, _ = model(wav[r0:r0 + 2].to(d), emb[r0:r0 + 2].to(d))
            df = (big[r0:r0 + 2] - small)
            print(gm, lm, r0, "rel", float(df.norm() / small.norm()), "max", float(df.abs().max()), flush=True)
