"""Diagnosis: the speaker-fusion gradient of the TF-GridNet emb_ks = 4 fixture case on the device vs the oracle, per
frequency bin (tests/test_tfgridnet_gpu.py found 1.5e-2 on spk_fuse.fc.linear.* in that case only)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import bsrnn_oracle as O, tfgridnet_oracle as TG  # noqa: E402
from oracle.make_golden import TFGRIDNET_CASES, tfgridnet_batch  # noqa: E402
from wesep_amd.models import get_model  # noqa: E402
from wesep_amd.utils.losses import parse_loss  # noqa: E402

d = torch.device("cuda:0")
for name in sys.argv[1:] or ["tfgridnet_ks4_r2_t1600"]:
    kw, R, T, seed = TFGRIDNET_CASES[name]
    cfg = TG.TFGridNetConfig(**kw)
    params = TG.synth_params(cfg, seed)
    wav, tgt, emb = tfgridnet_batch(cfg, R, T, seed)
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    out = TG.tfgridnet_forward(p, cfg, wav, emb)
    ref = out[0] if isinstance(out, (tuple, list)) else out
    O.sisdr_loss(ref.reshape(-1, T), tgt.reshape(-1, T)).backward()
    for env in ({}, {"WESEP_GEMM": "f32", "WESEP_LSTM": "f32"}):
        os.environ.pop("WESEP_GEMM", None), os.environ.pop("WESEP_LSTM", None)
        os.environ.update(env)
        model = get_model("TFGridNet")(**kw, joint_training=False)
        model.load_state_dict(params, strict=True)
        model = model.to(d).train()
        est, _ = model(wav.to(d), emb.to(d))
        parse_loss("SISDR")[0](est.reshape(-1, T), tgt.to(d).reshape(-1, T)).backward()
        torch.cuda.synchronize()
        print(name, env or "bf16x3")
        rels = {k: float((prm.grad.cpu().double() - p[k].grad.double()).norm() / (p[k].grad.double().norm() + 1e-30))
                for k, prm in model.named_parameters()}
        for k in sorted(rels, key=lambda k: -rels[k])[:8]:
            print(f"   {k:40s} rel {rels[k]:.2e}  |oracle| {float(p[k].grad.norm()):.3e}")
        b = dict(model.named_parameters())["spk_fuse.fc.linear.bias"].grad.cpu().double()
        bo = p["spk_fuse.fc.linear.bias"].grad.double()
        print("   bias grad per bin: (got - want) / max|want|:",
              " ".join(f"{float(x):+.1e}" for x in ((b - bo) / bo.abs().max())[:70]))
