"""Numerics probe for the 2-byte storage of the saved recurrence state (round 4): what do fp16 saved gates, bf16 d(gates),
fp16 cell state and bf16 d(hcat) cost in gradient / trajectory accuracy?  Runs the PRODUCT's host composition
(models.BSRNN -> functional.ResRNNBlkFn) and rounds the fp32 buffers in place between the kernels (WESEP_H2_PROBE bits,
wesep_amd/functional.py) -- on the CPU emulation of the entry points (tests/emu_*.py; no GPU needed: `--cpu`) or on the
MI355X through the real kernels.  Compares per-tensor gradients and a short Adam trajectory against the fp32 oracle.

    python tools/r04_h2_numerics.py --cpu [--steps 20]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cpu", action="store_true")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--masks", default="0,1,2,3,7,23")
    ap.add_argument("--case", default="bsrnn_film_multi_r2_t3000")
    ap.add_argument("--T", type=int, default=0, help="override the case's length in samples (64000 = the headline's 501 frames)")
    ap.add_argument("--repeat", type=int, default=0, help="override num_repeat (6 = the recipe)")
    ap.add_argument("--full", action="store_true", help="trajectory at the fixture's own size (R = 4 x 1 s) also on the CPU")
    a = ap.parse_args()
    from oracle import bsrnn_oracle as O
    from oracle import make_trajectory as MT
    from oracle.make_golden import CASES
    if a.cpu:
        from _pytest.monkeypatch import MonkeyPatch
        from tests import emu_blk, emu_bsrnn, emu_dev
        mp = MonkeyPatch()
        emu_dev.install(mp)
        emu_blk.install(mp)
        emu_bsrnn.install(mp, real_resrnn=True)
        mp.setattr(torch.Tensor, "is_cuda", property(lambda self: True))
        os.environ["WESEP_WGRAD_OVERLAP"] = "0"
        d = torch.device("cpu")
    else:
        d = torch.device("cuda:0")
    from wesep_amd.models import get_model

    def build(kw, seed):
        cfg = O.BSRNNConfig(**kw)
        params = O.synth_params(cfg, seed)
        model = get_model("BSRNN")(spk_emb_dim=cfg.spk_emb_dim, sr=cfg.sr, win=cfg.win, stride=cfg.stride,
                                   feature_dim=cfg.feature_dim, num_repeat=cfg.num_repeat,
                                   use_spk_transform=cfg.use_spk_transform, spk_fuse_type=cfg.spk_fuse_type,
                                   multi_fuse=cfg.multi_fuse, joint_training=False)
        model.load_state_dict(params, strict=True)
        return cfg, params, (model if a.cpu else model.to(d))

    def loss_of(model, wav, tgt, emb):
        from wesep_amd import functional as f0
        est, _ = model(wav.to(d), emb.to(d))
        return est, f0.SISDRFn.apply(est, tgt.to(d), 1e-8)

    # ---- (1) per-tensor gradients on a fixture-sized case -------------------------------------------------------------
    kw, R, T, seed = CASES[a.case]
    T = a.T or T
    if a.repeat:
        kw = dict(kw, num_repeat=a.repeat)
    cfg, params, _ = build(kw, seed)
    wav, tgt, emb = O.synth_batch(R, T, seed)
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    loss_o = O.sisdr_loss(O.bsrnn_forward(p, cfg, wav, emb), tgt)
    loss_o.backward()
    g_o = {k: v.grad for k, v in p.items()}
    for mask in [int(x) for x in a.masks.split(",")]:
        os.environ["WESEP_H2_PROBE"] = str(mask)
        cfg, params, model = build(kw, seed)
        model.train()
        est, loss = loss_of(model, wav, tgt, emb)
        loss.backward()
        per = {k: rel(prm.grad, g_o[k]) for k, prm in model.named_parameters()}
        wk = max(per, key=per.get)
        vals = sorted(per.values())
        print(f"[grad] probe={mask:2d}: dloss {abs(loss.item() - loss_o.item()):.2e} dB  worst {per[wk]:.2e} ({wk})  "
              f"median {vals[len(vals) // 2]:.2e}  p90 {vals[int(len(vals) * .9)]:.2e}", flush=True)

    # ---- (2) a short training trajectory (oracle's step semantics, torch-side clip + Adam on both) --------------------
    if a.steps <= 0:
        return
    bs = MT.batches()
    small = a.cpu and not a.full
    Rr = 2 if small else MT.R
    bs = [(w[:Rr, :8000 if small else MT.T], t[:Rr, :8000 if small else MT.T], e[:Rr]) for w, t, e in bs]
    cfgT = O.BSRNNConfig(**MT.KW)
    init = O.synth_params(cfgT, MT.SEED)

    def run(kind, mask):
        os.environ["WESEP_H2_PROBE"] = str(mask)
        pp = {k: v.clone() for k, v in init.items()}
        m = {k: torch.zeros_like(v) for k, v in pp.items()}
        v = {k: torch.zeros_like(val) for k, val in pp.items()}
        model = None
        if kind == "product":
            _, _, model = build(MT.KW, MT.SEED)
            model.train()
        losses = []
        for step in range(1, a.steps + 1):
            wav, tgt, emb = bs[(step - 1) % len(bs)]
            lr = O.exponential_decrease_lr(step - 1, MT.STEPS, MT.LR0, MT.LR1)
            if kind == "oracle":
                q = {k: t.clone().requires_grad_(True) for k, t in pp.items()}
                loss = O.sisdr_loss(O.bsrnn_forward(q, cfgT, wav, emb), tgt)
                loss.backward()
                grads = {k: t.grad for k, t in q.items()}
            else:
                model.load_state_dict({k: t.to(d) for k, t in pp.items()}, strict=True)
                for prm in model.parameters():
                    prm.grad = None
                from wesep_amd import dev
                dev.bump_weight_epoch()
                _, loss = loss_of(model, wav, tgt, emb)
                loss.backward()
                grads = {k: prm.grad.detach().cpu().clone() for k, prm in model.named_parameters()}
            losses.append(float(loss))
            O.clip_gradients_(grads, MT.CLIP)
            for k in pp:
                O.adam_l2_step_(pp[k], grads[k], m[k], v[k], step, lr, weight_decay=MT.WD)
        return np.asarray(losses), pp

    t0 = time.time()
    lo, po = run("oracle", 0)
    print(f"[traj] oracle {a.steps} steps in {time.time() - t0:.0f} s; loss {lo[0]:+.4f} -> {lo[-1]:+.4f} dB", flush=True)
    for mask in [int(x) for x in a.masks.split(",")]:
        lp, pp = run("product", mask)
        num = sum(float(((pp[k] - po[k]).double() ** 2).sum()) for k in po)
        den = sum(float(((po[k] - init[k]).double() ** 2).sum()) for k in po)
        from wesep_amd import functional as f0
        print(f"[traj] probe={mask:2d}: max |dloss| {np.abs(lp - lo).max():.2e} dB (last {abs(lp[-1] - lo[-1]):.2e}); "
              f"accumulated update rel-L2 {np.sqrt(num / den):.2e}; fp16 saturations {f0._PROBE_SAT[0]}, "
              f"largest scaled |d(gates)| / 65504 = {f0._PROBE_SAT[1]:.3e}", flush=True)


if __name__ == "__main__":
    main()
