"""GPU bench of the Conv-TasNet / SpEx+ row (BASELINE.json configs[0]: batch 2 mixtures
= 4 rows of 4 s, fixed 256-d embeddings): fwd + multi-scale SI-SDR + bwd + per-tensor clip + Adam, one JSON
line with `roofline` (dominant kernel class, HIP events + algorithmic bytes).  `--cpu` times the oracle (CPU restatement
of the reference) on the same batch beside it (`cpu_baseline`).
Not the headline metric (bench.py keeps that); results are quoted in DESIGN.md."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SPEXPLUS = dict(N=256, L=20, B=256, H=512, P=3, X=8, R=4, spk_emb_dim=256, norm="gLN", use_spk_transform=False)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=4)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--cpu", action="store_true")
    args = ap.parse_args()
    from wesep_amd.functional import SISDRFn
    from wesep_amd.models import get_model
    from wesep_amd.optim import FusedClipAdam
    from wesep_amd.utils.synthetic import synth_batch
    d = torch.device("cuda:0")
    torch.manual_seed(0)
    model = get_model("ConvTasNet")(**SPEXPLUS, joint_training=False).to(d).train()
    opt = FusedClipAdam(model.parameters(), lr=1e-3, weight_decay=1e-4, clip_grad=5.0)
    T = 64000
    wav, tgt, emb = (t.to(d) for t in synth_batch(args.rows, T, 42))

    def step():
        ests = model(wav, emb)
        loss = sum(w * SISDRFn.apply(e, tgt, 1e-8) for w, e in zip((0.8, 0.1, 0.1), ests))
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss

    from wesep_amd import dev, _lib as L
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_common as BC
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    dev.prof_enable(True)                     # HIP events around the library's four timed kernel classes
    dev.alg_reset(True)                       # algorithmic bytes / flops per launch
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    dev.prof_enable(False)
    roof = BC.roofline(dev, L, args.steps)
    dev.alg_reset(False)
    out = {"metric": "utterances/sec (4 s, 16 kHz) fwd+bwd, Conv-TasNet SpEx+ (fixed embeddings)",
           "value": args.rows * args.steps / el, "unit": "utterances/s", "ms_per_step": el / args.steps * 1e3,
           "rows": args.rows, "steps": args.steps, "dtype": "bf16x3", "data": "synthetic",
           "final_loss_dB": float(loss.item()),
           "params_M": sum(p.numel() for p in model.parameters()) / 1e6,
           "peak_mem_GB": torch.cuda.max_memory_allocated() / 1e9, "roofline": roof, "cpu_baseline": None,
           "n_gpus": 1, "higher_is_better": True}
    if args.cpu:
        from oracle import convtasnet_oracle as CT
        torch.set_num_threads(16)
        cfg = CT.ConvTasNetConfig(**{k: v for k, v in SPEXPLUS.items()})
        p = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.state_dict().items()}
        w, t, e = wav.cpu(), tgt.cpu(), emb.cpu()
        ts = []
        for i in range(2):
            t0 = time.perf_counter()
            l = CT.multiscale_sisdr_loss(CT.convtasnet_forward(p, cfg, w, e), t)
            l.backward()
            ts.append(time.perf_counter() - t0)
        out["cpu_baseline"] = {"value": args.rows / ts[-1], "unit": "utterances/s", "cores": 16, "kind": "port",
                               "sample": f"oracle fwd + loss + bwd on the same {args.rows} rows, second of 2 runs"}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
