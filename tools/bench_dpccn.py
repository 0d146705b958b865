"""GPU bench of the DPCCN row (default reference configuration, fixed embeddings):
fwd + SI-SDR + bwd + per-tensor clip + Adam on `--rows` rows of 4 s, one JSON line.  Not the headline metric."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=4)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--joint", action="store_true",
                    help="BASELINE configs[2]: jointly trained wespeaker ResNet34 on [R, 398, 80] fbank enrollment "
                         "(examples/librimix/tse/v2/confs/dpccn.yaml) instead of fixed embeddings")
    ap.add_argument("--cpu", action="store_true", help="also time the CPU oracle on a bounded sample (cpu_baseline)")
    args = ap.parse_args()
    from wesep_amd.functional import SISDRFn
    from wesep_amd.models import get_model
    from wesep_amd.optim import FusedClipAdam
    from wesep_amd.utils.synthetic import synth_batch
    d = torch.device("cuda:0")
    torch.manual_seed(0)
    if args.joint:
        model = get_model("DPCCN")(joint_training=True, spk_model="ResNet34", spk_feat=True,
                                   spk_args=dict(feat_dim=80, embed_dim=256, pooling_func="TSTP", two_emb_layer=False))
    else:
        model = get_model("DPCCN")(joint_training=False)
    model = model.to(d).train()
    opt = FusedClipAdam(model.parameters(), lr=1e-3, weight_decay=1e-4, clip_grad=5.0)
    wav, tgt, emb = (t.to(d) for t in synth_batch(args.rows, 64000, 42))
    if args.joint:
        fb = torch.randn(args.rows, 398, 80, generator=torch.Generator().manual_seed(43))
        emb = (fb - fb.mean(1, keepdim=True)).to(d)

    def step():
        est, _ = model(wav, emb)
        loss = SISDRFn.apply(est, tgt, 1e-8)
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss

    from wesep_amd import dev, _lib as L
    import bench_common as BC
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    dev.prof_enable(True)
    dev.alg_reset(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    dev.prof_enable(False)
    roof = BC.roofline(dev, L, args.steps)
    dev.alg_reset(False)
    cpu = BC.cpu_baseline("dpccn") if args.cpu else None
    print(json.dumps({"metric": "utterances/sec (4 s, 16 kHz) fwd+bwd, DPCCN (" +
                                ("joint ResNet34 speaker encoder" if args.joint else "fixed embeddings") + ")",
                      "value": args.rows * args.steps / el, "unit": "utterances/s",
                      "ms_per_step": el / args.steps * 1e3, "rows": args.rows, "steps": args.steps, "dtype": "bf16x3",
                      "data": "synthetic", "final_loss_dB": float(loss.item()),
                      "params_M": sum(p.numel() for p in model.parameters()) / 1e6,
                      "peak_mem_GB": torch.cuda.max_memory_allocated() / 1e9, "roofline": roof, "cpu_baseline": cpu,
                      "n_gpus": 1, "higher_is_better": True}), flush=True)


if __name__ == "__main__":
    main()
