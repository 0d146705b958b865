"""GPU bench of the two recipe variants around the headline step, through `Executor.train`:

  ssa     joint pBSRNN (ResNet34 on fbank enrollment) with SSA_enroll_prob = 1: a no-grad pass on the given enrollment,
          kaldi fbank + CMN of its estimate on the device, then the real step on that self-enrollment (executor.py:89-102)
  multi   BSRNN_Multi (bsrnn_multi_optim.yaml): two separator passes over one band split, SI-SDR on both outputs

R rows x 4 s, fwd + loss + bwd + clip + Adam per step, one JSON line each.  Not the headline metric."""
import argparse
import json
import os
import random
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SPK = dict(feat_dim=80, embed_dim=256, pooling_func="TSTP", two_emb_layer=False)
KW = dict(sr=16000, win=512, stride=128, feature_dim=128, num_repeat=6, spk_fuse_type="FiLM", use_spk_transform=False,
          multi_fuse=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--what", default="ssa,multi,joint")
    ap.add_argument("--rows", type=int, default=32)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=2)
    a = ap.parse_args()
    from wesep_amd.models import get_model
    from wesep_amd.optim import FusedClipAdam
    from wesep_amd.utils.executor import Executor
    from wesep_amd.utils.losses import parse_loss
    from wesep_amd.utils.schedulers import ExponentialDecrease
    from wesep_amd.utils.synthetic import synth_batch
    d = torch.device("cuda:0")
    R, T = a.rows, 64000
    wav, tgt, _ = synth_batch(R, T, 42)
    fb = torch.randn(R, 398, 80, generator=torch.Generator().manual_seed(43))
    fb = fb - fb.mean(1, keepdim=True)
    for what in a.what.split(","):
        torch.manual_seed(0)
        random.seed(0)
        if what == "multi":
            model = get_model("BSRNN_Multi")(joint_training=True, spk_model="ResNet34", spk_args=SPK, spk_emb_dim=256,
                                             spk_feat=False, feat_type="consistent", **{**KW, "spk_fuse_type": "multiply",
                                                                                       "multi_fuse": False})
            enroll = 0.1 * torch.randn(R, 48000, generator=torch.Generator().manual_seed(44))
            kw = dict(se_loss_weight=([[0, 1]], [[0.4, 0.6]]), speaker_feat=False)
        else:
            model = get_model("BSRNN")(joint_training=True, spk_model="ResNet34", spk_args=SPK, spk_emb_dim=256,
                                       spk_feat=True, **KW)
            enroll = fb
            kw = dict(se_loss_weight=([[0]], [[1.0]]), speaker_feat=True, SSA_enroll_prob=1.0 if what == "ssa" else 0.0,
                      fbank_args=dict(num_mel_bins=80, frame_length=25, frame_shift=10, dither=0.0), sample_rate=16000)
        model = model.to(d).train()
        opt = FusedClipAdam(model.parameters(), lr=1e-3, weight_decay=1e-4, clip_grad=5.0)
        sched = ExponentialDecrease(opt, num_epochs=150, epoch_iter=1000, initial_lr=1e-3, final_lr=2.5e-5, warm_up_epoch=0)
        batch = {"wav_mix": wav, "wav_targets": tgt, "spk_embeds": enroll, "spk_label": torch.zeros(0)}   # host rows: the loop stages them (pinned, one batch ahead) like a DataLoader's
        ex = Executor()
        crit = parse_loss("SISDR")

        def run(n):
            return ex.train([batch] * n, [model], n, [opt], crit, [sched], scaler=None, epoch=1, enable_amp=False, logger=None,
                            clip_grad=5.0, log_batch_interval=10 ** 9, device=d, **kw)[0]
        run(a.warmup)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loss = run(a.steps)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / a.steps * 1e3
        print(json.dumps({"metric": f"utterances/sec (4 s, 16 kHz) fwd+bwd, pBSRNN {what}", "variant": what,
                          "value": R / ms * 1e3, "unit": "utterances/s", "ms_per_step": ms, "rows": R, "steps": a.steps,
                          "dtype": "bf16x3+fp16x2+fp16x1 split products, fp32 accumulate (bench.py)", "data": "synthetic", "mean_loss_dB": float(loss),
                          "peak_mem_GB": torch.cuda.max_memory_allocated() / 1e9}), flush=True)
        del model, opt, ex
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()


if __name__ == "__main__":
    main()
