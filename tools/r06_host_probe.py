"""Round 6 probe: how far ahead of the GPU is the host in the headline step?  Host wall-clock of the enqueue phases (forward, loss,
zero_grad, backward, optimizer) of bench.py's step, without any synchronisation inside the step, next to the GPU's step time.

    python tools/r06_host_probe.py [--steps 10]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--rows", type=int, default=32)
    a = ap.parse_args()
    import bench as B
    from oracle.bsrnn_oracle import synth_batch
    from wesep_amd.models import get_model
    from wesep_amd.optim import FusedClipAdam
    from wesep_amd.utils.losses import parse_loss
    d = torch.device("cuda:0")
    torch.manual_seed(0)
    model = get_model("BSRNN")(**B.MODEL_KW) if hasattr(B, "MODEL_KW") else None
    if model is None:
        model = get_model("BSRNN")(spk_emb_dim=256, sr=16000, win=512, stride=128, feature_dim=128, num_repeat=6,
                                   use_spk_transform=False, use_bidirectional=True, spk_fuse_type="FiLM", multi_fuse=True,
                                   joint_training=False)
    model = model.to(d).train()
    opt = FusedClipAdam(model.parameters(), lr=1e-3, weight_decay=1e-4, clip_grad=5.0)
    crit = parse_loss("SISDR")[0]
    wav, tgt, emb = (t.to(d) for t in synth_batch(a.rows, 64000, 42))
    # host time inside the forward, by part: every child module of the separator + the two big autograd functions
    parts = {}

    def timed(label, fn):
        def w(*ar, **kw):
            t0 = time.perf_counter()
            out = fn(*ar, **kw)
            parts[label] = parts.get(label, 0.0) + time.perf_counter() - t0
            return out
        return w
    from wesep_amd import functional as F_
    for i, layer in enumerate(model.separator.separation):
        layer.forward = timed(f"separation[{i}] {type(layer).__name__}", layer.forward)
    model._speaker = timed("_speaker", model._speaker)
    model.separator.make_carriers = timed("make_carriers", model.separator.make_carriers)
    _bs, _md = F_.BandSplitFn.apply, F_.MaskDecodeFn.apply
    F_.BandSplitFn.apply = timed("BandSplitFn", _bs)
    F_.MaskDecodeFn.apply = timed("MaskDecodeFn", _md)
    # ... and by device entry point (every public function of wesep_amd.dev that launches something)
    from wesep_amd import dev as _dev
    import types
    for nm in dir(_dev):
        fn = getattr(_dev, nm)
        if isinstance(fn, types.FunctionType) and not nm.startswith("_") and fn.__module__ == _dev.__name__:
            setattr(_dev, nm, timed("dev." + nm, fn))
    names = ["forward", "loss", "zero_grad", "backward", "opt.step"]
    acc = [0.0] * 5

    def step(rec):
        t = [time.perf_counter()]
        est, _ = model(wav, emb)
        t.append(time.perf_counter())
        loss = crit(est, tgt).mean()
        t.append(time.perf_counter())
        opt.zero_grad()
        t.append(time.perf_counter())
        loss.backward()
        t.append(time.perf_counter())
        opt.step()
        t.append(time.perf_counter())
        if rec:
            for i in range(5):
                acc[i] += t[i + 1] - t[i]

    for _ in range(3):
        step(False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step(True)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"{a.steps} steps: host enqueue {t_host / a.steps * 1e3:.2f} ms per step, with the final synchronize {t_all / a.steps * 1e3:.2f} ms per step")
    for n, v in zip(names, acc):
        print(f"   host time in {n:10s} {v / a.steps * 1e3:7.2f} ms per step")
    for k, v in sorted(parts.items(), key=lambda kv: -kv[1]):
        if v / (a.steps + 3) * 1e3 >= 0.05:
            print(f"      host time in {k:36s} {v / (a.steps + 3) * 1e3:7.2f} ms per step (forward + backward)")


if __name__ == "__main__":
    main()
