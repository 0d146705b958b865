"""Diagnostic (GPU): which stage of the pBSRNN forward is not reproducible when two instances overlap on the GPU?
Every stage is run alone (reference), then N times alternating between two HIP streams without synchronising in
between, so instances of the same kernels overlap; every output is compared bit for bit with the reference.
Usage: python tools/stream_race.py [N]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import bsrnn_oracle as O  # noqa: E402
from wesep_amd import functional as F_  # noqa: E402
from wesep_amd.models import get_model  # noqa: E402

d = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
R, T = 2, 24000
kw = dict(num_repeat=1, spk_fuse_type="multiply", multi_fuse=False)
cfg = O.BSRNNConfig(**kw)
model = get_model("BSRNN")(use_spk_transform=False, joint_training=False, **kw)
model.load_state_dict(O.synth_params(cfg, 1))
model.to(d).eval()
wav, tgt, emb = (t.to(d) for t in O.synth_batch(R, T, 1))
plan = model._plan(d)
with torch.no_grad():
    z0, xbs = F_.BandSplitFn.apply(wav, plan, *model._bn_params())
    bsnet = [m for m in model.separator.separation if hasattr(m, "band_rnn")][0]
    fuse = [m for m in model.separator.separation if not hasattr(m, "band_rnn")][0]
    z1 = bsnet.band_rnn(z0, "time")
stages = {
    "stft + band split + BN": lambda: F_.BandSplitFn.apply(wav, plan, *model._bn_params())[0],
    "speaker fusion": lambda: fuse(z0, emb),
    "ResRNN time view": lambda: bsnet.band_rnn(z0, "time"),
    "ResRNN band view": lambda: bsnet.band_comm(z1, "band"),
    "mask MLP + iSTFT": lambda: F_.MaskDecodeFn.apply(z1, xbs, plan, T, *model._mask_params()),
    "whole forward": lambda: model(wav, emb)[0],
}
streams = [torch.cuda.Stream(device=d) for _ in range(2)]
for name, fn in stages.items():
    with torch.no_grad():
        ref = fn()
        torch.cuda.synchronize()
        again = fn()
        torch.cuda.synchronize()
        assert torch.equal(ref, again), name + ": not reproducible even alone"
        outs = []
        for s in streams:
            s.wait_stream(torch.cuda.current_stream())
        for i in range(N):
            with torch.cuda.stream(streams[i % 2]):
                outs.append(fn())
        torch.cuda.synchronize()
    bad = [float((o - ref).abs().max() / ref.abs().max()) for o in outs if not torch.equal(o, ref)]
    print(f"{name}: {len(bad)} of {N} overlapping instances differ" + (f", worst {max(bad):.1e} of peak" if bad else ""),
          flush=True)

# ---- pairs: stage A keeps stream 0 busy while stage B runs on stream 1 (and vice versa) -----------------------------
names = [n for n in stages if n != "whole forward"]
refs = {}
with torch.no_grad():
    for n in names:
        refs[n] = stages[n]()
    torch.cuda.synchronize()
    for a in names:
        row = []
        for b in names:
            outs = []
            for i in range(N // 2):
                with torch.cuda.stream(streams[0]):
                    stages[a]()
                with torch.cuda.stream(streams[1]):
                    outs.append(stages[b]())
            torch.cuda.synchronize()
            row.append(sum(0 if torch.equal(o, refs[b]) else 1 for o in outs))
        print(f"beside [{a}]: mismatching instances of " + ", ".join(f"[{b}] {k}" for b, k in zip(names, row)), flush=True)
