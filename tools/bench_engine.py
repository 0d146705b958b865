"""Real-time factor of the native runtime (runtime/libwesep_engine.so) next to the Python module tree on the same GPU:
the shipped pBSRNN recipe (6 repeats, multiply fusion, ResNet34 on fbank enrollment), one mixture with two enrollment
utterances per call -- the workload of the reference's runtime/bin/separate_main.cc.  Prints one JSON line.

    python tools/bench_engine.py [--seconds 4] [--iters 10]"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wesep_amd import engine as E  # noqa: E402
from wesep_amd.bin.export_engine import export_engine  # noqa: E402
from wesep_amd.models import get_model  # noqa: E402
from wesep_amd.utils.funcs import apply_cmvn, compute_fbank  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=4.0)
ap.add_argument("--enroll_seconds", type=float, default=4.0)
ap.add_argument("--iters", type=int, default=10)
args = ap.parse_args()
d = torch.device("cuda:0")
torch.manual_seed(0)
model = get_model("BSRNN")(num_repeat=6, spk_fuse_type="multiply", multi_fuse=False, use_spk_transform=False,
                           joint_training=True, spk_model="ResNet34", spk_feat=True,
                           spk_args=dict(feat_dim=80, embed_dim=256, pooling_func="TSTP", two_emb_layer=False))
path = os.path.join(tempfile.mkdtemp(), "bsrnn.wsw")
export_engine(model, path)
t0 = time.perf_counter()
eng = E.Engine(path)
load_ms = 1e3 * (time.perf_counter() - t0)
model = model.to(d).eval()
n, ne = int(16000 * args.seconds), int(16000 * args.enroll_seconds)
rng = np.random.default_rng(0)
mix = rng.integers(-3000, 3000, n).astype(np.int16)
e1, e2 = (rng.integers(-3000, 3000, ne).astype(np.int16) for _ in range(2))


def python_path():
    with torch.no_grad():
        m = torch.from_numpy(mix.astype(np.float32) / 32768).to(d).repeat(2, 1)
        en = torch.from_numpy(np.stack([e1, e2]).astype(np.float32) / 32768).to(d)
        est = model(m, apply_cmvn(compute_fbank(en, dither=0.0)))[0]
        return est.cpu().numpy()


def timed(fn):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(args.iters):
        out = fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t) / args.iters, out


ms_e, out_e = timed(lambda: eng.forward_pcm16(mix, e1, e2))
ms_p, out_p = timed(python_path)
rel = float(np.linalg.norm(out_e - out_p) / np.linalg.norm(out_p))
print(json.dumps({"workload": f"pBSRNN + ResNet34, 1 mixture x 2 enrollments, {args.seconds:g} s audio",
                  "engine_ms": round(ms_e, 2), "engine_rtf": round(ms_e / (1e3 * args.seconds), 5),
                  "python_ms": round(ms_p, 2), "python_rtf": round(ms_p / (1e3 * args.seconds), 5),
                  "engine_vs_python_rel": rel, "engine_load_ms": round(load_ms, 1),
                  "launches": eng.info("n_launches"), "arena_MiB": eng.info("arena_bytes") >> 20}))
