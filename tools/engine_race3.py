"""Diagnostic (GPU): which part of the engine's launch plan is not reproducible when two engines share the GPU?
Two Python threads (ctypes releases the GIL) each own an Engine and call separate() in a loop; variants isolate the
separator (fixed embeddings), the ResNet encoder (fbank enrollment) and the device kaldi fbank (waveform enrollment).
Usage: python tools/engine_race3.py [iters]"""
import os
import sys
import tempfile
import threading

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wesep_amd import engine as E  # noqa: E402
from wesep_amd.bin.export_engine import export_engine  # noqa: E402
from wesep_amd.models import get_model  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
td = tempfile.mkdtemp()
SPK_ARGS = dict(feat_dim=80, embed_dim=256, pooling_func="TSTP", two_emb_layer=False)
rng = np.random.default_rng(1)
mix = (rng.integers(-4000, 4000, (2, 24000)) / 32768).astype(np.float32)
wave_en = (rng.integers(-4000, 4000, (2, 32000)) / 32768).astype(np.float32)
fbank = rng.standard_normal((2, 198, 80)).astype(np.float32)
fbank -= fbank.mean(1, keepdims=True)
emb = rng.standard_normal((2, 256)).astype(np.float32)


def build(joint, fuse="multiply", rep=1):
    torch.manual_seed(5)
    kw = dict(num_repeat=rep, spk_fuse_type=fuse, multi_fuse=False, use_spk_transform=False)
    if joint:
        m = get_model("BSRNN")(joint_training=True, spk_model="ResNet18", spk_feat=True, spk_args=SPK_ARGS, **kw)
    else:
        m = get_model("BSRNN")(joint_training=False, **kw)
    path = os.path.join(td, f"m{int(joint)}_{fuse}_{rep}.wsw")
    export_engine(m, path)
    return path


def race(tag, path, enroll, kind, nthreads=2, T=None):
    mx = mix if T is None else mix[:, :T]
    ref_eng = E.Engine(path)
    ref = ref_eng.separate(mx, enroll, kind)
    again = ref_eng.separate(mx, enroll, kind)
    ref_eng.close()
    assert np.array_equal(ref, again), "not reproducible even alone"
    bad, worst, frames = [0] * nthreads, [0.0] * nthreads, [set() for _ in range(nthreads)]

    def work(k):
        eng = E.Engine(path)
        for _ in range(iters):
            out = eng.separate(mx, enroll, kind)
            if not np.array_equal(out, ref):
                bad[k] += 1
                dif = np.abs(out - ref)
                worst[k] = max(worst[k], float(dif.max() / np.abs(ref).max()))
                idx = np.nonzero(dif.max(0) > 1e-4 * np.abs(ref).max())[0]
                if len(idx):
                    frames[k].add((int(idx.min()) // 128, int(idx.max()) // 128))
        eng.close()

    ts = [threading.Thread(target=work, args=(k,)) for k in range(nthreads)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    print(f"{tag}: {nthreads} engines x {iters}: mismatches {bad}, worst {max(worst):.1e} of peak, "
          f"frame ranges {sorted(set().union(*frames))[:6]}", flush=True)


print("env:", {k: v for k, v in os.environ.items() if k.startswith("WS_ENGINE")}, flush=True)
p_fixed, p_joint = build(False), build(True)
race("separator only (fixed embeddings, 1 repeat)", p_fixed, emb, E.ENROLL_EMBEDDING)
race("separator only, 1 engine thread (control)", p_fixed, emb, E.ENROLL_EMBEDDING, nthreads=1)
race("ResNet18 + separator (fbank enrollment)", p_joint, fbank, E.ENROLL_FBANK)
race("kaldi fbank + ResNet18 + separator (waveform enrollment)", p_joint, wave_en, E.ENROLL_WAVE)
race("separator only, 4 engines", p_fixed, emb, E.ENROLL_EMBEDDING, nthreads=4)
