"""Diagnostic (GPU): as tools/engine_race.py, but the concurrency comes from P separate PROCESSES with one engine each
(own address spaces: no cross-engine memory interference possible) instead of worker threads in one process."""
import os
import subprocess
import sys
import tempfile
import wave

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wesep_amd.bin.export_engine import export_engine  # noqa: E402
from wesep_amd.models import get_model  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
procs = int(sys.argv[2]) if len(sys.argv) > 2 else 2
nutt = int(sys.argv[3]) if len(sys.argv) > 3 else 6
SPK_ARGS = dict(feat_dim=80, embed_dim=256, pooling_func="TSTP", two_emb_layer=False)
td = tempfile.mkdtemp()
torch.manual_seed(5)
model = get_model("BSRNN")(num_repeat=1, spk_fuse_type="multiply", multi_fuse=False, use_spk_transform=False,
                           joint_training=True, spk_model="ResNet18", spk_feat=True, spk_args=SPK_ARGS)
export_engine(model, os.path.join(td, "j.wsw"))
rng = np.random.default_rng(1)
for name, n in (("mix", 24000), ("e1", 32000), ("e2", 36000)):
    with wave.open(os.path.join(td, f"{name}.wav"), "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(16000)
        w.writeframes(rng.integers(-4000, 4000, n).astype(np.int16).tobytes())
exe = os.path.join(ROOT, "runtime", "separate_main")


def start(tag, n):
    out = os.path.join(td, tag)
    os.makedirs(out, exist_ok=True)
    scp = os.path.join(td, tag + ".scp")
    open(scp, "w").write("".join(f"u{i} {td}/mix.wav {td}/e1.wav {td}/e2.wav\n" for i in range(n)))
    return out, subprocess.Popen([exe, "--wav_scp", scp, "--model", os.path.join(td, "j.wsw"), "--output_dir", out],
                                 stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)


def read(out, n):
    return [np.frombuffer(open(os.path.join(out, f"u{i}-spk1.wav"), "rb").read()[44:], dtype=np.int16).astype(np.int32)
            for i in range(n)]


o, p = start("seq", 1)
p.wait()
ref = read(o, 1)[0]
for rep in range(reps):
    ps = [start(f"r{rep}p{k}", nutt) for k in range(procs)]
    bad = []
    for k, (o, p) in enumerate(ps):
        p.wait()
        assert p.returncode == 0, p.stderr.read()
        for i, x in enumerate(read(o, nutt)):
            if not np.array_equal(x, ref):
                idx = np.nonzero(x != ref)[0]
                bad.append(f"p{k}u{i}: {len(idx)} in [{idx.min()}, {idx.max()}] max {int(np.abs(x - ref).max())}")
    print(f"run {rep}: {procs} processes x {nutt} utterances: {len(bad)} differ; " + "; ".join(bad), flush=True)
