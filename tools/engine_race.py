"""Diagnostic (GPU): do concurrent engines (separate_main --jobs J) reproduce the single-engine output bit for bit?
Runs the CLI on N copies of one utterance and reports, per run, which outputs differ from the sequential reference and
where (sample range, maximum difference in 16-bit steps).  Env passes through (WS_ENGINE_NO_CLUSTER, WS_ENGINE_POISON).
Usage: python tools/engine_race.py [repeats] [jobs] [utterances]"""
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
jobs = int(sys.argv[2]) if len(sys.argv) > 2 else 2
nutt = int(sys.argv[3]) if len(sys.argv) > 3 else 8
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


def run(tag, n, j):
    out = os.path.join(td, tag)
    os.makedirs(out, exist_ok=True)
    scp = os.path.join(td, tag + ".scp")
    open(scp, "w").write("".join(f"u{i} {td}/mix.wav {td}/e1.wav {td}/e2.wav\n" for i in range(n)))
    r = subprocess.run([exe, "--wav_scp", scp, "--model", os.path.join(td, "j.wsw"), "--output_dir", out, "--jobs", str(j)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    notes = [l for l in r.stdout.splitlines() if l.startswith("note:")]
    return [np.frombuffer(open(os.path.join(out, f"u{i}-spk1.wav"), "rb").read()[44:], dtype=np.int16).astype(np.int32)
            for i in range(n)], notes


ref = run("seq", 1, 1)[0][0]
print("env:", {k: v for k, v in os.environ.items() if k.startswith("WS_ENGINE")}, flush=True)
for rep in range(reps):
    outs, notes = run(f"par{rep}", nutt, jobs)
    bad = []
    for i, o in enumerate(outs):
        if not np.array_equal(o, ref):
            idx = np.nonzero(o != ref)[0]
            bad.append(f"u{i}: {len(idx)} samples in [{idx.min()}, {idx.max()}], max |d| {int(np.abs(o - ref).max())}")
    print(f"run {rep}: jobs {jobs}, {nutt} utterances: {len(bad)} differ; notes {len(notes)}; " + "; ".join(bad), flush=True)
