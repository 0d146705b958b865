"""TEST INFRASTRUCTURE.  Golden TRAINING TRAJECTORY of pBSRNN from the CPU oracle (oracle/bsrnn_oracle.py, itself pinned
to the imported reference by tests/test_oracle_golden.py): the reference's step semantics -- forward, SI-SDR, backward,
per-tensor clip (wesep/utils/funcs.py:79-88), Adam with coupled L2 (wesep/bin/train.py:237-238), ExponentialDecrease
learning rate set before every step (wesep/utils/executor.py:80-81) -- repeated for 60 steps in fp32 on the CPU.

It is the stand-in available here for BASELINE.json's "SI-SNRi within 0.1 dB of reference": the product's split-bf16
step has to TRACK this fp32 loss curve step by step and arrive at the same parameters
(tests/test_bsrnn_gpu.py::test_training_trajectory_tracks_the_fp32_oracle).  Only tests read the fixture.

    python -m oracle.make_trajectory            # ~3 minutes on 8 cores -> tests/golden/bsrnn_trajectory_r4_t16000_s60.npz

Fixture: losses [60]; lr [60]; per parameter tensor 256 sampled elements (indices from a generator seeded with the
tensor's name) of the INITIAL and the FINAL value, and the L2 norm of the whole update.
"""
import os
import sys
import zlib

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import bsrnn_oracle as O  # noqa: E402

NAME = "bsrnn_trajectory_r4_t16000_s60"
KW = dict(num_repeat=2, spk_fuse_type="multiply", multi_fuse=False)
SEED, R, T, STEPS, NBATCH = 21, 4, 16000, 60, 6
LR0, LR1, WD, CLIP = 1e-3, 2.5e-5, 1e-4, 5.0
NSAMPLE = 256


def batches():
    """NBATCH distinct synthetic batches, cycled (so the model has something to fit)."""
    return [O.synth_batch(R, T, 200 + i) for i in range(NBATCH)]


def sample_index(name, numel):
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    return np.sort(rng.choice(numel, size=min(NSAMPLE, numel), replace=False))


def run(log=print):
    cfg = O.BSRNNConfig(**KW)
    params = O.synth_params(cfg, SEED)
    p = {k: v.clone() for k, v in params.items()}
    m = {k: torch.zeros_like(v) for k, v in p.items()}
    v = {k: torch.zeros_like(val) for k, val in p.items()}
    bs = batches()
    losses, lrs = [], []
    for step in range(1, STEPS + 1):
        wav, tgt, emb = bs[(step - 1) % NBATCH]
        lr = O.exponential_decrease_lr(step - 1, STEPS, LR0, LR1)
        q = {k: t.clone().requires_grad_(True) for k, t in p.items()}
        loss = O.sisdr_loss(O.bsrnn_forward(q, cfg, wav, emb), tgt)
        loss.backward()
        grads = {k: t.grad for k, t in q.items()}
        O.clip_gradients_(grads, CLIP)
        for k in p:
            O.adam_l2_step_(p[k], grads[k], m[k], v[k], step, lr, weight_decay=WD)
        losses.append(float(loss))
        lrs.append(lr)
        log(f"step {step:3d}  lr {lr:.3e}  loss {losses[-1]:+.4f} dB")
    out = {"losses": np.asarray(losses, np.float64), "lr": np.asarray(lrs, np.float64)}
    for k in p:
        idx = sample_index(k, p[k].numel())
        out["idx/" + k] = idx.astype(np.int64)
        out["init/" + k] = params[k].reshape(-1)[idx].numpy()
        out["final/" + k] = p[k].reshape(-1)[idx].numpy()
        out["updnorm/" + k] = np.float64((p[k].double() - params[k].double()).norm())
    return out


if __name__ == "__main__":
    torch.set_num_threads(os.cpu_count())
    res = run()
    path = os.path.join(ROOT, "tests", "golden", NAME + ".npz")
    np.savez_compressed(path, **res)
    print("wrote", path, os.path.getsize(path), "bytes")
