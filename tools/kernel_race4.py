"""Diagnostic (GPU): which launches of a pBSRNN training step disturb a vendor FFT (torch.fft.rfft -> rocFFT) running on
another stream?  Records every dev.* launch of one forward + backward, replays each beside the canary and reports the
launches with mismatches.  (Round 2: gemm_b2p does; the canary computes from its own buffers only.)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import bsrnn_oracle as O  # noqa: E402
from wesep_amd import dev  # noqa: E402
from wesep_amd import functional as F_  # noqa: E402
from wesep_amd.functional import SISDRFn  # noqa: E402
from wesep_amd.models import get_model  # noqa: E402

d = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
R, T = 2, 24000
os.environ["WESEP_WGRAD_OVERLAP"] = "0"
kw = dict(num_repeat=1, spk_fuse_type="FiLM", multi_fuse=True)
cfg = O.BSRNNConfig(**kw)
model = get_model("BSRNN")(use_spk_transform=False, joint_training=False, **kw)
model.load_state_dict(O.synth_params(cfg, 1))
model.to(d).train()
wav, tgt, emb = (t.to(d) for t in O.synth_batch(R, T, 1))

NAMES = [n for n in dir(dev) if callable(getattr(dev, n)) and not n.startswith("_") and n[0].islower() and n not in (
    "gemm_mode", "flat", "tn_splits", "tnb_splits", "lstm_mode", "lstm_blk_mode", "cu_count", "lstm_cluster_ok",
    "lstm_fuse_ok", "bl_num_blocks", "bl_positions", "to_blocked", "from_blocked", "prof_enable", "prof_collect",
    "conv_out", "row_splits", "weight_epoch", "bump_weight_epoch", "poll_cluster_status", "gn_bwd_fused_ok",
    "total_sum")]
calls, keep = [], []
real = {n: getattr(dev, n) for n in NAMES}


def wrap(n):
    def f(*a, **k):
        calls.append((n, a, k))
        return real[n](*a, **k)
    return f


orig = {n: getattr(torch, n) for n in ("empty", "empty_like", "zeros", "zeros_like")}
orig_empty = F_._empty


def keeper(fn_):
    def f(*a, **k):
        t = fn_(*a, **k)
        keep.append(t)
        return t
    return f


for n in NAMES:
    setattr(dev, n, wrap(n))
for n, fn_ in orig.items():
    setattr(torch, n, keeper(fn_))
F_._empty = keeper(orig_empty)
est, _ = model(wav, emb)
loss = SISDRFn.apply(est, tgt, 1e-8)
loss.backward()
torch.cuda.synchronize()
for n in NAMES:
    setattr(dev, n, real[n])
for n, fn_ in orig.items():
    setattr(torch, n, fn_)
F_._empty = orig_empty

sig = torch.randn(4096, 512, generator=torch.Generator().manual_seed(3)).to(d)
ref = torch.view_as_real(torch.fft.rfft(sig)).clone()
s0, s1 = torch.cuda.Stream(device=d), torch.cuda.Stream(device=d)
torch.cuda.synchronize()
print(f"{len(calls)} launches recorded", flush=True)
summary = {}
for i, (n, a, k) in enumerate(calls):
    bad = 0
    try:
        for _ in range(N):
            with torch.cuda.stream(s0):
                real[n](*a, **k)
            with torch.cuda.stream(s1):
                o = torch.view_as_real(torch.fft.rfft(sig))
            torch.cuda.synchronize()
            bad += int(not torch.equal(o, ref))
    except Exception as e:  # noqa: BLE001
        print(f"#{i} {n}: replay failed: {e}")
        continue
    key = n
    if n in ("gemm_p2b", "gemm_b2p"):
        key += f"(K={k.get('K')},N={k.get('N')})"
    if n in ("gemm_nt", "gemm_tn"):
        key += f"(groups={k.get('ngroups', 0)},vec={k.get('vec')})"
    if n == "gemm_tnb":
        key += f"(g_cols={k.get('g_cols')},a={k.get('a0_cols', 0) + k.get('a1_cols', 0)})"
    if n in ("lstm_fwd", "lstm_bwd"):
        key += f"(mode={a[-1] if not k else k.get('mode', a[-1])})"
    s = summary.setdefault(key, [0, 0, 0])
    s[0] += 1
    s[1] += bad
    s[2] += N
for key, (nl, bad, tot) in sorted(summary.items(), key=lambda kv: -kv[1][1]):
    print(f"{key}: {nl} launches, canary mismatches {bad} of {tot}", flush=True)
