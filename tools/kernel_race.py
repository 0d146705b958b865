"""Diagnostic (GPU): kernel-level version of tools/stream_race.py.  Records every dev.* launch of an 'aggressor' stage
(ResRNN band view) and of a 'victim' stage (STFT + band split + BN; mask MLP + iSTFT), then replays each victim
launch on stream 1 while one aggressor launch loops on stream 0, and reports the pairs whose victim outputs differ
from the victim run alone."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import bsrnn_oracle as O  # noqa: E402
from wesep_amd import dev  # noqa: E402
from wesep_amd import functional as F_  # noqa: E402
from wesep_amd.models import get_model  # noqa: E402

d = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
R, T = 2, 24000
kw = dict(num_repeat=1, spk_fuse_type="multiply", multi_fuse=False)
cfg = O.BSRNNConfig(**kw)
model = get_model("BSRNN")(use_spk_transform=False, joint_training=False, **kw)
model.load_state_dict(O.synth_params(cfg, 1))
model.to(d).eval()
wav, tgt, emb = (t.to(d) for t in O.synth_batch(R, T, 1))
plan = model._plan(d)
os.environ["WESEP_LSTM_CLUSTER"] = "0"
with torch.no_grad():
    z0, xbs = F_.BandSplitFn.apply(wav, plan, *model._bn_params())
    bsnet = [m for m in model.separator.separation if hasattr(m, "band_rnn")][0]
    z1 = bsnet.band_rnn(z0, "time")
    bsnet.band_comm(z1, "band")
torch.cuda.synchronize()

LAUNCHERS = [n for n in dir(dev) if callable(getattr(dev, n)) and n in (
    "stft_bandsplit", "group_stats", "gemm_nt", "gemm_p2b", "gemm_b2p", "lstm_fwd", "lstm_fwd_fused", "lstm_pack",
    "lstm_cat_ih", "pack_w", "mask_istft_frames", "istft_ola", "affine_fwd", "lstm_pack_fused")]


def record(fn):
    calls, real = [], {n: getattr(dev, n) for n in LAUNCHERS}

    def wrap(n):
        def f(*a, **k):
            calls.append((n, a, k))
            return real[n](*a, **k)
        return f
    for n in LAUNCHERS:
        setattr(dev, n, wrap(n))
    keep = []
    orig_empty = F_._empty
    orig_t = {n: getattr(torch, n) for n in ("empty", "empty_like", "zeros", "zeros_like")}

    def keeper(fn_):
        def f(*a, **k):                # keep EVERY buffer alive: the replay uses the same addresses, so a freed output
            t = fn_(*a, **k)           # must not be handed to a later recording
            keep.append(t)
            return t
        return f
    F_._empty = keeper(orig_empty)
    for n, fn_ in orig_t.items():
        setattr(torch, n, keeper(fn_))
    try:
        with torch.no_grad():
            out = fn()
        torch.cuda.synchronize()
    finally:
        for n in LAUNCHERS:
            setattr(dev, n, real[n])
        F_._empty = orig_empty
        for n, fn_ in orig_t.items():
            setattr(torch, n, fn_)
    keep.append(out)
    return calls, keep, out


def tensors(a, k):
    return [t for t in list(a) + list(k.values()) if isinstance(t, torch.Tensor) and t.is_floating_point()]


agg, keep_a, _ = record(lambda: bsnet.band_comm(z1, "band"))
vic1, keep_v1, _ = record(lambda: F_.BandSplitFn.apply(wav, plan, *model._bn_params()))
vic2, keep_v2, _ = record(lambda: F_.MaskDecodeFn.apply(z1, xbs, plan, T, *model._mask_params()))
s0, s1 = torch.cuda.Stream(device=d), torch.cuda.Stream(device=d)
print("aggressor launches:", [c[0] for c in agg])
for tag, vic in (("bandsplit", vic1),) if os.environ.get("WESEP_B2P_VARIANT") else (("bandsplit", vic1), ("maskdecode", vic2)):
    print(f"victim stage {tag}:", [c[0] for c in vic], flush=True)
    for vi, (vn, va, vk) in enumerate(vic):
        real = getattr(dev, vn)
        real(*va, **vk)
        torch.cuda.synchronize()
        snap = [t.clone() for t in tensors(va, vk)]
        row = []
        for an, aa, ak in agg:
            areal = getattr(dev, an)
            bad = 0
            for _ in range(N):
                with torch.cuda.stream(s0):
                    areal(*aa, **ak)
                with torch.cuda.stream(s1):
                    real(*va, **vk)
                torch.cuda.synchronize()
                if any(not torch.equal(t, s_) for t, s_ in zip(tensors(va, vk), snap)):
                    bad += 1
                    real(*va, **vk)            # restore the victim's outputs for the next trial
                    torch.cuda.synchronize()
            row.append(bad)
        print(f"  victim #{vi} {vn}: mismatches beside each aggressor launch {row}", flush=True)

# ---- forensic: what lands in the victim's output when gemm_b2p runs beside stft_bandsplit? -------------------------
an, aa, ak = [c for c in agg if c[0] == "gemm_b2p"][0]
vn, va, vk = vic1[0]
aout, vout = ak["C_out"], va[2]
print(f"aggressor gemm_b2p: C_out {aout.data_ptr():#x} +{aout.numel() * 4:#x}, A {ak['A'].data_ptr():#x} "
      f"+{ak['A'].numel() * 4:#x}, R {ak['R'].data_ptr():#x}, Wpack {ak['Wpack'].data_ptr():#x}; ldc {ak['ldc']} "
      f"K {ak['K']} sm {ak['sm']}")
print(f"victim stft_bandsplit: wav {va[0].data_ptr():#x} +{va[0].numel() * 4:#x}, xbs {vout.data_ptr():#x} "
      f"+{vout.numel() * 4:#x}")
getattr(dev, vn)(*va, **vk)
torch.cuda.synchronize()
ref = vout.clone()
getattr(dev, an)(*aa, **ak)
torch.cuda.synchronize()
aref = aout.clone()
for trial in range(40):
    with torch.cuda.stream(s0):
        getattr(dev, an)(*aa, **ak)
    with torch.cuda.stream(s1):
        getattr(dev, vn)(*va, **vk)
    torch.cuda.synchronize()
    if not torch.equal(vout, ref):
        idx = torch.nonzero(vout.reshape(-1) != ref.reshape(-1)).reshape(-1)
        got, want = vout.reshape(-1)[idx], ref.reshape(-1)[idx]
        rows = torch.unique(idx // vout.shape[1])
        print(f"trial {trial}: {idx.numel()} elements differ, flat idx {idx[:6].tolist()} .. {idx[-3:].tolist()}, "
              f"{rows.numel()} rows {rows[:8].tolist()}; got {got[:4].tolist()} want {want[:4].tolist()}; "
              f"aggressor output intact: {torch.equal(aout, aref)}")
        hit = [int((aref.reshape(-1) == g).sum()) for g in got[:4]]
        print(f"   occurrences of the first wrong values in the aggressor's output: {hit}; zeros among wrong values: "
              f"{int((got == 0).sum())}; finite {bool(torch.isfinite(got).all())}")
        break
else:
    print("no mismatch in 40 forensic trials")
