"""Shared pieces of the per-model bench tools (tools/bench_dpccn.py, tools/bench_tfgridnet.py): the `roofline` object
from the library's HIP-event timers + algorithmic work counters (wesep_amd.dev.ALG), and the `cpu_baseline` object
(the oracle -- CPU restatement of the reference, test infrastructure -- timed in a child process on a bounded sample)."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md
PEAK_BF16_TFLOPS = 2500.0      # dense


def roofline(dev, L, steps):
    """Dominant kernel class of the timed steps (largest HIP-event time among the library's four timed classes) with
    its algorithmic bytes / flops per launch (dev.ALG) -> achieved GB/s and ALGORITHMIC TFLOP/s (every fp32-equivalent
    product counted once: the kernels issue 1, 2 or 3 MFMAs per product depending on the class -- bench.py mfma_terms_census
    -- so a flat "x 3 executed" figure, as rounds 1-4 printed here, over-states the matrix cores' load) against the chip."""
    kinds = (("lstm_fwd", L.PROF_LSTM_FWD), ("lstm_bwd", L.PROF_LSTM_BWD), ("gemm_nt", L.PROF_GEMM_NT),
             ("gemm_tn", L.PROF_GEMM_TN))
    times = {name: dev.prof_collect(kind) for name, kind in kinds}
    per_step = {name: ms / steps for name, (ms, n) in times.items()}
    name = max(per_step, key=per_step.get)
    ms, n = times[name]
    alg = (dev.ALG or {}).get(name)
    out = {"kernel_ms_per_step": per_step, "kernel": name + " (all launches of that class)", "launches_per_step": n / steps,
           "ms_per_launch": ms / max(n, 1)}
    if alg and alg[2]:
        by, fl, calls = alg
        gbs = by / (ms * 1e-3) / 1e9 * (n / calls)      # counters and timers cover the same launches when calls == n
        tf = fl / (ms * 1e-3) / 1e12 * (n / calls)
        bound = "hbm" if gbs / PEAK_HBM_GBS >= tf / PEAK_BF16_TFLOPS else "mfma"
        out.update({"bound": bound, "achieved": gbs if bound == "hbm" else tf,
                    "peak": PEAK_HBM_GBS if bound == "hbm" else PEAK_BF16_TFLOPS,
                    "unit": "GB/s" if bound == "hbm" else "TFLOP/s (algorithmic: one per fp32-equivalent product)",
                    "frac": max(gbs / PEAK_HBM_GBS, tf / PEAK_BF16_TFLOPS),
                    "alg_gbs": gbs, "alg_tflops": tf, "bytes_per_launch": by / calls, "traffic": None,
                    "counted_launches": calls, "timed_launches": n})
    return out


def _worker(model, threads, budget_s):
    import torch
    sys.path.insert(0, ROOT)
    torch.set_num_threads(threads)
    from oracle import bsrnn_oracle as OB
    if model == "dpccn":
        from oracle import dpccn_oracle as O
        cfg = O.DPCCNConfig()
        fwd, T, what = O.dpccn_forward, 64000, "DPCCN default configuration, fixed embeddings"
    else:
        from oracle import tfgridnet_oracle as O
        cfg = O.TFGridNetConfig(n_fft=128, stride=64, n_layers=6, lstm_hidden_units=192, attn_n_head=4,
                                attn_approx_qk_dim=512, emb_dim=128, emb_ks=1, emb_hs=1)
        fwd, T, what = O.tfgridnet_forward, 96000, "TF-GridNet recipe geometry, fixed embeddings"
    p = {k: v.clone().requires_grad_(True) for k, v in O.synth_params(cfg, 0).items()}
    wav, tgt, emb = (t[:1].contiguous() for t in OB.synth_batch(2, T, 42))   # one row of the bench's synthetic pair

    def step():
        for t_ in p.values():
            t_.grad = None
        est = fwd(p, cfg, wav, emb)
        est = est[0] if isinstance(est, tuple) else est
        OB.sisdr_loss(est, tgt).backward()

    t0 = time.perf_counter()
    step()
    warm = time.perf_counter() - t0
    n = max(1, min(3, int(budget_s / max(warm, 1e-3)) - 1))
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    dt = (time.perf_counter() - t0) / n
    print(json.dumps({"value": 1.0 / dt, "unit": "utterances/s", "cores": threads, "kind": "port",
                      "sample": f"oracle (torch CPU fp32 restatement of the reference), {what}, 1 row x {T / 16000:.0f} s, "
                                f"1 warm-up + {n} timed fwd + SI-SDR + bwd, {dt:.2f} s/step, {threads} of {os.cpu_count()} host cores"}))


def cpu_baseline(model, budget_s=25.0, hard_limit_s=240.0):
    threads = min(os.cpu_count() or 1, 16)
    cmd = [sys.executable, "-c", f"import sys; sys.path.insert(0, {os.path.join(ROOT, 'tools')!r}); import bench_common as b; "
                                 f"b._worker({model!r}, {threads}, {budget_s})"]
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), HIP_VISIBLE_DEVICES="")
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=hard_limit_s, env=env, cwd=ROOT)
        return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    except Exception as e:
        return {"value": None, "unit": "utterances/s", "cores": threads, "kind": "port",
                "sample": f"cpu baseline did not finish within {hard_limit_s:.0f} s ({type(e).__name__})"}
