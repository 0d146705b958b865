#!/usr/bin/env python
"""Headline benchmark: utterances/sec (4 s, 16 kHz, 2-speaker rows) of one pBSRNN training
step -- forward, SI-SDR, backward, per-tensor clip, Adam -- on N MI355X.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1] / north_star "batch 32 x 4 s"): pBSRNN, FiLM multi-fuse,
6 repeats, feature_dim 128, fixed [R,256] embeddings, R = 32 rows (16 two-speaker mixtures) of
64000 samples per GPU.  Arithmetic: split-bf16 ("bf16x3": fp32 operands as bf16 hi+lo, three bf16
MFMAs per product, fp32 accumulation and fp32 storage) -- holds the reference's fp32 results to
~1e-4 on the separated waveforms (tests/test_bsrnn_gpu.py; tolerance 1e-3).  Weak scaling: every rank runs the same per-GPU batch on its own synthetic
rows (seed 42 + rank); DDP all-reduces gradients over RCCL.  Prints ONE JSON line on rank 0.
"""
import argparse
import contextlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ROWS = 32
T = 64000
MODEL_KW = dict(spk_emb_dim=256, sr=16000, win=512, stride=128, feature_dim=128, num_repeat=6,
                use_spk_transform=False, spk_fuse_type="FiLM", multi_fuse=True, joint_training=False)
LR0, LR1, WD, CLIP = 1e-3, 2.5e-5, 1e-4, 5.0          # confs/bsrnn.yaml:95-114
BF16_MFMA_PEAK_TFLOPS = 2500.0                         # MI355X_MICROARCH.md chip table (dense)
HBM_PEAK_GBS = 8000.0                                  # HBM3E spec peak (6.3 TB/s measured copy)


def _cpu_baseline_worker(threads, budget_s):
    """Runs in a child process (hard wall-clock limit enforced by the parent)."""
    from oracle import bsrnn_oracle as O
    torch.set_num_threads(threads)
    cfg = O.BSRNNConfig(num_repeat=6, spk_fuse_type="FiLM", multi_fuse=True)
    params = O.synth_params(cfg, 0)
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    m = {k: torch.zeros_like(v) for k, v in params.items()}
    v2 = {k: torch.zeros_like(v) for k, v in params.items()}
    R = 2
    wav, tgt, emb = O.synth_batch(R, T, 42)

    def step(i):
        for t_ in p.values():
            t_.grad = None
        est = O.bsrnn_forward(p, cfg, wav, emb)
        loss = O.sisdr_loss(est, tgt)
        loss.backward()
        grads = {k: t_.grad for k, t_ in p.items()}
        O.clip_gradients_(grads, CLIP)
        with torch.no_grad():
            for k in p:
                O.adam_l2_step_(p[k], grads[k], m[k], v2[k], i, LR0, weight_decay=WD)

    t0 = time.perf_counter()
    step(1)                                  # warm-up (thread pools, oneDNN primitives)
    warm = time.perf_counter() - t0
    n = max(1, min(3, int(budget_s / max(warm, 1e-3)) - 1))
    t0 = time.perf_counter()
    for i in range(n):
        step(i + 2)
    dt = (time.perf_counter() - t0) / n
    print(json.dumps({"value": R / dt, "unit": "utterances/s", "cores": torch.get_num_threads(),
                      "kind": "port",
                      "sample": f"oracle (torch CPU fp32 restatement of the reference step), R={R} rows x 4 s, "
                                f"1 warm-up + {n} timed steps of fwd+SI-SDR+bwd+clip+Adam, {dt:.2f} s/step, "
                                f"{threads} of {os.cpu_count()} host cores (reference recipe: OMP_NUM_THREADS=8)"}))


def _cpu_reference_worker(threads, budget_s):
    """The REFERENCE'S OWN modules timed on the host: wesep.models.bsrnn.BSRNN imported from
    /root/reference (third-party imports stubbed, oracle/ref_import.py), its own clip_gradients, torch.optim.Adam; the
    SI-SDR loss is the restatement (auraloss is absent everywhere).  Only possible where /root/reference exists -- the
    authoring container, not the GPU box -- so it is an explicit flag (--cpu-baseline reference), used once to check
    that the oracle port times like the code it restates."""
    from oracle import bsrnn_oracle as O
    from oracle.ref_import import import_reference
    get_model = import_reference()
    from wesep.utils.funcs import clip_gradients        # wesep/utils/funcs.py:79-88
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    kw = dict(MODEL_KW)
    model = get_model("BSRNN")(**kw).train()
    opt = torch.optim.Adam(model.parameters(), lr=LR0, weight_decay=WD)
    R = 2
    wav, tgt, emb = O.synth_batch(R, T, 42)

    def step():
        est, _ = model(wav, emb)
        loss = O.sisdr_loss(est, tgt)
        opt.zero_grad()
        loss.backward()
        clip_gradients(model, CLIP)
        opt.step()

    t0 = time.perf_counter()
    step()
    warm = time.perf_counter() - t0
    n = max(1, min(3, int(budget_s / max(warm, 1e-3)) - 1))
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    dt = (time.perf_counter() - t0) / n
    print(json.dumps({"value": R / dt, "unit": "utterances/s", "cores": torch.get_num_threads(), "kind": "reference",
                      "sample": f"wesep.models.bsrnn.BSRNN imported from /root/reference (torch CPU fp32), R={R} rows x "
                                f"4 s, 1 warm-up + {n} timed steps of fwd+SI-SDR+bwd+clip_gradients+Adam, {dt:.2f} "
                                f"s/step, {threads} of {os.cpu_count()} host cores"}))


def cpu_baseline(budget_s=30.0, hard_limit_s=240.0, kind="port"):
    """Oracle (CPU port of the reference step) timed on this host's cores in a child process so a
    pathological host (hundreds of cores, oversubscription) cannot stall the benchmark.

    Cores (BASELINE.md section 3 asks for the host's cores, count stated): the step is timed twice on a bounded sample --
    with 16 threads (twice the reference recipe's OMP_NUM_THREADS=8; what a torch CPU LSTM over 2 rows x 32 bands can keep
    busy) and with every core of the host (capped at 128) -- and the FASTER run is reported, with both in `sample`: on the
    256-core GPU hosts the all-core run is slower (oversubscribed oneDNN / intra-op pools on 501 sequential steps)."""
    import subprocess
    ncpu = os.cpu_count() or 1
    worker = "_cpu_reference_worker" if kind == "reference" else "_cpu_baseline_worker"
    tries = sorted({min(ncpu, 16), min(ncpu, 128)})
    runs = []
    for threads in tries:
        cmd = [sys.executable, "-c",
               f"import sys; sys.path.insert(0, {ROOT!r}); import bench; bench.{worker}({threads}, {budget_s / len(tries)})"]
        env = dict(os.environ, OMP_NUM_THREADS=str(threads), HIP_VISIBLE_DEVICES="")
        try:
            out = subprocess.run(cmd, capture_output=True, text=True, timeout=hard_limit_s / len(tries), env=env, cwd=ROOT)
            runs.append(json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1]))
        except Exception as e:  # timeout or failure: report it rather than hiding it
            runs.append({"value": None, "unit": "utterances/s", "cores": threads, "kind": kind,
                         "sample": f"cpu baseline with {threads} threads did not finish within "
                                   f"{hard_limit_s / len(tries):.0f} s ({type(e).__name__})"})
    ok = [r for r in runs if r["value"]]
    if not ok:
        return runs[0]
    best = max(ok, key=lambda r: r["value"])
    if len(runs) > 1:
        best = dict(best, sample=best["sample"] + "; the faster of the runs with " +
                    " / ".join(f"{r['cores']} threads ({r['value']:.3f} utt/s)" if r["value"] else f"{r['cores']} threads (failed)"
                               for r in runs))
    return best


def _latest_profile(kind):
    """profiles/rNN_<kind>.json of the highest round."""
    import glob
    hits = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_{kind}.json")))
    if not hits:
        raise OSError(kind)
    return hits[-1]


def _sha16(path):
    import hashlib
    return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rows", type=int, default=ROWS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline", choices=("port", "reference"), default="port",
                    help="port: the oracle (travels to the GPU box); reference: the reference's own modules imported "
                         "from /root/reference (only where that exists)")
    ap.add_argument("--cpu-only", action="store_true", help="print the CPU baseline object and exit (no GPU needed)")
    ap.add_argument("--joint", action="store_true",
                    help="the shipped confs/bsrnn.yaml variant: speaker encoder (wespeaker ResNet34 on 80-d fbank, "
                         "398 frames) trained jointly instead of fixed 256-d embeddings; not the headline line")
    args = ap.parse_args()

    if args.cpu_only:
        print(json.dumps(cpu_baseline(kind=args.cpu_baseline)), flush=True)
        return
    from wesep_amd import dev
    from wesep_amd import _lib as L
    from wesep_amd.models import get_model
    from wesep_amd.optim import FusedClipAdam
    from wesep_amd.parallel import all_ranks, all_ranks_tensor_spread, barrier, comm_info, init_distributed, max_over_ranks, rank_seed, wrap_ddp
    from wesep_amd.utils.losses import parse_loss
    from wesep_amd.utils.schedulers import ExponentialDecrease
    from wesep_amd.utils.synthetic import synth_batch

    rank, local_rank, world = init_distributed()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local_rank)
    d = torch.device("cuda", local_rank)

    torch.manual_seed(0)                      # identical default-init weights on every rank
    kw = dict(MODEL_KW)
    if args.joint:
        kw.update(joint_training=True, spk_model="ResNet34", spk_feat=True,
                  spk_args=dict(feat_dim=80, embed_dim=256, pooling_func="TSTP", two_emb_layer=False))
    model = get_model("BSRNN")(**kw)
    with torch.no_grad():                     # FiLM is zero-init in the reference; make it do work
        for mod in model.separator.separation:
            if hasattr(mod, "fc") and hasattr(mod.fc, "gamma_fcs"):
                torch.nn.init.normal_(mod.fc.gamma_fcs[0].weight, std=0.02)
                torch.nn.init.normal_(mod.fc.beta_fcs[0].weight, std=0.02)
    model = model.to(d).train()
    ddp = wrap_ddp(model, local_rank)
    opt = FusedClipAdam(model.parameters(), lr=LR0, weight_decay=WD, clip_grad=CLIP)
    sched = ExponentialDecrease(opt, num_epochs=150, epoch_iter=1000, initial_lr=LR0, final_lr=LR1,
                                warm_up_epoch=0)
    crit = parse_loss("SISDR")[0]
    R = args.rows
    wav, tgt, emb = (t.to(d) for t in synth_batch(R, T, rank_seed(42, rank)))
    if args.joint:                            # fbank-like enrollment [R, 398, 80], CMN'd
        fb = torch.randn(R, 398, 80, generator=torch.Generator().manual_seed(rank_seed(43, rank)))
        emb = (fb - fb.mean(1, keepdim=True)).to(d)

    def step(i):
        sched.step(i)
        est, _ = ddp(wav, emb)
        loss = crit(est, tgt).mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss

    # WESEP_MAIN_PRIORITY (experiment): run the step on a stream of that HIP priority instead of the default stream
    main_pr = os.environ.get("WESEP_MAIN_PRIORITY")
    main_ctx = torch.cuda.stream(torch.cuda.Stream(device=d, priority=int(main_pr))) if main_pr else contextlib.nullcontext()
    with main_ctx:
        for i in range(args.warmup):
            step(i)
        torch.cuda.synchronize()
        barrier()
        dev.prof_enable(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            loss = step(args.warmup + i)
        torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    dev.prof_enable(False)
    per_rank_ms = all_ranks(elapsed / args.steps * 1e3, d)       # rank-ordered, for the SCALE record
    # outside the timed region: do all replicas hold the same parameters?  (Data parallelism keeps them bit-identical;
    # a disturbed all-reduce -- profiles/r02_kernel_race.md -- would not.)  Checksum of every parameter, max - min over ranks.
    with torch.no_grad():
        chk = torch.stack([p.detach().double().sum() for p in model.parameters()])
    spread = all_ranks_tensor_spread(chk, d)
    elapsed = max_over_ranks(elapsed, d)
    final_loss = float(loss.item())

    # dominant kernels: the two BLSTM recurrences (HIP events on the launch stream, see runtime.hip)
    K, Tf = 32, 1 + T // 128
    P = R * K * Tf
    flops_per_launch = 2.0 * P * 2 * L.LSTM_H * 4 * L.LSTM_H        # fp32-equivalent, both directions
    # algorithmic HBM bytes of one recurrence launch: per position, direction and hidden unit the BPTT
    # reads 4 saved gates, c, d(h) (c_{t-1} is the next step's c: L2) and writes 4 d(gates): 40 B with the round-3 format
    # (fp32 gates, split-pair d(gates)), 24 B with the default WS_GATES_H2 (unorm16 gates, bf16 d(gates)), 32 B with H2S;
    # the forward reads 4 fp32 pre-activations (time view; the band view's fused projection reads the 128-wide input
    # instead: 1 B per cell) and writes 4 gates + c + h: 40 B before, 32 / 17 B now (launch average 24.5)
    gfmt = dev.gates_fmt()
    bwd_cell = dev._bptt_bytes(gfmt)
    fwd_cell = 40.0 if gfmt == L.GATES_F32 else 24.5
    cell_bytes = {"lstm_bwd": float(bwd_cell), "lstm_fwd": fwd_cell}
    prof = {}
    for name, kind in (("lstm_fwd", L.PROF_LSTM_FWD), ("lstm_bwd", L.PROF_LSTM_BWD),
                       ("gemm_nt", L.PROF_GEMM_NT), ("gemm_tn", L.PROF_GEMM_TN)):
        ms, n = dev.prof_collect(kind)
        prof[name] = {"ms_total": ms, "launches": n, "ms_avg": ms / max(n, 1)}
    dom = "lstm_bwd" if prof["lstm_bwd"]["ms_total"] >= prof["lstm_fwd"]["ms_total"] else "lstm_fwd"
    bytes_per_launch = cell_bytes[dom] * P * 2 * L.LSTM_H
    sec = prof[dom]["ms_avg"] * 1e-3 if prof[dom]["launches"] else float("inf")
    gbs = bytes_per_launch / sec / 1e9
    tfl = flops_per_launch / sec / 1e12
    # HBM bytes per launch of that kernel from the committed rocprofv3 PMC passes (tools/pmc_summary.py);
    # bench.py cannot collect counters itself, so this is the profile of the same command, or null
    traffic, pmc_src, noop = None, {}, set()
    try:
        pmc_file = _latest_profile("pmc_traffic")
        doc = json.load(open(pmc_file))
        pmc = doc["kernels"]
        pmc_src["traffic"] = {"file": os.path.relpath(pmc_file, ROOT), **doc.get("collected", {})}
        # streaming (32/16-sequence), pair and cluster variants; NOT the predicated fall-back launches of the pair BPTT
        # (lstm_bwd_s16 with run_if = 0 returns at once: kilobytes per launch, no HIP-event interval either)
        noop = {k for k, v in pmc.items() if dom in k and v["hbm_bytes_per_launch_corrected"] < 1e6}
        hit = [v for k, v in pmc.items() if dom in k and k not in noop]
        n = sum(v["launches"] for v in hit)
        traffic = sum(v["hbm_bytes_per_launch_corrected"] * v["launches"] for v in hit) / n if n else None
    except (OSError, KeyError, ValueError):
        pass

    # MFMA-busy fraction of the same kernels from the committed SQ_VALU_MFMA_BUSY_CYCLES pass (tools/pmc_mfma_summary.py)
    mfma_busy = None
    try:
        pm_file = _latest_profile("pmc_mfma")
        doc = json.load(open(pm_file))
        pm = doc["kernels"]
        pmc_src["mfma"] = {"file": os.path.relpath(pm_file, ROOT), **doc.get("collected", {})}
        hit = [v for k, v in pm.items() if dom in k and k not in noop]
        n = sum(v["launches"] for v in hit)
        mfma_busy = sum(v["mfma_busy_frac"] * v["launches"] for v in hit) / n if n else None
    except (OSError, KeyError, ValueError):
        pass

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        out = {
            "metric": "utterances/sec (4 s, 16 kHz, 2-spk) fwd+bwd, pBSRNN",
            "value": world * R * args.steps / elapsed, "unit": "utterances/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16x3",
            "data": "synthetic",
            "config": {"workload": "pBSRNN FiLM multi-fuse, 6 repeats, feature_dim 128, " +
                                   ("jointly trained wespeaker ResNet34 speaker encoder on [R, 398, 80] fbank"
                                    if args.joint else "fixed 256-d embeddings") +
                                   "; fwd + SI-SDR + bwd + per-tensor clip + Adam-L2; split-bf16 "
                                   "products (3 bf16 MFMAs), fp32 accumulate + storage",
                       "rows_per_gpu": R, "global_rows": world * R, "samples_per_row": T,
                       "parallelism": f"dp{world}", "final_loss_dB": final_loss},
            # the recurrence kernels are bound by memory paths (HBM activations + the per-step L2 weight
            # stream), not by the matrix cores: HBM is the roof they are priced against
            "roofline": {"bound": "hbm", "kernel": dom + " recurrence kernels, launch average over the time view (" +
                                   ("lstm_fwd_cluster_kernel" if dom == "lstm_fwd" else
                                    ("lstm_bwd_pair_kernel" if os.environ.get("WESEP_LSTM_PAIR_BWD", "1") != "0"
                                     else "lstm_bwd_s16_kernel")) +
                                   ") and the band view (" + dom + "_bf16_kernel<BLK>)",
                         "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                         "traffic": traffic, "bytes_per_launch": bytes_per_launch,
                         "bytes_per_cell": cell_bytes[dom],
                         "gates_format": {L.GATES_F32: "f32", L.GATES_H2: "h2b (unorm16 gates, bf16 d(gates))",
                                          L.GATES_H2S: "h2s (unorm16 gates, split-pair d(gates))",
                                          L.GATES_H2F: "h2 (unorm16 gates, scaled-fp16 d(gates))"}[gfmt],
                         # the same launches priced at round 3's 40 B per cell (fp32 gates, split-pair d(gates)), for
                         # comparison with the earlier rounds' lines: bytes that no longer move are not achieved bandwidth
                         "frac_at_round3_bytes": 40.0 * P * 2 * L.LSTM_H / sec / 1e9 / HBM_PEAK_GBS,
                         "ms_per_launch": prof[dom]["ms_avg"],
                         "mfma": {"alg_tflops": tfl, "executed_bf16_tflops": 3 * tfl,
                                  "peak_bf16_tflops": BF16_MFMA_PEAK_TFLOPS,
                                  "frac_executed": 3 * tfl / BF16_MFMA_PEAK_TFLOPS,
                                  "busy_frac_pmc": mfma_busy}},
            "kernel_ms_per_step": {k: v["ms_total"] / args.steps for k, v in prof.items()},
        }
        # counters cannot be collected inside this run: say which run they come from (commit + bench.py hash at
        # collection time, written by tools/pmc_summary.py), next to this run's own bench.py hash
        out["roofline"]["counters_from"] = pmc_src
        out["bench_py_sha16"] = _sha16(os.path.abspath(__file__))
        # whole step against SURVEY 8d's algorithmic work (per row, 4 s, fwd+bwd): 1.017 TFLOP fp32-equivalent
        # (x3 executed as split-bf16) and the layer-fused minimum of 2.26 GB of fp32 traffic
        sec_step = elapsed / args.steps
        alg_flops, alg_bytes = 1.017e12 * R, 2.26e9 * R
        t_mfma = 3 * alg_flops / (BF16_MFMA_PEAK_TFLOPS * 1e12)
        t_hbm = alg_bytes / (HBM_PEAK_GBS * 1e9)
        out["step_roofline"] = {
            "alg_flops_per_step": alg_flops, "alg_bytes_per_step": alg_bytes,
            "alg_tflops": alg_flops / sec_step / 1e12, "executed_bf16_tflops": 3 * alg_flops / sec_step / 1e12,
            "frac_mfma_executed": t_mfma / sec_step, "alg_gbs": alg_bytes / sec_step / 1e9,
            "frac_hbm_alg": t_hbm / sec_step, "bound": "mfma" if t_mfma >= t_hbm else "hbm",
            "frac": max(t_mfma, t_hbm) / sec_step,
            "frac_mfma_algorithmic": alg_flops / (BF16_MFMA_PEAK_TFLOPS * 1e12) / sec_step,
            "note": "per GPU; SURVEY 8d: achieved := max(bytes_alg / 8 TB/s, 3 * flops_alg / 2.5 PFLOP/s) / t_step -- "
                    "`frac` / `frac_mfma_executed` count the 3 bf16 MFMAs EXECUTED per fp32 product against the bf16 peak; "
                    "`frac_mfma_algorithmic` counts each product once"}
        out["per_rank_ms_per_step"] = per_rank_ms
        out["replicas_in_sync"] = bool(spread == 0.0)
        out["replica_checksum_spread"] = spread
        out["comm"] = comm_info()
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(kind=args.cpu_baseline)
        print(json.dumps(out), flush=True)
    barrier()
    # a scaling record of replicas that drifted apart, or of fewer ranks than asked for, is not a measurement
    if comm_info()["world_size"] != args.gpus or spread != 0.0:
        sys.stderr.write(f"bench.py: world_size {comm_info()['world_size']} (asked for {args.gpus}), replica checksum "
                         f"spread {spread}: FAILED\n")
        sys.exit(3)


if __name__ == "__main__":
    main()
