"""GPU: a BPTT launch whose bounded wait times out must not cost the run.

The time-view BPTT runs on PAIRS of co-resident workgroups (lstm_pair.hip); whatever holds compute units during the
backward -- under DistributedDataParallel RCCL's resident all-reduce kernels do -- can keep a partner off the chip.
Round 3: the kernel poisoned d(gates) with NaN in place and the optimizer applied them before the host noticed.  Now:
  * with the 2-byte formats (default) the pair BPTT writes d(gates) out of place and the streaming BPTT stands behind it,
    predicated on the launch's time-out word: the layer is recomputed on the device, nothing is lost;
  * ws_grad_norms raises a guard word on any non-finite gradient and ws_clip_adam_step skips the whole update of that step
    on the device (also what protects the OTHER ranks of a data-parallel job after the all-reduce);
  * the in-place formats' status word is a skip word of the update too."""
import time
import warnings

import pytest
import torch

pytestmark = pytest.mark.gpu

N = 128


def _cuda():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _resrnn_grads(d, view="time", dims=(2, 32, 70), seed=5):
    from wesep_amd.models.bsrnn import ResRNN
    torch.manual_seed(seed)
    R, K, Tf = dims
    blk = ResRNN(N, 2 * N).to(d)
    z = torch.randn(R, K, Tf, N, device=d)
    go = torch.randn(R, K, Tf, N, device=d)
    zd = z.clone().requires_grad_(True)
    blk(zd, view).backward(go)
    torch.cuda.synchronize()
    return zd.grad.clone(), {k: p.grad.clone() for k, p in blk.named_parameters()}


@pytest.mark.parametrize("fmt", ["h2", "h2s"])
def test_pair_timeout_is_repaired_on_the_device(fmt, monkeypatch):
    """A forced time-out of the pair BPTT (test build of the kernel: pair 0 gives up at step 2 and poisons its output):
    the predicated streaming BPTT behind the launch recomputes the layer -- the gradients are those of the streaming
    kernel, bit for bit, and nothing is NaN."""
    from wesep_amd import dev
    d = _cuda()
    monkeypatch.setenv("WESEP_GATES", fmt)
    monkeypatch.setenv("WESEP_LSTM_PAIR_BWD", "0")
    ref = _resrnn_grads(d)                              # the streaming BPTT itself
    monkeypatch.setenv("WESEP_LSTM_PAIR_BWD", "1")
    monkeypatch.setenv("WESEP_PAIR_FORCE_TIMEOUT", "1")
    before = dev.poll_cluster_status(d, block=True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        got = _resrnn_grads(d)
        after = dev.poll_cluster_status(d, block=True)  # must NOT raise: the time-out was repaired
    assert after > before                               # ... and it was counted
    assert not torch.isnan(got[0]).any()
    assert torch.equal(got[0], ref[0])
    for k in ref[1]:
        assert torch.equal(got[1][k], ref[1][k]), k
    monkeypatch.delenv("WESEP_PAIR_FORCE_TIMEOUT")
    clean = _resrnn_grads(d)                            # the pair kernel proper: same result up to its summation order
    assert float((clean[0] - ref[0]).norm() / ref[0].norm()) < 1e-4


def _tiny_opt(d):
    from wesep_amd.optim import FusedClipAdam
    torch.manual_seed(1)
    ps = [torch.nn.Parameter(torch.randn(300, 70, device=d)), torch.nn.Parameter(torch.randn(1000, device=d))]
    opt = FusedClipAdam(ps, lr=1e-2, weight_decay=1e-4, clip_grad=5.0)
    return ps, opt


def test_nonfinite_gradient_skips_the_update_on_the_device():
    d = _cuda()
    ps, opt = _tiny_opt(d)
    for p in ps:
        p.grad = torch.randn_like(p)
    opt.step()                                           # a clean step: state exists, weights moved
    torch.cuda.synchronize()
    snap = [p.detach().clone() for p in ps]
    st = [(opt.state[p]["exp_avg"].clone(), opt.state[p]["exp_avg_sq"].clone()) for p in ps]
    for p in ps:
        p.grad = torch.randn_like(p)
    ps[1].grad[123] = float("nan")                       # ONE poisoned element in ONE tensor
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        opt.step()
        torch.cuda.synchronize()
        for p, s0, (m0, v0) in zip(ps, snap, st):        # the WHOLE step was skipped: every tensor, weights and moments
            assert torch.equal(p.detach(), s0)
            assert torch.equal(opt.state[p]["exp_avg"], m0) and torch.equal(opt.state[p]["exp_avg_sq"], v0)
        assert opt._poll_guard(d, block=True) >= 1
    for p in ps:
        p.grad = torch.randn_like(p)
    opt.step()                                           # finite again: training goes on
    torch.cuda.synchronize()
    assert all(not torch.equal(p.detach(), s0) for p, s0 in zip(ps, snap))
    assert all(torch.isfinite(p).all() for p in ps)
    ps[0].grad[0, 0] = float("inf")
    opt.step()
    torch.cuda.synchronize()
    assert all(torch.isfinite(p).all() for p in ps)


def test_skipped_steps_do_not_advance_adam_bias_correction_on_the_device():
    """ABI v17 (ws_clip_adam_step step_lag, ws_guard_commit): the host counts attempted steps and never waits; the kernels
    use step - lag.  Eight steps with three poisoned ones must land where torch.optim.Adam lands after the five finite ones
    -- with and without the host's asynchronous reconciliation in between -- and the checkpoint carries five steps."""
    d = _cuda()
    for poll_between in (False, True):
        ps, opt = _tiny_opt(d)
        ref = [torch.nn.Parameter(p.detach().clone()) for p in ps]
        ropt = torch.optim.Adam(ref, lr=1e-2, weight_decay=1e-4)
        g = torch.Generator(device="cpu").manual_seed(3)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            for it in range(1, 9):
                grads = [torch.randn(p.shape, generator=g).to(d) * 0.01 for p in ps]     # (per-tensor norms 1.4 / 0.3: nothing is clipped at 5.0)
                bad = it in (2, 5, 6)
                for p, gr in zip(ps, grads):
                    p.grad = gr.clone()
                if bad:
                    ps[0].grad[3, 3] = float("inf")
                opt.step()
                if poll_between:
                    opt._poll_guard(d, block=True)
                if not bad:
                    for p, gr in zip(ref, grads):
                        p.grad = gr.clone()
                    ropt.step()
            torch.cuda.synchronize()
            for p, r in zip(ps, ref):
                assert torch.allclose(p.detach(), r.detach(), rtol=2e-6, atol=2e-7), (poll_between, float((p - r).abs().max()))
            sd = opt.state_dict()
        assert opt.skipped_steps == 3
        assert all(int(st["step"]) == 5 for st in sd["state"].values())


def test_consecutive_skips_raise_on_the_device_count():
    from wesep_amd import _lib as L
    from wesep_amd.optim import FusedClipAdam
    d = _cuda()
    p = torch.nn.Parameter(torch.randn(64, device=d))
    opt = FusedClipAdam([p], lr=1e-2, max_consecutive_skips=3)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        with pytest.raises(L.WesepHipError, match="all skipped"):
            for it in range(10):
                p.grad = torch.full_like(p, float("nan"))
                opt.step()
                opt._poll_guard(d, block=True)
    assert it == 2 and torch.isfinite(p).all()


def test_inplace_bptt_timeout_never_reaches_the_weights(monkeypatch):
    """The ABI <= 14 format keeps d(gates) in place: a pair time-out cannot be repaired.  The update is skipped on the device
    (status word + non-finite guard), the weights stay intact, the host raises at its next look."""
    from wesep_amd import _lib as L
    from wesep_amd import dev
    from wesep_amd.models.bsrnn import ResRNN
    from wesep_amd.optim import FusedClipAdam
    d = _cuda()
    monkeypatch.setenv("WESEP_GATES", "f32")
    monkeypatch.setenv("WESEP_PAIR_FORCE_TIMEOUT", "1")
    torch.manual_seed(3)
    blk = ResRNN(N, 2 * N).to(d)
    opt = FusedClipAdam(blk.parameters(), lr=1e-3, clip_grad=5.0)
    snap = {k: p.detach().clone() for k, p in blk.named_parameters()}
    z = torch.randn(2, 32, 70, N, device=d, requires_grad=True)
    blk(z, "time").sum().backward()
    torch.cuda.synchronize()
    assert any(torch.isnan(p.grad).any() for p in blk.parameters())      # the poison is there ...
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        try:
            opt.step()
        except L.WesepHipError:
            pass
        torch.cuda.synchronize()
    for k, p in blk.named_parameters():                                  # ... and did not reach a single weight
        assert torch.equal(p.detach(), snap[k]), k
    with pytest.raises(L.WesepHipError):
        dev.poll_cluster_status(d, block=True)
    dev.poll_cluster_status(d, block=True)                               # the word was cleared with the report


@pytest.mark.parametrize("held", [32, 64])
def test_backward_with_compute_units_held_by_a_resident_kernel(held):
    """The headline step's backward (32 rows x 4 s, FiLM multi-fuse, 6 repeats) while a resident kernel holds 32 / 64 CUs
    on a third stream for the whole backward (ws_debug_occupy: 112 KB of LDS per workgroup -- no pair-BPTT, cluster or
    weight-gradient workgroup fits beside it; the shape of RCCL's all-reduce kernels): the step completes, no bounded wait
    times out, and the gradients are bit-identical to the undisturbed run's.  Prints the backward times."""
    from oracle import bsrnn_oracle as O
    from tests.test_bsrnn_gpu import _build
    from wesep_amd import dev
    from wesep_amd.functional import SISDRFn
    d = _cuda()
    kw = dict(num_repeat=6, spk_fuse_type="FiLM", multi_fuse=True)
    cfg, params, model = _build(kw, 16, d)
    model.train()
    wav, tgt, emb = (t.to(d) for t in O.synth_batch(32, 64000, 16))
    side = torch.cuda.Stream(device=d)
    stop = torch.zeros(1, device=d, dtype=torch.int32)

    def run(nheld):
        for p in model.parameters():
            p.grad = None
        est, _ = model(wav, emb)
        loss = SISDRFn.apply(est, tgt, 1e-8)
        torch.cuda.synchronize()
        stop.zero_()
        torch.cuda.synchronize()
        if nheld:
            with torch.cuda.stream(side):
                dev.debug_occupy(nheld, 1500000, stop)       # until released below (bounded at 1.5 s)
            time.sleep(0.02)                                  # the occupant is resident before the backward starts
        t0 = time.perf_counter()
        loss.backward()
        torch.cuda.current_stream().synchronize()
        dt = time.perf_counter() - t0
        stop.fill_(1)
        torch.cuda.synchronize()
        return dt, {k: p.grad.clone() for k, p in model.named_parameters()}

    run(0)                                                    # warm-up (weight packs, allocator)
    base_fb = dev.poll_cluster_status(d, block=True)
    t_free, g_free = run(0)
    t_held, g_held = run(held)
    fb = dev.poll_cluster_status(d, block=True)               # raises on an unrepaired time-out
    print(f"backward at 32 rows x 4 s: {t_free * 1e3:.1f} ms undisturbed, {t_held * 1e3:.1f} ms with {held} CUs held; "
          f"repaired time-outs {fb - base_fb}")
    for k in g_free:
        assert torch.equal(g_free[k], g_held[k]), k


@pytest.mark.parametrize("held", [8, 64])
def test_forward_with_compute_units_held_by_a_resident_kernel(held):
    """the same occupant over the FORWARD.  `ws_lstm_fwd_cluster2` needs 256 of 256 CUs co-resident; with
    8 / 64 of them held for the whole forward its bounded waits expire, the launch poisons its outputs and sets its time-out
    word, and the predicated streaming kernels behind it recompute the layer on the device (include/wesep_hip.h).  The
    forward must complete, every value must be finite and within the cross-kernel tolerance of the undisturbed forward (the
    fall-back is the three-term streaming arithmetic, not the cluster kernel's), a free chip must not run the fall-back, and
    the repaired time-outs are counted.  Prints the forward times: what a co-tenant costs the first N > 1 run if RCCL's
    kernels ever stay resident across a forward."""
    import warnings
    from oracle import bsrnn_oracle as O
    from tests.test_bsrnn_gpu import _build
    from wesep_amd import dev
    d = _cuda()
    kw = dict(num_repeat=6, spk_fuse_type="FiLM", multi_fuse=True)
    cfg, params, model = _build(kw, 16, d)
    model.train()
    wav, tgt, emb = (t.to(d) for t in O.synth_batch(32, 64000, 16))
    side = torch.cuda.Stream(device=d)
    stop = torch.zeros(1, device=d, dtype=torch.int32)

    def run(nheld):
        torch.cuda.synchronize()
        stop.zero_()
        torch.cuda.synchronize()
        if nheld:
            with torch.cuda.stream(side):
                dev.debug_occupy(nheld, 1500000, stop)       # until released below (bounded at 1.5 s)
            time.sleep(0.02)                                  # the occupant is resident before the forward starts
        t0 = time.perf_counter()
        est, _ = model(wav, emb)                              # the training forward (grad mode: what the step runs)
        torch.cuda.current_stream().synchronize()
        dt = time.perf_counter() - t0
        stop.fill_(1)
        torch.cuda.synchronize()
        return dt, est.detach().clone()

    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)       # the first repaired time-out warns once
        run(0)                                                # warm-up (weight packs, allocator)
        fb0 = dev.poll_cluster_status(d, block=True)
        t_free, e_free = run(0)
        fb1 = dev.poll_cluster_status(d, block=True)
        t_held, e_held = run(held)
        fb2 = dev.poll_cluster_status(d, block=True)          # raises on an unrepaired time-out
    r = float((e_held.double() - e_free.double()).norm() / e_free.double().norm())
    print(f"forward at 32 rows x 4 s: {t_free * 1e3:.1f} ms undisturbed, {t_held * 1e3:.1f} ms with {held} CUs held; "
          f"repaired time-outs {fb2 - fb1} (free chip: {fb1 - fb0}); est rel {r:.2e}")
    assert fb1 == fb0
    assert torch.isfinite(e_held).all()
    assert r < 1e-4
