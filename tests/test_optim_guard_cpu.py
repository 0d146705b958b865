"""CPU: the host logic of wesep_amd.optim.FusedClipAdam on the emulation of its two launches (tests/emu_optim.py) --
per-tensor clip + Adam-L2 against the oracle's step (wesep/utils/funcs.py:79-88, wesep/bin/train.py:237-238), the
non-finite-gradient guard (round 4: the WHOLE update is skipped, weights and moments intact, the step is counted), and
the same under DistributedDataParallel at world size 2 on gloo: a NaN born on ONE rank reaches every rank through the
all-reduce, every rank skips the same step, the replicas stay bit-identical and train on."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _net(seed):
    torch.manual_seed(seed)
    return torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.Tanh(), torch.nn.Linear(8, 3))


def test_fused_clip_adam_host_logic_matches_the_oracle_step(monkeypatch):
    from oracle import bsrnn_oracle as O
    from tests import emu_optim
    from wesep_amd.optim import FusedClipAdam
    emu_optim.install(monkeypatch)
    net = _net(0)
    ref = {k: v.detach().clone() for k, v in net.named_parameters()}
    m = {k: torch.zeros_like(v) for k, v in ref.items()}
    v2 = {k: torch.zeros_like(v) for k, v in ref.items()}
    opt = FusedClipAdam(net.parameters(), lr=1e-2, weight_decay=1e-3, clip_grad=0.05)
    g = torch.Generator().manual_seed(1)
    for step in range(1, 6):
        x, y = torch.randn(16, 6, generator=g), torch.randn(16, 3, generator=g)
        opt.zero_grad()
        ((net(x) - y) ** 2).mean().backward()
        grads = {k: p.grad.detach().clone() for k, p in net.named_parameters()}
        opt.step()
        O.clip_gradients_(grads, 0.05)
        for k in ref:
            O.adam_l2_step_(ref[k], grads[k], m[k], v2[k], step, 1e-2, weight_decay=1e-3)
        for k, p in net.named_parameters():
            assert torch.allclose(p.detach(), ref[k], rtol=1e-6, atol=1e-7), (step, k)
    assert len(opt.last_grad_norms()) == 4 and opt.skipped_steps == 0


def test_nonfinite_gradient_skips_the_whole_update_and_is_counted(monkeypatch):
    from tests import emu_optim
    from wesep_amd.optim import FusedClipAdam
    emu_optim.install(monkeypatch)
    net = _net(2)
    opt = FusedClipAdam(net.parameters(), lr=1e-2, weight_decay=1e-3, clip_grad=5.0)
    x, y = torch.randn(4, 6), torch.randn(4, 3)

    def backward():
        opt.zero_grad()
        ((net(x) - y) ** 2).mean().backward()

    backward()
    opt.step()                                                   # a finite step: state exists now
    snap = lambda: ([p.detach().clone() for p in net.parameters()],
                    [opt.state[p]["exp_avg"].clone() for p in net.parameters()],
                    [opt.state[p]["exp_avg_sq"].clone() for p in net.parameters()])
    before = snap()
    backward()
    list(net.parameters())[2].grad[1, 3] = float("nan")          # one element of one tensor
    with pytest.warns(RuntimeWarning, match="not finite"):
        opt.step()
        opt._poll_guard(torch.device("cpu"), block=True)         # (on the GPU the count arrives one poll later, without a sync)
    after = snap()
    for a, b in zip(before, after):
        assert all(torch.equal(s, t) for s, t in zip(a, b))      # every weight and both moments of EVERY tensor intact
    assert opt.skipped_steps == 1
    backward()
    opt.step()                                                   # the next finite step trains on
    assert not torch.equal(snap()[0][0], before[0][0]) and opt.skipped_steps == 1


def test_skipped_step_does_not_advance_the_bias_correction_and_checkpoints_carry_applied_steps(monkeypatch):
    """torch.optim.Adam under a GradScaler does not call step() on an overflow; here the host counts attempted steps, the
    device corrects by its own lag word (ws_clip_adam_step step_lag, ws_guard_commit) and the host reconciles one poll late:
    the trajectory equals the oracle's that never saw the bad step -- whether or not the host has polled in between."""
    from oracle import bsrnn_oracle as O
    from tests import emu_optim
    from wesep_amd.optim import FusedClipAdam
    emu_optim.install(monkeypatch)
    for poll_between in (False, True):
        net = _net(4)
        ref = {k: v.detach().clone() for k, v in net.named_parameters()}
        m = {k: torch.zeros_like(v) for k, v in ref.items()}
        v2 = {k: torch.zeros_like(v) for k, v in ref.items()}
        opt = FusedClipAdam(net.parameters(), lr=1e-2, weight_decay=1e-3, clip_grad=0.05)
        g = torch.Generator().manual_seed(5)
        applied = 0
        for it in range(1, 9):
            x, y = torch.randn(16, 6, generator=g), torch.randn(16, 3, generator=g)
            opt.zero_grad()
            ((net(x) - y) ** 2).mean().backward()
            bad = it in (2, 5, 6)
            if bad:
                list(net.parameters())[0].grad[0, 0] = float("inf")
            grads = {k: p.grad.detach().clone() for k, p in net.named_parameters()}
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore", RuntimeWarning)
                opt.step()
                if poll_between:
                    opt._poll_guard(torch.device("cpu"), block=True)
            if not bad:
                applied += 1
                O.clip_gradients_(grads, 0.05)
                for k in ref:
                    O.adam_l2_step_(ref[k], grads[k], m[k], v2[k], applied, 1e-2, weight_decay=1e-3)
            for k, p in net.named_parameters():
                assert torch.allclose(p.detach(), ref[k], rtol=1e-6, atol=1e-7), (poll_between, it, k)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            sd = opt.state_dict()
        assert opt.skipped_steps == 3
        assert all(int(st["step"]) == applied for st in sd["state"].values()), [st["step"] for st in sd["state"].values()]


def test_consecutive_skipped_steps_raise(monkeypatch):
    from tests import emu_optim
    from wesep_amd import _lib as L
    from wesep_amd.optim import FusedClipAdam
    emu_optim.install(monkeypatch)
    net = _net(6)
    opt = FusedClipAdam(net.parameters(), lr=1e-2, clip_grad=5.0, max_consecutive_skips=4)
    x, y = torch.randn(4, 6), torch.randn(4, 3)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        with pytest.raises(L.WesepHipError, match="all skipped"):
            for it in range(12):
                opt.zero_grad()
                ((net(x) - y) ** 2).mean().backward()
                list(net.parameters())[1].grad[0] = float("nan")
                opt.step()
                opt._poll_guard(torch.device("cpu"), block=True)
        assert it == 3 and opt.skipped_steps == 4
        # a finite step in between resets the run of skips
        opt2 = FusedClipAdam(net.parameters(), lr=1e-2, clip_grad=5.0, max_consecutive_skips=3)
        for it in range(10):
            opt2.zero_grad()
            ((net(x) - y) ** 2).mean().backward()
            if it % 3 != 2:
                list(net.parameters())[1].grad[0] = float("nan")
            opt2.step()
            opt2._poll_guard(torch.device("cpu"), block=True)
        assert opt2.skipped_steps == 7


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from _pytest.monkeypatch import MonkeyPatch
    from tests import emu_optim
    from wesep_amd.optim import FusedClipAdam
    from wesep_amd.parallel import init_distributed
    init_distributed(backend="gloo")
    mp_ = MonkeyPatch()
    net = _net(5)                                                # same initial weights on both ranks
    ddp = torch.nn.parallel.DistributedDataParallel(net)
    emu_optim.install(mp_)                                       # (after DDP's construction: it looks at the real device type)
    opt = FusedClipAdam(net.parameters(), lr=1e-2, weight_decay=1e-3, clip_grad=5.0)
    g = torch.Generator().manual_seed(100 + rank)                # rank-sharded rows
    trace = []
    for step in range(4):
        x, y = torch.randn(8, 6, generator=g), torch.randn(8, 3, generator=g)
        opt.zero_grad()
        loss = ((ddp(x) - y) ** 2).mean()
        if step == 1 and rank == 1:
            loss = loss * float("nan")                           # the NaN is born on ONE rank ...
        loss.backward()                                          # ... and the all-reduce spreads it
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            opt.step()
            opt._poll_guard(torch.device("cpu"), block=True)
        trace.append([p.detach().clone() for p in net.parameters()])
    torch.save({"trace": trace, "skipped": opt.skipped_steps}, os.path.join(out, f"r{rank}.pt"))
    mp_.undo()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(300)
def test_a_nan_on_one_rank_is_skipped_by_every_rank_world2_gloo(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = (torch.load(tmp_path / f"r{i}.pt") for i in range(2))
    assert r0["skipped"] == r1["skipped"] == 1
    for s in range(4):
        for a, b in zip(r0["trace"][s], r1["trace"][s]):
            assert torch.equal(a, b) and bool(torch.isfinite(a).all())      # replicas bit-identical, never poisoned
    assert all(torch.equal(a, b) for a, b in zip(r0["trace"][0], r0["trace"][1]))      # step 1 changed nothing
    assert not torch.equal(r0["trace"][1][0], r0["trace"][2][0])                       # step 2 trained on


class _TseToy(torch.nn.Module):
    """(wav_mix, enrollment) -> (estimate,): the call shape Executor.train drives."""

    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(32, 32)

    def forward(self, wav, emb):
        return (self.a(wav) + emb[:, :1],)


def _replica_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from _pytest.monkeypatch import MonkeyPatch
    from tests import emu_optim
    from wesep_amd.optim import FusedClipAdam
    from wesep_amd.parallel import init_distributed
    from wesep_amd.utils.executor import Executor, ReplicaDivergence
    from wesep_amd.utils.schedulers import ExponentialDecrease
    init_distributed(backend="gloo")
    mp_ = MonkeyPatch()
    torch.manual_seed(3)
    net = _TseToy()
    ddp = torch.nn.parallel.DistributedDataParallel(net)
    emu_optim.install(mp_)
    opt = FusedClipAdam(net.parameters(), lr=1e-3)
    sched = ExponentialDecrease(opt, num_epochs=1, epoch_iter=8, initial_lr=1e-3, final_lr=1e-4, warm_up_epoch=0)
    g = torch.Generator().manual_seed(10 + rank)

    def batches():
        for i in range(8):
            if i == 3 and rank == 1:                             # a silent corruption on ONE rank, mid-epoch
                with torch.no_grad():
                    net.a.bias[0] += 1e-3
            yield {"wav_mix": torch.randn(4, 32, generator=g), "wav_targets": torch.randn(4, 32, generator=g),
                   "spk_embeds": torch.randn(4, 8, generator=g), "spk_label": torch.zeros(0)}

    seen = None
    ex = Executor()
    try:
        ex.train(batches(), [ddp], 8, [opt], [torch.nn.MSELoss(reduction="none")], [sched], scaler=None, epoch=1, enable_amp=False,
                 logger=None, clip_grad=5.0, device=torch.device("cpu"), se_loss_weight=([[0]], [[1.0]]),
                 replica_check_interval=1)
    except ReplicaDivergence as e:
        seen = str(e)
    torch.save({"seen": seen, "steps": ex.step}, os.path.join(out, f"d{rank}.pt"))
    mp_.undo()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(300)
def test_in_loop_replica_check_reports_a_divergence_within_the_interval_world2_gloo(tmp_path):
    """the data-parallel replica check also runs INSIDE the epoch (Executor.train replica_check_interval) when
    every rank runs the same number of steps: a parameter disturbed on one rank during step 3 stops BOTH ranks at step 3,
    not at the end of the epoch."""
    world, port = 2, _free_port()
    mp.spawn(_replica_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = (torch.load(tmp_path / f"d{i}.pt") for i in range(2))
    # (the prefetcher pulls batch i + 1 while step i runs: the disturbance lands during step 3)
    assert r0["seen"] and r1["seen"] and "step 3" in r0["seen"] and "step 3" in r1["seen"], (r0, r1)
    assert r0["steps"] == r1["steps"] == 3


def test_clip_gradients_zeroes_a_poisoned_step_for_optimizers_without_a_guard(monkeypatch):
    """wesep_amd.optim.clip_gradients in front of a stock torch optimizer (wesep/utils/funcs.py:79-88 +
    wesep/bin/train.py:237-238 when the optimizer is not Adam).  The scaled-fp16 d(gates) are not clamped since round 5, so an
    overflow arrives as Inf: clip / (Inf + eps) = 0, Inf * 0 = NaN -- the reference's arithmetic would write NaN into the weights.
    Here the clip launch is skipped, every gradient of the step is zeroed, the step is counted, and the weights stay finite."""
    from tests import emu_optim
    from wesep_amd import optim
    emu_optim.install(monkeypatch)
    monkeypatch.setattr(optim.clip_gradients, "skipped_steps", 0, raising=False)
    net = _net(3)
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    x, y = torch.randn(4, 6), torch.randn(4, 3)
    ((net(x) - y) ** 2).mean().backward()
    norms = optim.clip_gradients(net, 0.05)                      # a finite step: clipped per tensor, nothing skipped
    assert all(n == n and n != float("inf") for n in norms) and optim.clip_gradients.skipped_steps == 0
    assert all(float(p.grad.norm()) <= 0.05 + 1e-6 for p in net.parameters())
    opt.step()
    before = [p.detach().clone() for p in net.parameters()]
    opt.zero_grad()
    ((net(x) - y) ** 2).mean().backward()
    list(net.parameters())[0].grad[2, 1] = float("inf")
    with pytest.warns(RuntimeWarning, match="not finite"):
        norms = optim.clip_gradients(net, 0.05)
    assert any(n == float("inf") for n in norms) and optim.clip_gradients.skipped_steps == 1
    assert all(not p.grad.any() for p in net.parameters())       # every gradient of the step is zero, none is NaN
    opt.step()
    assert all(torch.equal(a, p.detach()) for a, p in zip(before, net.parameters()))   # SGD on zero gradients: intact


def test_step_fence_bounds_the_run_ahead():
    """dev.StepFence: after step n is enqueued the host waits for step n - depth, never for a later one (round 6: an unbounded
    run-ahead cost 49 GB of allocator growth per step and multi-second hipMalloc calls inside the timed region)."""
    from wesep_amd import dev
    log = []

    class Ev:
        n = 0

        def __init__(self):
            self.i = Ev.n
            Ev.n += 1

        def record(self):
            log.append(("rec", self.i))

        def synchronize(self):
            log.append(("sync", self.i))

    for depth, want in ((1, [None, 0, 1, 2]), (0, [0, 1, 2, 3]), (2, [None, None, 0, 1]), (-1, [None] * 4)):
        Ev.n = 0
        f = dev.StepFence(depth=depth, make_event=Ev)
        got = []
        for _ in range(4):
            del log[:]
            f.fence()
            syncs = [i for k, i in log if k == "sync"]
            assert len(syncs) <= 1
            got.append(syncs[0] if syncs else None)
            if depth >= 0:
                assert log[0][0] == "rec"          # the step's own event is recorded before anything is waited for
        assert got == want, (depth, got)
