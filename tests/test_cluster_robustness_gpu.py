"""GPU: the weight-stationary cluster recurrence (lstm_cluster.hip) needs all of its workgroups co-resident, which the
launcher can check against the CU count only -- not against CUs held by other streams or processes.  Every wait in the
kernel is bounded; a timeout poisons the launch's outputs, sets its timeout word, and the streaming kernels enqueued
behind it (predicated on that word: `run_if`, include/wesep_hip.h) recompute the layer on the device.  These tests pin
that contract: a forced timeout must still yield the layer's result (never NaN), a clean launch must not run the
fall-back, and a launch beside a busy second stream must be correct whichever way it went."""
import warnings

import pytest
import torch

pytestmark = pytest.mark.gpu

NAMES = ("rnn.weight_ih_l0", "rnn.weight_hh_l0", "rnn.bias_ih_l0", "rnn.bias_hh_l0", "rnn.weight_ih_l0_reverse",
         "rnn.weight_hh_l0_reverse", "rnn.bias_ih_l0_reverse", "rnn.bias_hh_l0_reverse", "proj.weight", "proj.bias")


def _cuda():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _params(seed, d):
    g = torch.Generator().manual_seed(seed)
    shapes = {"norm.weight": (128,), "norm.bias": (128,), "rnn.weight_ih_l0": (1024, 128), "rnn.weight_hh_l0": (1024, 256),
              "rnn.bias_ih_l0": (1024,), "rnn.bias_hh_l0": (1024,), "rnn.weight_ih_l0_reverse": (1024, 128),
              "rnn.weight_hh_l0_reverse": (1024, 256), "rnn.bias_ih_l0_reverse": (1024,),
              "rnn.bias_hh_l0_reverse": (1024,), "proj.weight": (128, 512), "proj.bias": (128,)}
    p = {k: (0.06 * torch.randn(s, generator=g)).to(d) for k, s in shapes.items()}
    p["norm.weight"] = (1.0 + 0.1 * torch.randn(128, generator=g)).to(d)
    return p


def _run(p, z):
    from wesep_amd import functional as F0
    with torch.no_grad():
        out = F0.ResRNNBlkFn.apply(z, None, None, None, "time", p["norm.weight"], p["norm.bias"], *(p[n] for n in NAMES))
    torch.cuda.synchronize()
    return out


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


@pytest.mark.parametrize("R,Tf", [(2, 70), (32, 65)])
def test_forced_timeout_is_repaired_on_the_device(monkeypatch, R, Tf):
    from wesep_amd import dev
    d = _cuda()
    p = _params(3, d)
    z = torch.randn(R, 32, Tf, 128, generator=torch.Generator().manual_seed(R)).to(d)
    assert dev.lstm_cluster_ok(dev.SeqMap(R * 32, dev.BIG, 0, Tf, 1, Tf), d)
    monkeypatch.setenv("WESEP_LSTM_CLUSTER", "0")
    stream_only = _run(p, z)                                 # the streaming kernels alone
    monkeypatch.setenv("WESEP_LSTM_CLUSTER", "1")
    before = dev.poll_cluster_status(d, block=True)
    clean = _run(p, z)
    assert dev.poll_cluster_status(d, block=True) == before  # clean launch: no timeout, fall-back launches were empty
    assert rel(clean, stream_only) < 4e-5 and not torch.equal(clean, stream_only)   # (different MFMA order)
    monkeypatch.setenv("WESEP_CLUSTER_FORCE_TIMEOUT", "1")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        forced = _run(p, z)
        after = dev.poll_cluster_status(d, block=True)
    assert torch.isfinite(forced).all()
    assert torch.equal(forced, stream_only)                  # the fall-back IS the streaming path, bit for bit
    assert after == before + 1                               # ... and the timeout was counted / reported


def test_cluster_launch_beside_a_busy_stream_is_correct():
    """256 workgroups (R = 32) need every CU; a second stream keeps the chip busy with large GEMMs while the cluster
    kernel is launched.  Whether the grid became resident in time or the bounded waits expired and the predicated
    streaming kernels took over, the result must be the layer's result."""
    from wesep_amd import dev
    d = _cuda()
    p = _params(5, d)
    z = torch.randn(32, 32, 65, 128, generator=torch.Generator().manual_seed(9)).to(d)
    quiet = _run(p, z)
    side = torch.cuda.Stream(device=d)
    a = torch.randn(8192, 8192, device=d)
    torch.cuda.synchronize()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        for _ in range(3):
            with torch.cuda.stream(side):
                for _ in range(6):
                    a2 = a @ a                               # ~1.1 TFLOP each: tens of ms of full-chip work
            busy = _run(p, z)
            side.synchronize()
            assert torch.isfinite(busy).all()
            assert rel(busy, quiet) < 4e-5
        dev.poll_cluster_status(d, block=True)
    del a2
