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


def test_pair_bptt_beside_busy_streams_is_bit_identical():
    """The pair BPTT (lstm_pair.hip) has NO fall-back: its 128 workgroups wait for their partners with bounded polls, and a
    poll that expires poisons the launch.  In the training step it always runs beside the weight-gradient stream, and under
    DDP beside RCCL's kernels as well.  Here: the headline launch geometry (1024 sequences x 70 steps) repeated while (a) a
    second stream runs chip-filling GEMMs and (b) a third runs the library's own gemm_b2p -- every launch must finish
    without a timeout and be bit-identical to the launch made on an idle GPU."""
    from wesep_amd import dev, _lib as L
    from wesep_amd.functional import _view_maps
    d = _cuda()
    R, K, Tf, N, H = 32, 32, 70, 128, 256
    P = R * K * Tf
    geo, smap, seq, _ = _view_maps("time", R, K, Tf, N)
    g = torch.Generator().manual_seed(7)
    nb = dev.bl_num_blocks(seq)
    whf = (0.06 * torch.randn(4 * H, H, generator=g)).to(d)
    whr = (0.06 * torch.randn(4 * H, H, generator=g)).to(d)
    gates = dev.to_blocked(torch.randn(P, 8 * H, generator=g).to(d), seq)
    pf, pb = torch.empty(L.LSTM_PACK_FLOATS, device=d), torch.empty(L.LSTM_PACK_FLOATS, device=d)
    dev.lstm_pack(whf, whr, pf, pb, L.LSTM_BF16X3_BLK)
    cbuf, hcat = torch.zeros(nb, 2 * H // 4, 32, 4, device=d), torch.zeros(nb, 2 * H // 4, 32, 4, device=d)
    dev.lstm_fwd(gates, cbuf, hcat, pf, seq, L.LSTM_BF16X3_BLK)
    dh = dev.to_blocked((1e-2 * torch.randn(P, 2 * H, generator=g)).to(d), seq)
    pp = torch.empty(L.LSTM_PACK_FLOATS, device=d)
    dev.lstm_pack_pair(whf, whr, pp)
    status = torch.zeros(1, device=d, dtype=torch.int32)
    ref = gates.clone()
    assert int(dev.lstm_bwd_pair(ref, cbuf, dh, pp, seq, status=status).item()) == 0
    torch.cuda.synchronize()
    # aggressors: a torch GEMM stream and the library's gemm_b2p (the kernel of profiles/r03_kernel_race.md)
    a = torch.randn(6144, 6144, device=d)
    An = torch.randn(8192 * 41 * 256, device=d)
    wp = torch.empty(128 * 256, device=d)
    dev.pack_w(0.05 * torch.randn(128, 256, device=d), 128, 256, 256, wp, order=1)
    Cb = torch.empty(8192 * 41, 128, device=d)
    bseq = dev.SeqMap(nseq=8192, div=1 << 30, s1=0, s2=41, step_rows=1, L=41)
    s_mm, s_b2p = torch.cuda.Stream(device=d), torch.cuda.Stream(device=d)
    torch.cuda.synchronize()
    for trial in range(6):
        with torch.cuda.stream(s_mm):
            for _ in range(4):
                a2 = a @ a
        with torch.cuda.stream(s_b2p):
            for _ in range(40):
                dev.gemm_b2p(A=An, K=256, sm=bseq, Wpack=wp, C_out=Cb, ldc=128)
        g2 = gates.clone()
        tw = dev.lstm_bwd_pair(g2, cbuf, dh, pp, seq, status=status)
        torch.cuda.synchronize()
        assert int(tw.item()) == 0 and int(status.item()) == 0, trial
        assert torch.equal(g2, ref), trial
    del a2
