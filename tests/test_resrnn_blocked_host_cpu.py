"""CPU: numerics of the pBSRNN ResRNN on the blocked layout (functional.ResRNNBlkFn -- the production path of the
headline metric) on the blocked-layout emulation (tests/emu_blk.py), against the oracle's ResRNN with torch autograd:
output, input gradient and all twelve parameter gradients, both views, in the 16-sequence, cluster and
fused-projection branches.  The kernels are covered by tests/test_bsrnn_gpu.py; this pins the HOST composition
(sequence / statistics maps, pack orders, gradient routing, weight-gradient shifts) so that it can be refactored
without a GPU."""
import pytest
import torch

from oracle import bsrnn_oracle as O
from tests import emu_blk, emu_dev

NAMES = ("rnn.weight_ih_l0", "rnn.weight_hh_l0", "rnn.bias_ih_l0", "rnn.bias_hh_l0", "rnn.weight_ih_l0_reverse",
         "rnn.weight_hh_l0_reverse", "rnn.bias_ih_l0_reverse", "rnn.bias_hh_l0_reverse", "proj.weight", "proj.bias")


@pytest.fixture
def emu(monkeypatch):
    emu_dev.install(monkeypatch)
    emu_blk.install(monkeypatch)
    monkeypatch.setattr(torch.Tensor, "is_cuda", property(lambda self: True))
    monkeypatch.setenv("WESEP_WGRAD_OVERLAP", "0")


def _params(seed):
    g = torch.Generator().manual_seed(seed)
    shapes = {"norm.weight": (128,), "norm.bias": (128,), "rnn.weight_ih_l0": (1024, 128), "rnn.weight_hh_l0": (1024, 256),
              "rnn.bias_ih_l0": (1024,), "rnn.bias_hh_l0": (1024,), "rnn.weight_ih_l0_reverse": (1024, 128),
              "rnn.weight_hh_l0_reverse": (1024, 256), "rnn.bias_ih_l0_reverse": (1024,),
              "rnn.bias_hh_l0_reverse": (1024,), "proj.weight": (128, 512), "proj.bias": (128,)}
    p = {k: (0.06 * torch.randn(s, generator=g)) for k, s in shapes.items()}
    p["norm.weight"] = 1.0 + 0.1 * torch.randn(128, generator=g)
    return {k: v.requires_grad_(True) for k, v in p.items()}


# storage format of the saved gates / d(gates) (wesep_hip.h WS_GATES_*) -> tolerance of the gradients: the fp32 format is
# exact in the emulation; unorm16 gates cost ~1e-5; scaled-fp16 d(gates) (H2F, the default "h2") 2^-12 per element of
# d(gates), bf16 d(gates) ("h2b") 2^-9
GATE_FORMATS = [("f32", 2e-4), ("h2s", 2e-4), ("h2", 5e-4), ("h2b", 3e-3)]


@pytest.mark.parametrize("fmt,gtol", GATE_FORMATS)
@pytest.mark.parametrize("view,R,K,Tf,branch", [
    ("time", 1, 3, 9, "16-sequence"), ("time", 2, 32, 66, "cluster"), ("band", 2, 4, 5, "16-sequence"),
    ("band", 2, 3, 2100, "fused projection")])
def test_resrnn_blocked_matches_oracle(emu, monkeypatch, view, R, K, Tf, branch, fmt, gtol):
    from wesep_amd import dev
    from wesep_amd import functional as F0
    monkeypatch.setenv("WESEP_GATES", fmt)
    p = _params(R * 100 + K)
    g = torch.Generator().manual_seed(Tf)
    z = torch.randn(R, K, Tf, 128, generator=g).requires_grad_(True)
    probe = torch.randn(R, K, Tf, 128, generator=g)
    _, _, seq, _ = F0._view_maps(view, R, K, Tf, 128)
    cluster = dev.lstm_cluster_ok(seq, torch.device("cpu"))
    assert ("cluster" in branch) == cluster and ("fused" in branch) == dev.lstm_fuse_ok(seq.nseq, cluster)
    out = F0.ResRNNBlkFn.apply(z, None, None, None, view, p["norm.weight"], p["norm.bias"], *(p[n] for n in NAMES))
    (out * probe).sum().backward()
    got = {"z": z.grad.clone(), **{k: v.grad.clone() for k, v in p.items()}}
    z.grad = None
    for v in p.values():
        v.grad = None
    if view == "time":          # sequences (r, k) over t
        x = z.permute(0, 1, 3, 2).reshape(R * K, 128, Tf)
        ref = O.res_rnn(p, "", x).view(R, K, 128, Tf).permute(0, 1, 3, 2)
    else:                       # sequences (r, t) over k
        x = z.permute(0, 2, 3, 1).reshape(R * Tf, 128, K)
        ref = O.res_rnn(p, "", x).view(R, Tf, 128, K).permute(0, 3, 1, 2)
    (ref * probe).sum().backward()
    # the cluster branch of the 2-byte formats is ws_lstm_fwd_cluster2 (round 5): fp16 h (11 bits) in the recurrent product,
    # the x-projection on the fp16 copy of the normalised input -- 2^-12 per operand element instead of the emulation's fp32
    c2 = cluster and fmt != "f32" and dev.lstm_cluster2_on()
    assert float((out - ref).norm() / ref.norm()) < (1e-4 if c2 else 1e-5)
    want = {"z": z.grad, **{k: v.grad for k, v in p.items()}}
    for k in want:
        assert float((got[k] - want[k]).norm()) <= max(gtol, 5e-4 if c2 else 0.0) * float(want[k].norm()) + 1e-6, k


def test_streaming_bptt_arithmetic_and_its_own_input_gradient_reach_the_kernel(emu, monkeypatch):
    """Round 6: the band view's streaming BPTT runs ws_lstm_args.rfmt = 3 by default (the FP8 pack, ws_lstm_pack_bwd_f8, the lo
    term on the FP8 matrix instruction; rfmt 2 -- both terms on the fp16 MFMA -- under WESEP_BAND_DX=1); with
    WESEP_BAND_DX=1 ("default" below: the opt-in) d(xn) is computed by the BPTT launch itself (ws_lstm_args.dxn +
    ws_lstm_pack_dx_f8, ABI v19: no ws_gemm_b2p over d(gates), the fused GroupNorm backward adds the two directions) -- 32-sequence
    blocked kernels with the default 2-byte format only.  WESEP_BAND_RF=0 restores the three-term product.  Gradients
    against the oracle within the format's tolerance in all three modes (the emulation models the stored fp16 d(gates) and the
    16-bit weights of both products)."""
    from wesep_amd import dev
    from wesep_amd import functional as F0
    monkeypatch.setenv("WESEP_GATES", "h2")
    R, K, Tf = 2, 4, 2100
    seen = []
    real_bwd, real_pack, real_b2p, real_px = dev.lstm_bwd, dev.lstm_pack_bwd_f8, dev.gemm_b2p, dev.lstm_pack_dx_f8
    monkeypatch.setattr(dev, "lstm_bwd", lambda *a, **k: (seen.append(("bwd", k.get("rfmt", 0), k.get("dxn") is not None)),
                                                          real_bwd(*a, **k))[1])
    monkeypatch.setattr(dev, "lstm_pack_bwd_f8", lambda *a, **k: (seen.append(("pack8",)), real_pack(*a, **k))[1])
    monkeypatch.setattr(dev, "lstm_pack_dx_f8", lambda *a, **k: (seen.append(("packdx",)), real_px(*a, **k))[1])
    monkeypatch.setattr(dev, "gemm_b2p", lambda **k: (seen.append(("b2p", k.get("a_fmt", 0))), real_b2p(**k))[1])
    grads = {}
    for mode, env in (("default", {"WESEP_BAND_DX": "1"}), ("gemm", {}), ("rf0", {"WESEP_BAND_RF": "0"})):
        for k_ in ("WESEP_BAND_DX", "WESEP_BAND_RF"):
            monkeypatch.delenv(k_, raising=False)
        for k_, v_ in env.items():
            monkeypatch.setenv(k_, v_)
        seen.clear()
        p = _params(31)
        g = torch.Generator().manual_seed(5)
        z = torch.randn(R, K, Tf, 128, generator=g).requires_grad_(True)
        probe = torch.randn(R, K, Tf, 128, generator=g)
        out = F0.ResRNNBlkFn.apply(z, None, None, None, "band", p["norm.weight"], p["norm.bias"], *(p[n] for n in NAMES))
        (out * probe).sum().backward()
        grads[mode] = {"z": z.grad.clone(), **{k: v.grad.clone() for k, v in p.items()}}
        assert (("pack8",) in seen) == (mode != "rf0") and (("packdx",) in seen) == (mode == "default")
        assert [r[1:] for r in seen if r[0] == "bwd"] == [({"rf0": 0, "default": 2, "gemm": 3}[mode], mode == "default")]
        # the d(xn) GEMM over the scaled-fp16 d(gates) (a_fmt 3: the lo term on the FP8 MFMA, the round-6 default of
        # functional.dxn_fmt) runs exactly when the BPTT did not write d(xn) itself
        assert (("b2p", 3) in seen) == (mode != "default") and ("b2p", 2) not in seen
    for v in p.values():
        v.grad = None
    z2 = z.detach().clone().requires_grad_(True)
    ref = O.res_rnn(p, "", z2.permute(0, 2, 3, 1).reshape(R * Tf, 128, K)).view(R, Tf, 128, K).permute(0, 3, 1, 2)
    (ref * probe).sum().backward()
    want = {"z": z2.grad, **{k: v.grad for k, v in p.items()}}
    for k in want:
        for mode in grads:
            assert float((grads[mode][k] - want[k]).norm()) <= 5e-4 * float(want[k].norm()) + 1e-6, (k, mode)
    assert any(not torch.equal(grads["rf0"][k], grads["gemm"][k]) for k in want)      # (the variable does reach the arithmetic)
    # the BPTT's own d(xn) and the GEMM's differ only in the weight's low bits (FP8 vs fp16 remainder) and the summation order
    assert float((grads["default"]["z"] - grads["gemm"]["z"]).norm()) <= 2e-5 * float(grads["gemm"]["z"].norm())
    monkeypatch.setenv("WESEP_BAND_RF", "1")
    with pytest.raises(ValueError):
        F0.band_rfmt(3, 4)


def _oracle_time(p, z, R, K, Tf):
    x = z.permute(0, 1, 3, 2).reshape(R * K, 128, Tf)
    return O.res_rnn(p, "", x).view(R, K, 128, Tf).permute(0, 1, 3, 2)


def test_cluster_timeout_falls_back_to_the_streaming_kernels(emu, monkeypatch):
    """A cluster launch that times out (emulated: NaN-poisoned outputs + its timeout word set, as lstm_cluster.hip does)
    must be repaired by the predicated gemm_p2b + lstm_fwd launches behind it -- and those must be no-ops after a
    clean launch (the emulation would otherwise apply the recurrence twice to the activated gates)."""
    from wesep_amd import functional as F0
    R, K, Tf = 2, 32, 66
    p = _params(7)
    z = torch.randn(R, K, Tf, 128, generator=torch.Generator().manual_seed(3))
    ref = _oracle_time(p, z, R, K, Tf)
    for force in ("0", "1"):
        monkeypatch.setenv("WESEP_CLUSTER_FORCE_TIMEOUT", force)
        with torch.no_grad():
            out = F0.ResRNNBlkFn.apply(z, None, None, None, "time", p["norm.weight"], p["norm.bias"], *(p[n] for n in NAMES))
        assert torch.isfinite(out).all()
        # (clean launch: ws_lstm_fwd_cluster2's fp16 h / fp16 xn; forced time-out: the streaming kernels' fp32 emulation)
        assert float((out - ref).norm() / ref.norm()) < (1e-4 if force == "0" else 1e-5), force


def test_pack_cache_follows_the_weights_and_second_backward_is_refused(emu):
    from wesep_amd import dev
    from wesep_amd import functional as F0
    from wesep_amd import _lib as L
    R, K, Tf = 1, 3, 9
    p = _params(11)
    z = torch.randn(R, K, Tf, 128, generator=torch.Generator().manual_seed(5))
    cache = F0.PackCache()
    args = lambda: (z, None, None, cache, "time", p["norm.weight"], p["norm.bias"], *(p[n] for n in NAMES))
    out1 = F0.ResRNNBlkFn.apply(*args())
    items1 = dict(cache.items)
    out2 = F0.ResRNNBlkFn.apply(*args())
    assert torch.equal(out1, out2) and all(cache.items[k] is v for k, v in items1.items())     # packs reused
    with torch.no_grad():
        p["rnn.weight_hh_l0"].mul_(0.5)                                                        # version counter moves
    out3 = F0.ResRNNBlkFn.apply(*args())
    assert float((out3 - _oracle_time(p, z, R, K, Tf)).norm() / out3.norm()) < 1e-5            # rebuilt, not stale
    p["proj.weight"].data.mul_(2.0)          # raw write, invisible to torch (what ws_clip_adam_step does) ...
    dev.bump_weight_epoch()                  # ... announced the way dev.clip_adam_step announces it
    out4 = F0.ResRNNBlkFn.apply(*args())
    assert float((out4 - _oracle_time(p, z, R, K, Tf)).norm() / out4.norm()) < 1e-5
    # a second backward through one graph would differentiate the d(gates) the first one left in the saved buffer
    out4.sum().backward(retain_graph=True)
    with pytest.raises(L.WesepHipError, match="second backward"):
        out4.sum().backward()


@pytest.mark.parametrize("view,R,K,Tf", [("time", 2, 32, 66), ("band", 2, 3, 2100)])
def test_fp16_copies_of_the_weight_gradient_operand_touch_the_lstm_weight_gradients_only(emu, monkeypatch, view, R, K, Tf):
    """ABI v16 (round 4): with the default storage format the weight-gradient GEMM reads fp16 copies of [xn | h] that
    gemm_p2b / the output projection write on the way (WESEP_TNB_A16, default on).  Against the split-pair operand: output,
    input gradient, norm and proj gradients bit for bit (nothing else reads the copies), the LSTM weight gradients to fp16
    rounding of the operand (2^-12 per element, averaged over the positions), the bias gradients bit for bit (column sums
    of d(gates))."""
    from wesep_amd import functional as F0
    monkeypatch.setenv("WESEP_GATES", "h2")
    res = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("WESEP_TNB_A16", flag)
        p = _params(R * 100 + K)
        g = torch.Generator().manual_seed(Tf)
        z = torch.randn(R, K, Tf, 128, generator=g).requires_grad_(True)
        probe = torch.randn(R, K, Tf, 128, generator=g)
        out = F0.ResRNNBlkFn.apply(z, None, None, None, view, p["norm.weight"], p["norm.bias"], *(p[n] for n in NAMES))
        (out * probe).sum().backward()
        res[flag] = (out.detach().clone(), {"z": z.grad.clone(), **{k: v.grad.clone() for k, v in p.items()}})
    assert torch.equal(res["1"][0], res["0"][0])
    for k, g0 in res["0"][1].items():
        g1 = res["1"][1][k]
        if "weight_ih" in k or "weight_hh" in k:
            assert not torch.equal(g1, g0) and float((g1 - g0).norm()) <= 4e-4 * float(g0.norm()), k      # (2^-12 = 2.4e-4 per element)
        else:
            assert torch.equal(g1, g0), k
