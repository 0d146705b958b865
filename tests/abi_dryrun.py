"""TEST INFRASTRUCTURE.  Argument-contract dry run of the C ABI on a machine without a GPU.

Every `ws_*` entry point validates its arguments (WS_REQUIRE -> WS_ERR_INVALID, rc -1) BEFORE it launches; without a
device the launch itself then fails with rc -2 ("no ROCm-capable device").  `install()` lets host code run on CPU
tensors straight into the real libwesep_hip.so and records (entry point, rc, message) per call, so a test can assert
that a whole host path -- shapes, leading dimensions, alignment flags, split counts -- passes every entry point's
contract, i.e. would reach the launch on the GPU.  Nothing is computed (outputs stay uninitialised): numerics are
covered by tests/emu_dev.py on CPU and by the `-m gpu` tests."""
import ctypes as C

import torch

WS_ERR_INVALID = -1


def install(monkeypatch):
    import wesep_amd._lib as L
    import wesep_amd.functional as f0
    import wesep_amd.functional_dpccn as fd
    import wesep_amd.functional_resnet as fr
    import wesep_amd.functional_tasnet as ft
    import wesep_amd.functional_tfgridnet as fg
    import wesep_amd.functional_campplus as fc
    import wesep_amd.functional_ecapa as fe
    calls = []

    def check(rc, what=""):
        calls.append((what, rc, L.lib().ws_last_error().decode("utf-8", "replace") if rc else ""))

    monkeypatch.setattr(L, "check", check)
    monkeypatch.setattr(L, "stream_ptr", lambda: C.c_void_p(0))
    monkeypatch.setattr(torch.Tensor, "is_cuda", property(lambda self: True))
    import wesep_amd.dev as dev
    monkeypatch.setattr(dev, "cu_count", lambda device: 256)          # MI355X
    monkeypatch.setenv("WESEP_WGRAD_OVERLAP", "0")                    # no side streams without a device
    for mod in (f0, fd, ft, fg, fr, fc, fe):
        monkeypatch.setattr(mod, "_need_cuda", lambda t, who: None)
    return calls


def assert_contracts_hold(calls, at_least=1):
    assert len(calls) >= at_least, f"only {len(calls)} entry-point calls recorded"
    bad = [(w, m) for w, rc, m in calls if rc == WS_ERR_INVALID]
    assert not bad, bad[:5]
    assert all(rc != 0 for _, rc, _ in calls), "a launch succeeded: this harness is for GPU-less machines"
