"""TEST INFRASTRUCTURE.  Stand-ins for torch.cuda streams / events on a machine without a GPU, so that the product's
side-stream machinery (functional.WGradCarrierFn, the deferred weight-gradient jobs of ResRNNBlkFn, flush / mark-ready)
can run on the CPU emulation of the device entry points (tests/emu_dev.py, tests/emu_blk.py): a "stream" executes at
once, an "event" is always complete.  The ORDER of operations -- which job is released where, which carrier delivers
which gradients to autograd and when DistributedDataParallel's hooks therefore fire -- is the product's own; only the
asynchrony is gone.  Nothing outside tests/ imports this."""
import contextlib

import torch


class FakeStream:
    cuda_stream = 0

    def __init__(self, device=None, priority=0):
        self.device = device

    def wait_event(self, event):
        pass

    def wait_stream(self, stream):
        pass

    def synchronize(self):
        pass

    def record_event(self, event=None):
        return event or FakeEvent()


class FakeEvent:
    def __init__(self, enable_timing=False):
        pass

    def record(self, stream=None):
        pass

    def wait(self, stream=None):
        pass

    def query(self):
        return True

    def synchronize(self):
        pass


def install(monkeypatch):
    monkeypatch.setattr(torch.cuda, "Stream", FakeStream)
    monkeypatch.setattr(torch.cuda, "Event", FakeEvent)
    monkeypatch.setattr(torch.cuda, "current_stream", lambda device=None: FakeStream())
    monkeypatch.setattr(torch.cuda, "stream", lambda s: contextlib.nullcontext())
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    monkeypatch.setattr(torch.Tensor, "record_stream", lambda self, s: None, raising=False)
