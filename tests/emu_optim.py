"""TEST INFRASTRUCTURE.  CPU stand-ins for the two multi-tensor optimizer launches (elementwise.hip: ws_grad_norms,
ws_clip_adam_step) so that wesep_amd.optim.FusedClipAdam's HOST logic -- buckets by step count, the two-pass step, the
non-finite-gradient guard words, the skip of the whole update, the asynchronous count of skipped steps -- runs without a GPU.
The arithmetic below restates the kernels line by line (per-tensor clip coefficient clip / (norm + 1e-6) when < 1, coupled
L2, Adam with bias corrections); the "device table" is the list of tensor tuples itself.  Nothing outside tests/ imports this."""
import math

import torch


def grad_norms(tab, ntensors, norms, guard=None):
    bad = False
    for i, (p, g, m, v) in enumerate(tab[:ntensors]):
        n = g.detach().float().norm()
        norms[i] = n
        bad = bad or not bool(torch.isfinite(n))
    if guard is not None and bad:          # [0]: this step's skip word
        guard[0] = 1


def guard_commit(guard, skip1=None):
    """ws_guard_commit: [1] skipped steps, [2] consecutive skipped steps, [3] bias-correction lag."""
    if int(guard[0]) != 0 or (skip1 is not None and int(skip1.reshape(-1)[0]) != 0):
        guard[1] += 1
        guard[2] += 1
        guard[3] += 1
    else:
        guard[2] = 0


def clip_adam_step(tab, ntensors, norms, clip, lr, beta1, beta2, eps, weight_decay, step, clip_only=False, skip=(None, None),
                   step_lag=None):
    if any(w is not None and int(w.reshape(-1)[0]) != 0 for w in skip):
        return                               # the whole launch does nothing (clip_adam_kernel's first line)
    if step_lag is not None:
        step = max(step - int(step_lag.reshape(-1)[0]), 1)
    bc1 = 1.0 - beta1 ** step if not clip_only else 1.0
    bc2_sqrt = math.sqrt(1.0 - beta2 ** step) if not clip_only else 1.0
    for i, (p, g, m, v) in enumerate(tab[:ntensors]):
        if g is None:
            continue
        coef = 1.0
        if clip > 0:
            c = clip / (float(norms[i]) + 1e-6)
            if c < 1.0:
                coef = c
        with torch.no_grad():
            if coef != 1.0:
                g.mul_(coef)
            if clip_only:
                continue
            gg = g + weight_decay * p
            m.mul_(beta1).add_(gg, alpha=1.0 - beta1)
            v.mul_(beta2).addcmul_(gg, gg, value=1.0 - beta2)
            p.sub_((lr / bc1) * (m / (v.sqrt() / bc2_sqrt + eps)))


def install(monkeypatch):
    """FusedClipAdam on CPU tensors: the table is the tuple list, the launches are the functions above, streams / events /
    pinned memory are stand-ins (tests/emu_streams.py)."""
    from tests import emu_streams
    from wesep_amd import dev, optim
    emu_streams.install(monkeypatch)
    monkeypatch.setattr(optim, "_table", lambda refs, device: list(refs))
    monkeypatch.setattr(dev, "grad_norms", grad_norms)
    monkeypatch.setattr(dev, "clip_adam_step", clip_adam_step)
    monkeypatch.setattr(dev, "guard_commit", guard_commit)
    monkeypatch.setattr(dev, "bptt_status_word", lambda device: None)
    monkeypatch.setattr(dev, "poll_cluster_status", lambda device, block=False: 0)
    monkeypatch.setattr(torch.Tensor, "is_cuda", property(lambda self: True))
    monkeypatch.setattr(torch.Tensor, "pin_memory", lambda self, *a, **k: self)
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
