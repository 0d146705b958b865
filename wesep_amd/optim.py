"""Per-tensor gradient clip + Adam (coupled L2) as two multi-tensor HIP launches.

Reference semantics reproduced exactly: `clip_gradients` clips EACH parameter tensor by its
own L2 norm with eps 1e-6 (wesep/utils/funcs.py:79-88, ~640 `.item()` host syncs per step in
the reference, zero here) and torch.optim.Adam(weight_decay=wd) adds wd*p to the gradient
(wesep/bin/train.py:237-238).  State keys (`step`, `exp_avg`, `exp_avg_sq`) match
torch.optim.Adam so optimizer checkpoints interchange (wesep/utils/checkpoint.py)."""
import os

import numpy as np
import torch

from . import _lib as L
from . import dev


def _table(refs, device):
    arr = np.zeros(len(refs), dtype=L.TENSOR_REF_DTYPE)
    for i, (p, g, m, v) in enumerate(refs):
        arr[i] = (p.data_ptr(), g.data_ptr() if g is not None else 0,
                  m.data_ptr() if m is not None else 0, v.data_ptr() if v is not None else 0, p.numel())
    return L.upload_struct_array(arr, device)


def clip_gradients(model, clip):
    """Drop-in for `wesep.utils.funcs.clip_gradients`: clips in place, returns the list of
    per-parameter norms (one device->host copy for all of them)."""
    ps = [p for _, p in model.named_parameters() if p.grad is not None]
    if not ps:
        return []
    device = ps[0].device
    for p in ps:
        if not (p.is_cuda and p.grad.is_contiguous() and p.grad.dtype == torch.float32):
            raise L.WesepHipError("clip_gradients: fp32 contiguous CUDA gradients required")
    tab = _table([(p, p.grad, None, None) for p in ps], device)
    norms = torch.empty(len(ps), device=device, dtype=torch.float32)
    dev.grad_norms(tab, len(ps), norms)
    dev.poll_cluster_status(device)      # (an optimizer other than FusedClipAdam: the BPTT status word is looked at here)
    dev.clip_adam_step(tab, len(ps), norms, float(clip), 0.0, 0.9, 0.999, 1e-8, 0.0, 1, clip_only=True)
    return norms.tolist()


class FusedClipAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, clip_grad=0.0):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, clip_grad=clip_grad)
        super().__init__(params, defaults)
        self._norms = None
        self._norm_params = None
        # non-finite-gradient guard (wesep_hip.h ws_grad_norms / ws_clip_adam_step): two device words per device that
        # ws_grad_norms sets when a gradient holds NaN / Inf -- [0], zeroed every step, makes ws_clip_adam_step skip the
        # WHOLE update of that step on the device; [1], sticky, is looked at asynchronously (no host sync), counted, cleared
        self._guard = {}
        self.skipped_steps = 0

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self._norms, self._norm_params = [], []
        cuda_dev = next((p.device for g in self.param_groups for p in g["params"] if p.is_cuda), None)
        if cuda_dev is not None and os.environ.get("WESEP_LSTM_CLUSTER_BWD", "0") == "1":
            # the opt-in in-place cluster BPTT cannot be repaired on the device: look at its status word BEFORE the
            # update (one host sync per step in this mode); raises WesepHipError on a timeout
            dev.poll_cluster_status(cuda_dev, block=True)
        launches = []     # (table, refs, step, group, norms): every (group, step-count) bucket of this call
        for group in self.param_groups:
            refs = []
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda:
                    raise L.WesepHipError("FusedClipAdam: parameters must live on the GPU (no CPU path)")
                if not p.grad.is_contiguous():
                    p.grad = p.grad.contiguous()
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] = int(st["step"]) + 1
                refs.append((p, p.grad, st["exp_avg"], st["exp_avg_sq"]))
            if not refs:
                continue
            # torch.optim.Adam keeps `step` per parameter: parameters that joined later (spk_model_freeze lifted
            # after a resume, a head that only sometimes receives gradients) have their own bias corrections ->
            # one launch per distinct step count (one in the usual case)
            buckets = {}
            for r in refs:
                buckets.setdefault(int(self.state[r[0]]["step"]), []).append(r)
            for step, brefs in sorted(buckets.items()):
                launches.append((_table(brefs, brefs[0][0].device), brefs, step, group))
        # pass 1: every bucket's norms -- they carry the non-finite guard, so they are taken also without clipping (one
        # read of the gradients); pass 2: the updates, ALL skipped on the device when ANY gradient of the step was not
        # finite (a BPTT time-out poisons with NaN; under DDP the all-reduce has spread it to every rank, so every rank
        # skips the same step) or when an in-place BPTT launch of this stream reported a time-out
        for dkey in {(l[1][0][0].device.type, l[1][0][0].device.index) for l in launches}:
            self._guard_word(torch.device(*dkey))[0:1].zero_()   # this step's skip word; [1] is the sticky copy
        normed = []
        for tab, brefs, step, group in launches:
            device = brefs[0][0].device
            norms = torch.empty(len(brefs), device=device, dtype=torch.float32)
            dev.grad_norms(tab, len(brefs), norms, guard=self._guard_word(device))
            if float(group["clip_grad"]) > 0:
                self._norms.append(norms)
                self._norm_params += [r[0] for r in brefs]
            normed.append(norms)
        for (tab, brefs, step, group), norms in zip(launches, normed):
            device = brefs[0][0].device
            clip = float(group["clip_grad"])
            b1, b2 = group["betas"]
            dev.clip_adam_step(tab, len(brefs), norms if clip > 0 else None, clip, float(group["lr"]), b1, b2,
                               group["eps"], group["weight_decay"], step,
                               skip=(self._guard_word(device)[0:1], dev.bptt_status_word(device)))
        if cuda_dev is not None:
            self._poll_guard(cuda_dev)
            dev.poll_cluster_status(cuda_dev)   # asynchronous: evaluates the copy started one step ago
        return loss

    def _guard_word(self, device):
        key = (device.type, device.index)
        if key not in self._guard:
            host = torch.zeros(1, dtype=torch.int32)
            if torch.cuda.is_available():
                host = host.pin_memory()
            self._guard[key] = dict(word=torch.zeros(2, device=device, dtype=torch.int32), host=host, event=None)
        return self._guard[key]["word"]

    def _poll_guard(self, device, block=False):
        """Asynchronous look at the guard word: evaluates the 4-byte copy started by the previous call once its event has
        completed, then starts the next one (`block`: copy now and wait).  A set word = the update of (at least) one step
        was skipped on the device: counted in `skipped_steps`, reported once, cleared -- training goes on with intact
        weights.  Returns `skipped_steps`."""
        g = self._guard.get((device.type, device.index))
        if g is None or not torch.cuda.is_available():
            return self.skipped_steps

        def evaluate():
            g["event"].synchronize()
            g["event"] = None
            if int(g["host"][0]):
                g["word"][1:2].zero_()
                if self.skipped_steps == 0:
                    import warnings
                    warnings.warn("FusedClipAdam: a gradient was not finite (NaN / Inf); the optimizer step was skipped "
                                  "on the device and the weights are intact.  Further skipped steps are counted in "
                                  "`skipped_steps`", RuntimeWarning)
                self.skipped_steps += 1

        if g["event"] is not None and (block or g["event"].query()):
            evaluate()
        if g["event"] is None:
            g["host"].copy_(g["word"][1:2], non_blocking=True)
            g["event"] = torch.cuda.Event()
            g["event"].record()
            if block:
                evaluate()
        return self.skipped_steps

    def last_grad_norms(self):
        """Per-parameter gradient norms of the last step (syncs)."""
        if not self._norms:
            return []
        return torch.cat(self._norms).tolist()
