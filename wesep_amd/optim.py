"""Per-tensor gradient clip + Adam (coupled L2) as two multi-tensor HIP launches.

Reference semantics reproduced exactly: `clip_gradients` clips EACH parameter tensor by its
own L2 norm with eps 1e-6 (wesep/utils/funcs.py:79-88, ~640 `.item()` host syncs per step in
the reference, zero here) and torch.optim.Adam(weight_decay=wd) adds wd*p to the gradient
(wesep/bin/train.py:237-238).  State keys (`step`, `exp_avg`, `exp_avg_sq`) match
torch.optim.Adam so optimizer checkpoints interchange (wesep/utils/checkpoint.py)."""
import math
import os

import numpy as np
import torch

from . import _lib as L
from . import dev


_TABLES = {}     # (device, tensors in the table, kind) -> (bytes, device tensor): the last table uploaded


def _table(refs, device):
    """Device table of (param, grad, exp_avg, exp_avg_sq, numel) rows for the multi-tensor launches.  The table of a step is
    almost always the table of the step before (the caching allocator hands the gradients the same blocks again), so the last one
    is kept and compared on the host: round 6 found the per-step upload -- a copy from pageable memory, i.e. a host wait for
    everything the stream still had queued -- draining the launch pipeline once per step (1.4 ms of idle GPU behind the last
    backward kernel + 1-2 ms of the next forward's launches arriving late: profiles/r06_bsrnn_trace_gaps.txt).  A table that did
    change goes through pinned memory, asynchronously."""
    arr = np.zeros(len(refs), dtype=L.TENSOR_REF_DTYPE)
    for i, (p, g, m, v) in enumerate(refs):
        arr[i] = (p.data_ptr(), g.data_ptr() if g is not None else 0,
                  m.data_ptr() if m is not None else 0, v.data_ptr() if v is not None else 0, p.numel())
    if os.environ.get("WESEP_TABLE_CACHE", "1") == "0":     # (A/B: the per-step blocking upload of rounds 1-5)
        return L.upload_struct_array(arr, device, blocking=True)
    key = (device.type, device.index, len(refs), refs[0][2] is None)
    raw = arr.tobytes()
    hit = _TABLES.get(key)
    if hit is not None and hit[0] == raw:
        return hit[1]
    tab = L.upload_struct_array(arr, device)
    _TABLES[key] = (raw, tab)
    return tab


def clip_gradients(model, clip):
    """Drop-in for `wesep.utils.funcs.clip_gradients`: clips in place, returns the list of
    per-parameter norms (one device->host copy for all of them).

    Non-finite gradients: since round 5 the scaled-fp16 d(gates) of the BPTT kernels are not clamped -- an
    overflow reaches the weight gradients as Inf / NaN by design, and FusedClipAdam skips such a step on the device.  This
    function serves every OTHER optimizer (torch.optim.*, the reference's own train.py), which has no skip: clip / (Inf + eps)
    = 0 and Inf * 0 = NaN would go straight into the weights.  So the norms carry the guard word here too: when any norm is
    not finite the clip launch is skipped on the device, and the host -- which reads the norms back anyway -- zeroes EVERY
    gradient of the step and warns: the following optimizer.step() sees a zero gradient instead of poison (Adam then only
    decays its moments; the weights stay finite), and `skipped_steps` on this function counts the events."""
    ps = [p for _, p in model.named_parameters() if p.grad is not None]
    if not ps:
        return []
    device = ps[0].device
    for p in ps:
        if not (p.is_cuda and p.grad.is_contiguous() and p.grad.dtype == torch.float32):
            raise L.WesepHipError("clip_gradients: fp32 contiguous CUDA gradients required")
    tab = _table([(p, p.grad, None, None) for p in ps], device)
    norms = torch.empty(len(ps), device=device, dtype=torch.float32)
    guard = torch.zeros(1, device=device, dtype=torch.int32)
    dev.grad_norms(tab, len(ps), norms, guard=guard)
    dev.poll_cluster_status(device)      # (an optimizer other than FusedClipAdam: the BPTT status word is looked at here)
    dev.clip_adam_step(tab, len(ps), norms, float(clip), 0.0, 0.9, 0.999, 1e-8, 0.0, 1, clip_only=True, skip=(guard, None))
    out = norms.tolist()
    if not all(math.isfinite(n) for n in out):
        import warnings
        clip_gradients.skipped_steps += 1
        for p in ps:
            p.grad.zero_()
        warnings.warn("clip_gradients: a gradient norm is not finite (NaN / Inf: an overflow of the scaled-fp16 d(gates) or a "
                      "diverged step); every gradient of this step was zeroed so that the optimizer cannot write NaN into the "
                      f"weights ({clip_gradients.skipped_steps} such steps so far)", RuntimeWarning)
    return out


clip_gradients.skipped_steps = 0


GUARD_POLL_EVERY = 16     # optimizer steps between two (blocking) looks at the guard words when there is more than one rank


def _world_size() -> int:
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


class FusedClipAdam(torch.optim.Optimizer):
    """Per-tensor clip + Adam-L2 in two multi-tensor launches, with a non-finite-gradient guard that never syncs the host.

    Guard (wesep_hip.h ws_grad_norms / ws_clip_adam_step / ws_guard_commit): four device words per device.  [0], zeroed
    every step and set by ws_grad_norms when a gradient norm is NaN / Inf, makes ws_clip_adam_step skip the WHOLE update of
    that step on the device.  ws_guard_commit then counts on the device: [1] skipped steps so far (exact), [2] CONSECUTIVE
    skipped steps, [3] the bias-correction lag.  The host reads the words asynchronously (a 16-byte copy whose event it
    queries at the next step):
      * `skipped_steps` is the device's exact count;
      * `state[p]["step"]` counts ATTEMPTED steps when `step()` returns; the kernels use step - lag for the bias
        corrections, so a skipped step never advances them (torch.optim.Adam under a GradScaler does not call step() on
        an overflow); whenever the host learns of n more skips it subtracts n from every state's `step` and from the
        device's lag word (stream-ordered, so every launch sees a consistent pair).  `state_dict()` reconciles first:
        checkpoints carry applied-step counts like torch.optim.Adam's;
      * `max_consecutive_skips` (default 10; WESEP_MAX_CONSECUTIVE_SKIPS): a run whose every step is non-finite has
        diverged -- the reference would show a NaN loss -- and training on nothing forever is worse than stopping:
        WesepHipError once the device has counted that many skipped steps in a row (noticed one poll late)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, clip_grad=0.0,
                 max_consecutive_skips=None):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, clip_grad=clip_grad)
        super().__init__(params, defaults)
        self._norms = None
        self._norm_params = None
        self._guard = {}
        self.skipped_steps = 0
        self.max_consecutive_skips = int(max_consecutive_skips if max_consecutive_skips is not None
                                         else os.environ.get("WESEP_MAX_CONSECUTIVE_SKIPS", "10"))
        self._calls = 0

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
        devs = {(l[1][0][0].device.type, l[1][0][0].device.index) for l in launches}
        for dkey in devs:
            self._guard_word(torch.device(*dkey))[0:1].zero_()   # this step's skip word
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
                               skip=(self._guard_word(device)[0:1], dev.bptt_status_word(device)),
                               step_lag=self._guard_word(device)[3:4])
        for dkey in devs:      # the step's book-keeping, on the device: skipped / consecutive / lag counters
            dv = torch.device(*dkey)
            dev.guard_commit(self._guard_word(dv), dev.bptt_status_word(dv))
        if cuda_dev is not None:
            # Under DistributedDataParallel every rank must reconcile (and, if it comes to that, raise) at the SAME step: an
            # event that "has completed" is a race between ranks.  With more than one rank the words are
            # looked at every GUARD_POLL_EVERY calls only, blocking (a 16-byte copy and one host sync per 16 steps); the skip
            # itself never waits for the host -- it happens on the device in the step it belongs to, on every rank alike.
            self._calls += 1
            if _world_size() > 1:
                if self._calls % GUARD_POLL_EVERY == 0:
                    self._poll_guard(cuda_dev, block=True)
            else:
                self._poll_guard(cuda_dev)
            dev.poll_cluster_status(cuda_dev)   # asynchronous: evaluates the copy started one step ago
            dev.step_fence(cuda_dev)            # the host stays at most one step ahead of the GPU (dev.StepFence)
        return loss

    def _guard_word(self, device):
        key = (device.type, device.index)
        if key not in self._guard:
            host = torch.zeros(4, dtype=torch.int32)
            if torch.cuda.is_available():
                host = host.pin_memory()
            self._guard[key] = dict(word=torch.zeros(4, device=device, dtype=torch.int32), host=host, event=None, seen=0)
        return self._guard[key]["word"]

    def _poll_guard(self, device, block=False):
        """Asynchronous look at the guard words: evaluates the 16-byte copy started by the previous call once its event has
        completed, then starts the next one (`block`: copy now and wait).  Newly skipped steps are counted in
        `skipped_steps` (reported once), taken off every state's `step` and off the device's lag word; raises WesepHipError
        when `max_consecutive_skips` steps in a row were skipped.  Returns `skipped_steps`."""
        g = self._guard.get((device.type, device.index))
        if g is None or not torch.cuda.is_available():
            return self.skipped_steps

        def evaluate():
            g["event"].synchronize()
            g["event"] = None
            total, consec, lag = int(g["host"][1]), int(g["host"][2]), int(g["host"][3])
            new = total - g["seen"]
            if new > 0:
                if self.skipped_steps == 0:
                    import warnings
                    warnings.warn("FusedClipAdam: a gradient was not finite (NaN / Inf); the optimizer step was skipped "
                                  "on the device and the weights are intact.  Further skipped steps are counted in "
                                  "`skipped_steps`", RuntimeWarning)
                g["seen"] = total
                self.skipped_steps += new
            if lag > 0:
                # the device has been correcting the bias terms by `lag` on its own; move that into the host's counts
                # (stream-ordered: launches enqueued before this line saw (step, lag), later ones see (step - lag, 0 + ...))
                for st in self.state.values():
                    if "step" in st:
                        st["step"] = max(int(st["step"]) - lag, 0)
                g["word"][3:4].sub_(lag)
            if self.max_consecutive_skips > 0 and consec >= self.max_consecutive_skips:
                raise L.WesepHipError(
                    f"FusedClipAdam: the last {consec} optimizer steps were all skipped on the device (non-finite gradients "
                    f"or BPTT time-outs in every step): the run has diverged or the GPU is shared; the weights are those of "
                    f"the last finite step.  max_consecutive_skips / WESEP_MAX_CONSECUTIVE_SKIPS raise the limit (0: never)")

        if g["event"] is not None and (block or g["event"].query()):
            evaluate()
        if g["event"] is None:
            g["host"].copy_(g["word"], non_blocking=True)
            g["event"] = torch.cuda.Event()
            g["event"].record()
            if block:
                evaluate()
        return self.skipped_steps

    def state_dict(self):
        """Reconciles the step counts with the device first (one host sync): checkpoints carry APPLIED steps."""
        for g in list(self._guard.values()):
            self._poll_guard(g["word"].device, block=True)
        return super().state_dict()

    def last_grad_norms(self):
        """Per-parameter gradient norms of the last step (syncs)."""
        if not self._norms:
            return []
        return torch.cat(self._norms).tolist()
