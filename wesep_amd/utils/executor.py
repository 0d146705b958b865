"""Train / validation step loop with the reference's `Executor` interface
(wesep/utils/executor.py:27-203) for the pBSRNN hot path on MI355X.

Same call signature, same loss composition (`se_loss_weight = (positions, weights)`), same
lr-before-step ordering and same return value `(mean_loss, 0)`.  Host-side differences, all
invisible to the caller: batches are pinned and copied one step ahead on a copy stream
(`utils/prefetch.DevicePrefetcher`); the per-step
`loss.item()` (executor.py:124) is replaced by a device-side running sum that is read only
when a log row is due and at the end of the epoch; per-tensor clipping + Adam are two
multi-tensor launches when the optimizer is `FusedClipAdam` (no per-parameter host syncs).
The SSA self-enrollment branch (executor.py:89-100: with probability `SSA_enroll_prob` a no-grad
pass estimates the target, its kaldi fbank + CMN replaces the enrollment, and the step runs on
that) computes the filterbank of the whole batch on the device (`utils/funcs.compute_fbank`,
SURVEY section 8f-2) instead of the reference's per-row host loop."""
import random
from contextlib import nullcontext

import torch

from ..optim import FusedClipAdam, clip_gradients
from .funcs import apply_cmvn, compute_fbank
from .prefetch import DevicePrefetcher


def _row(*cells):
    return " | ".join(f"{c:>10}" if not isinstance(c, float) else f"{c:10.4g}" for c in cells)


class ReplicaDivergence(RuntimeError):
    """Data-parallel replicas no longer hold the same parameters (Executor.train's self-check)."""


class Executor:
    def __init__(self, trace_losses=False):
        """trace_losses: keep every step's loss (device scalars, no host sync) in `self.loss_trace` -- tests and
        diagnosis; the reference interface is unchanged."""
        self.step = 0
        self.trace_losses = trace_losses
        self.loss_trace = []

    @staticmethod
    def _replica_check(model, device, where):
        """Every data-parallel replica must hold bit-identical parameters after an optimizer step (same averaged
        gradients, same update).  One small all-reduce pair per log row (wesep_amd.parallel.all_ranks_tensor_spread
        over a per-tensor checksum vector): a silent corruption on one rank -- a kernel fault, a bad collective -- is
        reported where it happens instead of surfacing as a diverged model hours later."""
        from ..parallel import all_ranks_tensor_spread
        with torch.no_grad():
            ps = [p for p in model.parameters()]
            sums = torch.stack([p.detach().double().sum() for p in ps] + [p.detach().double().abs().sum() for p in ps])
            # NaN compares unequal to itself: a non-finite checksum would read as "diverged" (or, through MAX / MIN, as
            # nothing at all).  It is its own finding: say so, with the flag all-reduced so that every rank raises
            finite = torch.isfinite(sums).all().to(torch.float64).reshape(1)
            sums = torch.nan_to_num(sums, nan=0.0, posinf=0.0, neginf=0.0)
        if all_ranks_tensor_spread(finite, device) != 0.0 or float(finite.item()) == 0.0:
            raise ReplicaDivergence(f"parameters are not finite on at least one data-parallel rank ({where})")
        spread = all_ranks_tensor_spread(sums, device)
        if spread != 0.0:
            raise ReplicaDivergence(f"data-parallel replicas diverged ({where}): parameter checksum spread {spread:.3e}")

    @staticmethod
    def _to_device(batch, device):
        mv = lambda t: t.float().to(device, non_blocking=True)
        spk = batch.get("spk_label")
        if isinstance(spk, torch.Tensor):
            spk = spk.to(device, non_blocking=True)
        return mv(batch["wav_mix"]), mv(batch["wav_targets"]), mv(batch["spk_embeds"]), spk

    @staticmethod
    def _loss(outputs, targets, spk_label, criterion, se_loss_weight, multi_task):
        if not isinstance(outputs, (list, tuple)):
            outputs = [outputs]
        if not isinstance(se_loss_weight, (list, tuple)):   # reference default 1.0: one loss on output 0
            se_loss_weight = ([[0]] * len(criterion), [[float(se_loss_weight)]] * len(criterion))
        positions, weights = se_loss_weight
        loss = 0
        for ii, crit in enumerate(criterion):
            is_ce = multi_task and crit.__class__.__name__ == "CrossEntropyLoss"
            for ji, pos in enumerate(positions[ii]):
                ref = spk_label if is_ce else targets
                loss = loss + weights[ii][ji] * crit(outputs[pos], ref).mean()
        return loss

    def train(self, dataloader, models, epoch_iter, optimizers, criterion, schedulers, scaler, epoch,
              enable_amp, logger, clip_grad=5.0, log_batch_interval=100, device=torch.device("cuda"),
              se_loss_weight=1.0, multi_task=False, SSA_enroll_prob=0, fbank_args=None,
              sample_rate=16000, speaker_feat=True, replica_check_interval=0):
        """Train one epoch.

        replica_check_interval (data-parallel runs; not in the reference's signature): > 0 checks every that many steps,
        INSIDE the loop, that all replicas still hold bit-identical, finite parameters -- a divergence is then reported
        within that many steps instead of at the end of an epoch (hours on the TF-GridNet recipe).  Only valid when every
        rank runs the same number of steps per epoch (the check is a collective of its own: under DistributedDataParallel's
        join() a rank that has run out of batches would not take part); bin/train.py turns it on for loaders it bounds by
        `epoch_iter` on every rank.  The end-of-epoch check below always runs, i.e. before any checkpoint is written."""
        if enable_amp and not getattr(self, "_amp_noted", False):
            # executor.py:88,130-134 wraps the step in torch.cuda.amp.autocast + GradScaler.  autocast rewrites the dtype of
            # ATen ops; this path has none to rewrite (every product is a split-bf16 MFMA with fp32 accumulation and fp32
            # storage -- at least the precision autocast's bf16 / fp16 GEMMs would have), and without fp16 gradients there
            # is nothing for a GradScaler to protect.  The switch is therefore accepted and changes nothing; a scaler
            # passed in is still honoured (scale -> unscale_ -> step -> update), so reference training scripts run as is.
            self._amp_noted = True
            if logger is not None:
                logger.info("enable_amp=True: the HIP path already multiplies in split-bf16 with fp32 accumulation; "
                            "autocast has nothing to downcast, the step is unchanged")
        model, optimizer, scheduler = models[0], optimizers[0], schedulers[0]
        model.train()
        ddp = isinstance(model, torch.nn.parallel.DistributedDataParallel)
        fused = isinstance(optimizer, FusedClipAdam)
        loss_sum = torch.zeros((), device=device)
        n_steps = 0
        with (model.join() if ddp else nullcontext()):
            # batch i+1 is pinned and copied on a side stream while step i computes (utils/prefetch.py)
            for i, batch in enumerate(DevicePrefetcher(dataloader, device)):
                cur_iter = (epoch - 1) * epoch_iter + i
                scheduler.step(cur_iter)
                features, targets, enroll, spk_label = self._to_device(batch, device)   # no-ops after the prefetch
                if SSA_enroll_prob > 0 and SSA_enroll_prob > random.random():
                    with torch.no_grad():
                        self_fbank = model(features, enroll)[0]
                        if fbank_args is not None and speaker_feat:
                            self_fbank = apply_cmvn(compute_fbank(self_fbank, **fbank_args, sample_rate=sample_rate))
                    enroll = self_fbank
                outputs = model(features, enroll)
                loss = self._loss(outputs, targets, spk_label, criterion, se_loss_weight, multi_task)
                loss_sum += loss.detach()
                if self.trace_losses:
                    self.loss_trace.append(loss.detach().clone())
                n_steps += 1
                optimizer.zero_grad()
                if scaler is not None:
                    scaler.scale(loss).backward()
                    scaler.unscale_(optimizer)
                else:
                    loss.backward()
                if fused:
                    for group in optimizer.param_groups:
                        group["clip_grad"] = clip_grad
                else:
                    clip_gradients(model, clip_grad)
                if scaler is not None:
                    scaler.step(optimizer)
                    scaler.update()
                else:
                    optimizer.step()
                self.step += 1
                if ddp and replica_check_interval > 0 and (i + 1) % replica_check_interval == 0 and (i + 1) < epoch_iter:
                    self._replica_check(model, device, f"epoch {epoch}, step {i + 1}")
                if (i + 1) % log_batch_interval == 0:
                    if logger is not None:
                        logger.info(_row("TRAIN", epoch, i + 1, float(loss_sum.item() / n_steps),
                                         float(optimizer.param_groups[0]["lr"])))
                if (i + 1) == epoch_iter:
                    break
        if ddp:
            # OUTSIDE the join context: Join shadows DistributedDataParallel's own collectives for ranks that ran out of
            # batches, not anybody else's -- a check issued from inside the loop would pair the active ranks' all-reduces
            # with nothing (uneven per-rank batch counts are what join() is there for).  Here every rank has left the
            # loop and Join's post-hook has synchronised the replicas to the last rank that stepped
            self._replica_check(model, device, f"end of epoch {epoch}")
        return float(loss_sum.item() / max(n_steps, 1)), 0

    def cv(self, dataloader, models, val_iter, criterion, epoch, enable_amp, logger,
           log_batch_interval=100, device=torch.device("cuda")):
        """Cross validation: criterion[0] on outputs[0] (executor.py:189)."""
        model = models[0]
        model.eval()
        loss_sum = torch.zeros((), device=device)
        n_steps = 0
        with torch.no_grad():
            for i, batch in enumerate(dataloader):
                features, targets, enroll, _ = self._to_device(batch, device)
                outputs = model(features, enroll)
                if not isinstance(outputs, (list, tuple)):
                    outputs = [outputs]
                loss_sum += criterion[0](outputs[0], targets).mean()
                n_steps += 1
                if (i + 1) % log_batch_interval == 0 and logger is not None:
                    logger.info(_row("VAL", epoch, i + 1, float(loss_sum.item() / n_steps), "-"))
                if (i + 1) == val_iter:
                    break
        return float(loss_sum.item() / max(n_steps, 1)), 0
