"""Loss registry with the reference's contract (`parse_loss`, wesep/utils/losses.py:8-41):
name -> instantiated callable `c(est, target) -> Tensor`.  SISDR/SISNR run on the HIP path;
CE is `CrossEntropyLoss` below (the class NAME is what the executor dispatches on,
executor.py:112-113), also on the HIP path.  Losses the hot path does not cover raise instead of silently falling
back."""
import torch.nn as nn

from .. import functional as F_


class SISDRLoss(nn.Module):
    """auraloss.time.SISDRLoss(zero_mean=True, eps=1e-8, reduction='mean') on gfx950."""

    def __init__(self, zero_mean=True, eps=1e-8, reduction="mean"):
        super().__init__()
        if not zero_mean or reduction != "mean":
            raise NotImplementedError("SISDRLoss kernel covers zero_mean=True, reduction='mean'")
        self.eps = eps

    def forward(self, input, target):
        return F_.SISDRFn.apply(input, target, self.eps)


class CrossEntropyLoss(nn.Module):
    """nn.CrossEntropyLoss() (mean reduction, class-index targets) on gfx950 (losses.py:11)."""

    def forward(self, input, target):
        from ..functional_tasnet import CrossEntropyFn
        return CrossEntropyFn.apply(input, target)


valid_losses = {
    "SISDR": SISDRLoss(),
    "SISNR": SISDRLoss(),
    "CE": CrossEntropyLoss(),
}


def parse_loss(loss):
    names = loss if isinstance(loss, list) else [loss]
    out = []
    for name in names:
        if name not in valid_losses:
            raise NotImplementedError(f"loss {name!r} is outside the built hot path (SISDR, SISNR, CE)")
        out.append(valid_losses[name])
    return out
