"""Band-view ResRNN at a ragged geometry (2505 sequences: a partly filled last workgroup) with torch.empty() NaN-filled
(torch.utils.deterministic.fill_uninitialized_memory): every output and gradient must equal the plain run bit for bit.
Both recurrence selections (two-kernel path, fused input projection).  Prints one line per mode."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(blk, z0, gout):
    z = z0.clone().requires_grad_(True)
    for p_ in blk.parameters():
        p_.grad = None
    out = blk(z, "band")
    out.backward(gout)
    torch.cuda.synchronize()
    return [out.detach().clone(), z.grad.detach().clone()] + [v.grad.detach().clone() for v in blk.parameters()]


def main():
    from wesep_amd.models.bsrnn import ResRNN
    d = torch.device("cuda:0")
    torch.manual_seed(11)
    R, K, Tf, N = 5, 32, 501, 128
    blk = ResRNN(N, 2 * N).to(d)
    z0 = torch.randn(R, K, Tf, N, device=d)
    gout = torch.randn(R, K, Tf, N, device=d)
    names = ["out", "dz"] + [k for k, _ in blk.named_parameters()]
    for view_T in ((R, K, Tf),):
        for fuse in ("0", "1"):
            os.environ["WESEP_LSTM_FUSE"] = fuse
            plain = run(blk, z0, gout)
            # poison the caching allocator's free blocks too: allocate and free a large NaN buffer
            junk = torch.full((1 << 28,), float("nan"), device=d)
            del junk
            again = run(blk, z0, gout)
            torch.use_deterministic_algorithms(True, warn_only=True)
            torch.utils.deterministic.fill_uninitialized_memory = True
            try:
                filled = run(blk, z0, gout)
            finally:
                torch.utils.deterministic.fill_uninitialized_memory = False
                torch.use_deterministic_algorithms(False)
            bad1 = [n for n, a, b in zip(names, plain, again) if not torch.equal(a, b)]
            bad2 = [n for n, a, b in zip(names, plain, filled) if not torch.equal(a, b)]
            nan = [n for n, a in zip(names, filled) if not torch.isfinite(a).all()]
            print(f"fuse={fuse}: after NaN-poisoned free blocks differ: {bad1}; NaN-filled empty() differ: {bad2}; non-finite: {nan}")


if __name__ == "__main__":
    main()
