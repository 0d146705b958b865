"""Diagnostic (GPU): run-to-run identity of the cluster recurrence while other streams keep the chip busy.

Two streams each loop the forward cluster kernel on their own buffers (as two inference engines on one GPU do), a
third one runs GEMMs.  Every output is compared bit for bit with the first one of its stream.  `legacy` = round 1's
block -> cluster mapping (dbg bit 16), under which the members of small launches sit on different XCDs.
Usage: python tools/cluster_stress.py [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wesep_amd import dev  # noqa: E402
from wesep_amd.dev import BIG, SeqMap  # noqa: E402

d = torch.device("cuda:0")
H = 256
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 150


def setup(nseq, L, seed):
    g = torch.Generator().manual_seed(seed)
    seq = SeqMap(nseq, BIG, 0, L, 1, L)
    nb = dev.bl_num_blocks(seq)
    pre = (0.5 * torch.randn(nb, 32 * 8 * H, generator=g)).to(d)
    whf, whr = ((0.06 * torch.randn(4 * H, H, generator=g)).to(d) for _ in range(2))
    return seq, nb, pre, whf, whr


def run(nseq, L, legacy, noise):
    streams = [torch.cuda.Stream(device=d) for _ in range(2)]
    noise_s = torch.cuda.Stream(device=d)
    jobs = [setup(nseq, L, 10 + i) for i in range(2)]
    status = torch.zeros(1, device=d, dtype=torch.int32)
    first = [None, None]
    bad = [0, 0]
    a = torch.randn(4096, 4096, device=d)
    torch.cuda.synchronize()
    outs = [[], []]
    for it in range(iters):
        if noise:
            with torch.cuda.stream(noise_s):
                for _ in range(2):
                    a @ a
        for i, (seq, nb, pre, whf, whr) in enumerate(jobs):
            with torch.cuda.stream(streams[i]):
                gates = pre.clone()
                cbuf, hcat = torch.empty(nb, 32 * 2 * H, device=d), torch.empty(nb, 32 * 2 * H, device=d)
                dev.lstm_fwd_cluster(gates, cbuf, hcat, whf, whr, seq, status=status, dbg=16 if legacy else 0)
                outs[i].append(hcat)
        if it % 25 == 24 or it == iters - 1:
            torch.cuda.synchronize()
            for i in range(2):
                for h in outs[i]:
                    if first[i] is None:
                        first[i] = h
                    elif not torch.equal(h, first[i]):
                        bad[i] += 1
                outs[i] = []
    torch.cuda.synchronize()
    print(f"nseq {nseq} L {L} mapping {'legacy' if legacy else 'xcd-local'} noise {int(noise)}: mismatching outputs "
          f"{bad[0]} + {bad[1]} of {iters} each, status {int(status.item())}", flush=True)


for nseq in (64, 128):
    for legacy in (True, False):
        for noise in (False, True):
            run(nseq, 188, legacy, noise)
