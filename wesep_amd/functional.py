"""Autograd boundary of the pBSRNN hot path: torch.autograd.Function shims whose forward and
backward are sequences of C-ABI launches (wesep_amd/dev.py).  PyTorch supplies device memory,
the stream, and the autograd graph between these coarse ops (so DistributedDataParallel's
bucket hooks fire on the registered Parameters); every FLOP runs in libwesep_hip.so.

Activation layout everywhere: Z = [R, K, Tf, N] fp32 (see include/wesep_hip.h)."""
import math
import os

import numpy as np
import torch

from . import _lib as L
from . import dev
from .dev import BIG, Geom, Rows, SeqMap, StatMap, flat

H = L.LSTM_H          # LSTM hidden size the recurrent kernels are built for
G4 = 4 * H
NBIN = 257
HOP = 128


def _empty(dev_, *shape):
    return torch.empty(*shape, device=dev_, dtype=torch.float32)


def _need_cuda(t, who):
    if not t.is_cuda:
        raise L.WesepHipError(f"{who}: wesep_amd has no CPU path; move the model and inputs to the GPU")


def _reduce_new(slab, nsplit, stride, shape):
    out = _empty(slab.device, *shape)
    dev.reduce_slabs(slab, nsplit, stride, int(np.prod(shape)), out)
    return out


# ---------------------------------------------------------------------------------------------
# ResRNN: GroupNorm -> BLSTM -> Linear -> +residual   (wesep/models/bsrnn.py:38-46)
# ---------------------------------------------------------------------------------------------
def _view_maps(view, R, K, Tf, N):
    if view == "time":      # band_rnn: sequences (r,k), steps over t  (bsrnn.py:73-75)
        geo = Geom(R * K, 1, Tf * N, 0, N, Tf, N)
        smap = StatMap(Tf, 1, 1, 0, 0)
        seq = SeqMap(R * K, BIG, 0, Tf, 1, Tf)
        shift = (1, Tf)                     # (seq_div, seq_len) of the h_{t-1} row shift
    elif view == "band":    # band_comm: sequences (r,t), steps over k  (bsrnn.py:78-81)
        geo = Geom(R * Tf, Tf, K * Tf * N, N, Tf * N, K, N)
        smap = StatMap(K * Tf, Tf, Tf, 1, 0)
        seq = SeqMap(R * Tf, Tf, K * Tf, 1, Tf, K)
        shift = (Tf, K)
    else:
        raise ValueError(view)
    return geo, smap, seq, shift


class ResRNNFn(torch.autograd.Function):
    """inputs: z, view, norm.weight, norm.bias, weight_ih_l0, weight_hh_l0, bias_ih_l0, bias_hh_l0,
    the four *_reverse tensors, proj.weight, proj.bias."""

    @staticmethod
    def forward(ctx, z, view, norm_w, norm_b, wih_f, whh_f, bih_f, bhh_f, wih_r, whh_r, bih_r,
                bhh_r, proj_w, proj_b):
        _need_cuda(z, "ResRNN")
        z = z.contiguous()
        R, K, Tf, N = z.shape
        if tuple(whh_f.shape) != (G4, H) or tuple(wih_f.shape) != (G4, N):
            raise L.WesepHipError(f"ResRNN kernels are built for hidden {H}; got {tuple(whh_f.shape)}")
        P = R * K * Tf
        d = z.device
        geo, smap, seq, _ = _view_maps(view, R, K, Tf, N)
        stats = _empty(d, geo.ngroups, 2)
        dev.group_stats(z, geo, stats)
        wcat, bcat = _empty(d, 2 * G4, N), _empty(d, 2 * G4)
        dev.lstm_cat_ih(wih_f.contiguous(), wih_r.contiguous(), bih_f, bhh_f, bih_r, bhh_r, N, wcat, bcat)
        pack_f, pack_b = _empty(d, L.LSTM_PACK_FLOATS), _empty(d, L.LSTM_PACK_FLOATS)
        mt = dev.lstm_mode(seq.nseq)
        dev.lstm_pack(whh_f.contiguous(), whh_r.contiguous(), pack_f, pack_b, mt)
        gates = _empty(d, P, 2 * G4)
        dev.gemm_nt(A=z, a_rows=flat(N), M=P, N=2 * G4, K=N, W=wcat, ldw=N, bias=bcat, C_out=gates,
                    c_rows=flat(2 * G4), stats=stats, gamma=norm_w, beta=norm_b, stat_map=smap)
        cbuf, hcat = _empty(d, P, 2 * H), _empty(d, P, 2 * H)
        dev.lstm_fwd(gates, cbuf, hcat, pack_f, seq, mt)
        out = torch.empty_like(z)
        pw = proj_w.contiguous()
        dev.gemm_nt(A=hcat, a_rows=flat(2 * H), M=P, N=N, K=2 * H, W=pw, ldw=2 * H, bias=proj_b,
                    R=z, C_out=out, c_rows=flat(N))
        ctx.save_for_backward(z, stats, gates, cbuf, hcat, wcat, pack_b, norm_w, norm_b, pw)
        ctx.view, ctx.mt = view, mt
        return out

    @staticmethod
    def backward(ctx, dout):
        z, stats, gates, cbuf, hcat, wcat, pack_b, norm_w, norm_b, pw = ctx.saved_tensors
        dout = dout.contiguous()
        R, K, Tf, N = z.shape
        P = R * K * Tf
        d = z.device
        geo, smap, seq, (seq_div, seq_len) = _view_maps(ctx.view, R, K, Tf, N)
        nsplit, rps = dev.tn_splits(P)
        # projection: data + weight gradients
        projT = _empty(d, 2 * H, N)
        dev.transpose(pw, N, 2 * H, 2 * H, projT)
        dhcat = _empty(d, P, 2 * H)
        dev.gemm_nt(A=dout, a_rows=flat(N), M=P, N=2 * H, K=N, W=projT, ldw=N, C_out=dhcat,
                    c_rows=flat(2 * H))
        slab, bslab = _empty(d, nsplit, N * 2 * H), _empty(d, nsplit, N)
        dev.gemm_tn(G=dout, g_rows=flat(N), A=hcat, a_rows=flat(2 * H), M=P, Nn=N, Kk=2 * H, slab=slab,
                    slab_stride=N * 2 * H, bslab=bslab, bslab_stride=N, nsplit=nsplit, rows_per_split=rps)
        dproj_w = _reduce_new(slab, nsplit, N * 2 * H, (N, 2 * H))
        dproj_b = _reduce_new(bslab, nsplit, N, (N,))
        # BPTT: gates (activated) -> d(pre-activation gates), in place
        dev.lstm_bwd(gates, cbuf, hcat, dhcat, pack_b, seq, ctx.mt)
        del dhcat
        # recurrent weight gradients: dW_hh[d] = dgates_d^T h_{t-1}
        dwhh = []
        slab = _empty(d, nsplit, G4 * H)
        for di in (0, 1):
            dev.gemm_tn(G=gates, g_rows=flat(2 * G4), g_off=di * G4, A=hcat, a_rows=flat(2 * H),
                        a_off=di * H, M=P, Nn=G4, Kk=H, slab=slab, slab_stride=G4 * H, nsplit=nsplit,
                        rows_per_split=rps, shift_rows=(-seq.step_rows if di == 0 else seq.step_rows),
                        seq_div=seq_div, seq_len=seq_len)
            dwhh.append(_reduce_new(slab, nsplit, G4 * H, (G4, H)))
        # input weight / bias gradients against the re-normalised input
        slab, bslab = _empty(d, nsplit, 2 * G4 * N), _empty(d, nsplit, 2 * G4)
        dev.gemm_tn(G=gates, g_rows=flat(2 * G4), A=z, a_rows=flat(N), M=P, Nn=2 * G4, Kk=N, slab=slab,
                    slab_stride=2 * G4 * N, bslab=bslab, bslab_stride=2 * G4, nsplit=nsplit,
                    rows_per_split=rps, stats=stats, gamma=norm_w, beta=norm_b, stat_map=smap)
        dwcat = _reduce_new(slab, nsplit, 2 * G4 * N, (2 * G4, N))
        dbcat = _reduce_new(bslab, nsplit, 2 * G4, (2 * G4,))
        del slab, bslab
        # d(normalised input) -> GroupNorm backward (+ residual path)
        wcatT = _empty(d, N, 2 * G4)
        dev.transpose(wcat, 2 * G4, N, N, wcatT)
        dxn = _empty(d, P, N)
        dev.gemm_nt(A=gates, a_rows=flat(2 * G4), M=P, N=N, K=2 * G4, W=wcatT, ldw=2 * G4, C_out=dxn,
                    c_rows=flat(N))
        ab = _empty(d, geo.ngroups, 2)
        dev.gn_bwd_reduce(z, dxn, stats, geo, ab, gamma=norm_w)
        ns2 = min(256, geo.ngroups)
        pslab = _empty(d, ns2, 2, N)
        dev.gn_param_grad(z, dxn, stats, geo, ns2, pslab)
        dgb = _reduce_new(pslab, ns2, 2 * N, (2, N))
        dz = torch.empty_like(z)
        dev.gn_bwd_apply(z, dxn, stats, ab, geo, dz, gamma=norm_w, res=dout)
        # b_ih and b_hh receive the same gradient; clone so their .grad never alias
        return (dz, None, dgb[0], dgb[1],
                dwcat[:G4], dwhh[0], dbcat[:G4], dbcat[:G4].clone(),
                dwcat[G4:], dwhh[1], dbcat[G4:], dbcat[G4:].clone(),
                dproj_w, dproj_b)


def resrnn_mode() -> str:
    """'blocked' (default): split-bf16 path on the blocked layout BL (gemm_blk.hip, lstm_bf16.hip);
    'plain': the generic row-addressed GEMMs + plain-layout recurrence (honours WESEP_GEMM /
    WESEP_LSTM, e.g. both f32 for the exact-fp32 reference path)."""
    return os.environ.get("WESEP_RESRNN", "blocked")


def _h2_probe() -> int:
    """NUMERICS PROBE (tools/r04_h2_numerics.py; off by default): emulate narrower storage of the saved recurrence state
    by rounding the fp32 buffers in place between kernels.  Bits: 1 = activated gates to fp16, 2 = d(gates) to bf16
    (the hi term of the split pair only), 4 = cell state to fp16, 8 = activated gates to unorm16, 16 = d(hcat) to bf16,
    64 / 128 = the A operand [xn | h] of the weight-gradient GEMMs to fp16 / bf16, 256 / 512 = the pre-activations of the
    unfused (time-view) forward to fp16 / bf16, 1024 = the proj weight gradient on fp16 operands (CPU emulation)."""
    return int(os.environ.get("WESEP_H2_PROBE", "0"))


_PROBE_SAT = [0, 0.0]     # (probe bit 32) saturated d(gates) elements, largest |scaled d(gates)| / 65504 seen


def _probe_round(t, kind, packed=False):
    """In-place rounding of an fp32 buffer.  packed: the buffer holds BLS pairs on the device (hi << 16 | lo): keeping the
    hi term only IS bf16(x); on the CPU emulation it holds plain fp32."""
    if kind == "f16":
        t.copy_(t.half().float())
    elif kind == "u16":      # BL(2048) activated gates: quad q = column >> 2, gate = (q >> 6) & 3; i, f, o in (0, 1), g in (-1, 1)
        v = t.view(-1, 2, 4, 64 * 128)
        v[:, :, 2].mul_(0.5).add_(0.5)
        v.copy_(torch.floor(v * 65535.0 + 0.5) / 65535.0)
        v[:, :, 2].mul_(2.0).sub_(1.0)
    elif packed and torch.cuda.is_available():
        t.view(torch.int32).bitwise_and_(-65536)
    else:
        t.copy_(t.bfloat16().float())


def tnb_a16() -> bool:
    """fp16 copies of [xn | h] for the weight-gradient GEMMs (ws_gemm_tnb a_fmt = 1, ABI v16; with the default WS_GATES_H2F
    only).  WESEP_TNB_A16=0 keeps the split-pair A operand of round 3."""
    return os.environ.get("WESEP_TNB_A16", "1") != "0"


def pair_rfmt(gfmt) -> int:
    """Arithmetic of the pair BPTT's recurrent product (ws_lstm_pair_args.rfmt): 3 (default with WS_GATES_H2F, ABI v20) = the
    stored scaled-fp16 d(gates) x W_hh as fp16 hi + block-scaled FP8 lo, all of W_hh resident on the compute unit, the lo term on
    the block-scaled FP8 matrix instruction (K = 64 at twice the fp16 rate; 0.4 ms per step: profiles/r06_ab/r06_c23_*);
    WESEP_PAIR_RF=2: the same weights, both terms on the fp16 MFMA (ABI v18, the default of round 5); 1: fp16 hi / lo, the lo
    plane streamed (ABI v17); 0: the three-term split-bf16 product of rounds 3-4."""
    rf = int(os.environ.get("WESEP_PAIR_RF", "3"))
    if rf not in (0, 1, 2, 3):
        raise ValueError(f"WESEP_PAIR_RF={rf}: 0, 1, 2 or 3")
    return rf if gfmt == L.GATES_H2F else 0


def dxn_fmt(g_fmt) -> int:
    """a_fmt of the d(xn) GEMM over the 2-byte d(gates) (ws_gemm_b2p): with scaled-fp16 d(gates) (g_fmt 2) 3 = the lo term of the
    product on the block-scaled FP8 matrix instruction (ABI v20, the default: 0.57 -> 0.53 ms per launch alone, 1 ms per step --
    profiles/r06_c30_band_probe.txt, r06_ab/r06_c30_*); WESEP_DXN_F8=0: the format itself (2: both terms on the fp16 MFMA)."""
    return 3 if g_fmt == 2 and os.environ.get("WESEP_DXN_F8", "1") != "0" else g_fmt


def _wiht_kind(g_fmt) -> str:
    return {0: "wihT", 1: "wihT", 2: "wihT16", 3: "wihT8"}[dxn_fmt(g_fmt)]


def band_rfmt(gfmt, lmode) -> int:
    """Arithmetic of the streaming BPTT's recurrent product (ws_lstm_args.rfmt): 2 (ABI v18, with WS_GATES_H2F on the 32-sequence
    blocked kernels) = the stored scaled-fp16 d(gates) x W_hh as fp16 hi + scaled-FP8 lo, two MFMAs per product and three
    quarters of the weight stream -- the pair BPTT's arithmetic (pair_rfmt); config 2's parity and the 60-step trajectory with
    it: profiles/r06_c1_parity_brf2.log, r05_c23_band_rf2_trajectory.log.  3 (ABI v20, the default): the same pack with the lo
    term on the block-scaled FP8 matrix instruction (2.07 -> 1.88 ms per launch alone, 0.6 ms per step:
    profiles/r06_c26_band_probe.txt, r06_ab/r06_c26_*).  WESEP_BAND_RF=0: the three-term split-bf16 product of rounds 1-5."""
    rf = int(os.environ.get("WESEP_BAND_RF", "3"))
    if rf not in (0, 2, 3):
        raise ValueError(f"WESEP_BAND_RF={rf}: 0, 2 or 3")
    if rf == 3 and os.environ.get("WESEP_BAND_DX", "0") == "1":
        rf = 2                                          # (d(xn) inside the BPTT rides on the fp16 lo term's fragments)
    return rf if gfmt == L.GATES_H2F and lmode == L.LSTM_BF16X3_BLK else 0


def band_dx(brf, seq, geo) -> bool:
    """d(xn) = d(gates) W_ih computed INSIDE the streaming BPTT (ws_lstm_args.dxn, ABI v19; OPT-IN: WESEP_BAND_DX=1, with
    band_rfmt 2): the kernel holds d(gates) in LDS when it produces them, so ws_gemm_b2p's second pass over that 2.1 GB buffer
    (per band-view layer at R = 32) and its launch disappear; the fused GroupNorm backward adds the two directions' shares.
    Measured (profiles/r06_c4_band_probe.txt, r06_ab/r06_c4_bench_{new,nodx}.json): correct to 4.9e-6 and 12.6 GB of HBM reads
    per step less, but NOT faster -- the BPTT is bound by its per-step weight stream from L2 (19 us per MB per workgroup:
    1.98 ms at 0.75 MB, 2.60 ms with W_ih^T's 0.4 MB beside it) and the 0.62 ms it gains equal the 0.70 ms the GEMM takes:
    step 101.8 vs 101.2 ms.  Kept for the traffic figure and for hosts where HBM is the scarcer resource; not the default."""
    return (brf == 2 and os.environ.get("WESEP_BAND_DX", "0") == "1" and not seq.nvalid and dev.gn_bwd_fused_ok(geo))


def wgrad_overlap() -> bool:
    """Weight-gradient GEMMs of the blocked ResRNN on a side stream (default on; WESEP_WGRAD_OVERLAP=0
    keeps everything on the current stream)."""
    return os.environ.get("WESEP_WGRAD_OVERLAP", "1") != "0"


_SIDE_STREAMS = {}


def _side_stream(device):
    key = (device.type, device.index)
    if key not in _SIDE_STREAMS:
        # WESEP_SIDE_PRIORITY: HIP stream priority of the weight-gradient side stream (torch convention: lower = more
        # urgent; unset = the default priority)
        pr = os.environ.get("WESEP_SIDE_PRIORITY")
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=device, priority=int(pr)) if pr else torch.cuda.Stream(device=device)
    return _SIDE_STREAMS[key]


_PENDING = {}   # device key -> deferred weight-gradient jobs (closures), oldest first


def _pending(device):
    return _PENDING.setdefault((device.type, device.index), [])


_TIME_LEFT = {}   # device key -> time-view ResRNNs of the current graph whose backward (with a carrier) has not run yet


def reset_deferred_wgrads(device):
    """Drop deferred jobs (start of a step: a failed backward may have left some behind)."""
    _pending(device).clear()
    _TIME_LEFT[(device.type, device.index)] = 0


def tail_flush() -> bool:
    """Release the LAST time-view layer's own weight-gradient jobs right behind its BPTT instead of leaving them to the carriers
    (default on; WESEP_TAIL_FLUSH=0): no further pair BPTT follows to hide them under, and the main stream used to sit out their
    1.4 ms at the end of every backward (profiles/r06_bsrnn_trace_gaps.txt: 'idle between gemm_tn_bf16 -> fill')."""
    return os.environ.get("WESEP_TAIL_FLUSH", "1") != "0"


_AMAX = {}   # (device, stream) -> [int32 words, cursor]: scale words of WS_GATES_H2F, handed out one per BPTT launch


def zero_words(device, n=1):
    """`n` consecutive zeroed int32 device words: ws_gemm_p2b's running max |d(hcat)| (the scale source of WS_GATES_H2F,
    wesep_hip.h) and the counters of the "the workgroups add their partials up themselves" epilogues (ws_last_block /
    ws_tree_sum256, which leave them at zero).  Words come from a block that is zero-filled ONCE per 8192 words instead of
    once per use: a small fill is a launch of its own, and every tiny main-stream launch can sit out a whole
    weight-gradient GEMM of the side stream before it gets a CU (profiles/r04_summary.md).  A block is never re-zeroed
    while words of it may still be read (the side stream's deferred jobs): a fresh block is allocated instead and the old
    one dies with its last reference."""
    key = (device.type, device.index, L.stream_ptr().value if device.type == "cuda" and torch.cuda.is_available() else 0)
    ent = _AMAX.get(key)
    if ent is None or ent[1] + n > ent[0].numel():
        ent = _AMAX[key] = [torch.zeros(max(8192, n), device=device, dtype=torch.int32), 0]
    ent[1] += n
    return ent[0][ent[1] - n:ent[1]]


def amax_word(device):
    return zero_words(device, 1)


def mark_wgrads_ready(device):
    """Event on the current stream after which every deferred job's operands are complete (None when
    nothing is pending)."""
    if not _pending(device):
        return None
    ready = torch.cuda.Event()
    ready.record(torch.cuda.current_stream())
    return ready


def flush_deferred_wgrads(device, ready=None):
    """Launch every deferred weight-gradient job on the side stream, ordered after `ready` (default:
    everything enqueued so far on the current stream).  A TIME-VIEW recurrence keeps 128 of 256 CUs
    busy for ~5 ms: its backward marks `ready`, launches the recurrence FIRST -- so its workgroups
    take their CUs at once instead of queueing behind a full-chip GEMM wave -- and then releases the
    jobs into the other half of the chip.  The carriers flush the leftovers."""
    jobs = _pending(device)
    if not jobs:
        return
    side = _side_stream(device)
    if ready is None:
        ready = mark_wgrads_ready(device)
    with torch.cuda.stream(side):
        # `ready` is a GATE, not only a dependency: it holds the jobs back until the kernels in front of the recurrence that
        # was just launched have finished, so that the recurrence's workgroups and the jobs become runnable together and the
        # recurrence (launched first) takes its CUs first.  Without it the jobs -- whose own operands were complete long ago
        # -- fill the chip at once and the recurrence waits for a whole gemm_tnb wave: measured, step 127 -> 135.6 ms (pBSRNN),
        # 305 -> 326 ms (TF-GridNet), profiles/r04_ab_runs.md
        side.wait_event(ready)
        for job, done in jobs:
            side.wait_event(done)        # the job's own producer stream (defer_wgrad); precedes `ready` on one stream
            job(side)
    jobs.clear()


def defer_wgrad(device, job):
    """Queue a weight-gradient job (a closure that takes the side stream).  The job carries an event of its PRODUCER stream,
    recorded now: its operands are complete once everything enqueued so far on the current stream is.  With several producer
    streams (the row streams of models.tfgridnet) a flush issued from one stream must not release another stream's job before
    that stream has produced its operands."""
    done = torch.cuda.Event()
    done.record(torch.cuda.current_stream())
    _pending(device).append((job, done))


class WGradBox:
    """Hand-over slot between a ResRNN's backward (producer, side stream) and its carrier node."""
    __slots__ = ("event", "grads", "keep")

    def __init__(self):
        self.event, self.grads, self.keep = None, None, None

    def __del__(self):
        # A carrier that never ran (torch.autograd.grad over a subset of the inputs, an exception inside backward) leaves the
        # operands of an already launched job in `keep`: no stream has waited for the job, so they go back to the allocator
        # the way rounds 3-5 returned every operand -- marked as in use by the side stream
        keep = self.keep
        if keep is not None:
            try:
                for t in keep[0]:
                    t.record_stream(keep[1])
            except Exception:       # interpreter shutdown
                pass


def wgrad_hold() -> bool:
    """How the operands of a deferred weight-gradient job (d(gates), xn, hcat, d(out): 4 GB per ResRNN, 49 GB per step) stay
    valid while the side stream reads them.  Rounds 3-5 marked them `record_stream(side)`: the caching allocator then takes a
    block back only once the GPU has PASSED the side stream's job, so every allocation the host makes ahead of the GPU misses
    the cache -- 49 GB of hipMalloc per step of run-ahead, calls of 1.6-3.7 s each now and then, and a block pattern that
    depends on timing (profiles/r06_c52_diag.txt).  Default now: the job's box keeps the operands until the carrier node has
    made the consumer stream wait for the job's event and drops them there -- an ordinary stream-ordered free on the stream
    that allocated them, no event bookkeeping, the same blocks every step.  WESEP_WGRAD_HOLD=0 restores record_stream."""
    return os.environ.get("WESEP_WGRAD_HOLD", "1") != "0"


def keep_for_side(box, tensors, side, prod):
    """Called by a deferred job once its launches are on `side`: keeps `tensors` -- allocated on the stream `prod`, the one
    the ResRNN's forward and backward ran on -- valid for them (wgrad_hold)."""
    tensors = tuple(t for t in tensors if t is not None)
    if wgrad_hold():
        box.keep = (tensors, side, prod)
    else:
        for t in tensors:
            t.record_stream(side)


class WGradCarrierFn(torch.autograd.Function):
    """Delivers the LSTM / proj weight gradients of one ResRNN to autograd.

    The time-view recurrences occupy 64 of the 256 CUs for ~7 ms each; the weight-gradient GEMMs are
    a side branch of the backward graph (nothing downstream reads them), so ResRNNBlkFn.backward
    launches them on a side HIP stream where they fill the idle CUs under the NEXT layers'
    recurrences.  Autograd, however, wants a node's gradients when its backward returns.  This node
    is the way out: it is created BEFORE every ResRNN of the step (lowest sequence numbers, so the
    engine runs it after all of them), takes the weights as inputs and hands their gradients over
    once the current stream has waited for the side stream's event."""

    @staticmethod
    def forward(ctx, box, *params):
        ctx.box = box
        return params[0].new_zeros(())

    @staticmethod
    def backward(ctx, _g):
        box = ctx.box
        if box.grads is None:
            flush_deferred_wgrads(_g.device)   # leftovers of the last layers
        if box.grads is None:
            raise L.WesepHipError("weight-gradient carrier ran before its ResRNN backward")
        cur = torch.cuda.current_stream()
        cur.wait_event(box.event)
        keep, box.keep = box.keep, None
        if keep is not None:
            # `cur` is ordered behind the job now: operands allocated on `cur` are simply dropped (stream-ordered reuse is
            # safe); an operand of another stream's pool (TF-GridNet's row streams) is not ordered by this wait
            if keep[2] != cur:
                for t in keep[0]:
                    t.record_stream(keep[1])
            del keep
        grads, box.grads = box.grads, None
        for g in grads:
            g.record_stream(cur)
        return (None,) + tuple(grads)


class PackCache:
    """Derived forms of one ResRNN's LSTM / proj weights -- concatenated W_ih, MFMA-fragment packs of W_hh (per
    recurrence kernel family), W_ih, W_proj and their transposes -- built once per weight VALUE instead of once per
    forward and once more per backward (round 1: 294 pack_w launches / 3.6 ms per step).  Owned by the module that
    owns the parameters (models.bsrnn.ResRNN), so entries die with it.  Signature of the source weights: storage
    addresses + torch version counters + dev.weight_epoch() (FusedClipAdam writes parameters through raw pointers
    and bumps the epoch instead)."""

    def __init__(self):
        self.sig = None
        self.items = {}
        # round 6, prefetch_packs: the kinds asked for under the current / the previous signature (a training step asks for
        # the same ones every step), the recurrence mode they were built for, and the side stream's event behind a prefetch
        self.kinds, self.kinds_prev, self.lmode, self.ready, self.waited = [], [], None, None, set()

    @staticmethod
    def signature(params):
        return (dev.weight_epoch(),) + tuple((p.data_ptr(), p._version) for p in params)

    def begin(self, params):
        """Signature of `params` now; drops the cached packs when it moved."""
        sig = self.signature(params)
        if sig != self.sig:
            if self.kinds:
                self.kinds_prev = self.kinds
            self.sig, self.items, self.kinds, self.ready, self.waited = sig, {}, [], None, set()
        return sig

    def note(self, kind):
        if kind not in self.kinds:
            self.kinds.append(kind)

    def get(self, sig, kind, build):
        """The pack `kind` for the weights of signature `sig`; built uncached when the cache has moved on (a
        backward through a graph whose forward predates a weight update)."""
        if sig != self.sig:
            return build()
        if self.ready is not None:       # built ahead on the side stream (prefetch_packs): every consumer stream waits once
            cur = torch.cuda.current_stream()
            if cur.cuda_stream not in self.waited:
                cur.wait_event(self.ready)
                self.waited.add(cur.cuda_stream)
        if kind not in self.items:
            self.items[kind] = build()
        return self.items[kind]


class _NoCache(PackCache):
    def begin(self, params):
        return None

    def note(self, kind):
        pass

    def get(self, sig, kind, build):
        return build()


_NO_CACHE = _NoCache()


class ResRNNBlkFn(torch.autograd.Function):
    """ResRNN on the blocked layout: gates / c / h / d(h) never exist in row-major form; every
    activation byte of the recurrence moves as part of a 512-byte contiguous run (include/wesep_hip.h,
    "blocked layout BL").  Same inputs as ResRNNFn."""

    @staticmethod
    def forward(ctx, z, dummy, box, cache, view, norm_w, norm_b, wih_f, whh_f, bih_f, bhh_f, wih_r, whh_r, bih_r,
                bhh_r, proj_w, proj_b):
        """dummy/box: None, or the output and the box of this ResRNN's WGradCarrierFn -- then the ten
        LSTM / proj tensors are passed detached and their gradients travel through the box.  cache: the owning
        module's PackCache or None."""
        _need_cuda(z, "ResRNN")
        z = z.contiguous()
        R, K, Tf, N = z.shape
        if tuple(whh_f.shape) != (G4, H) or tuple(wih_f.shape) != (G4, N) or N != 128:
            raise L.WesepHipError(f"blocked ResRNN kernels are built for input 128 / hidden {H}; got "
                                  f"{tuple(wih_f.shape)}, {tuple(whh_f.shape)}")
        d = z.device
        geo, smap, seq, _ = _view_maps(view, R, K, Tf, N)
        nb = dev.bl_num_blocks(seq)
        stats = _empty(d, geo.ngroups, 2)
        dev.group_stats(z, geo, stats)
        cache = cache if cache is not None else _NO_CACHE
        sig = cache.begin((wih_f, whh_f, bih_f, bhh_f, wih_r, whh_r, bih_r, bhh_r, proj_w, proj_b))
        lmode = dev.lstm_blk_mode(seq.nseq)
        W = _resrnn_packs(cache, sig, lmode, wih_f, whh_f, bih_f, bhh_f, wih_r, whh_r, bih_r, bhh_r, proj_w)
        wcat, bcat = W("cat")
        whf, whr = W("whh")
        xn = _empty(d, nb, 32 * N)
        cbuf, hcat = _empty(d, nb, 32 * 2 * H), _empty(d, nb, 32 * 2 * H)
        cluster = dev.lstm_cluster_ok(seq, d)
        ctx.bptt = _bptt_kind(seq, d, cluster)
        # storage of the saved gates / d(gates) (dev.gates_fmt, wesep_hip.h WS_GATES_*): unorm16 gates in a BLH buffer of half
        # the bytes by default; the opt-in cluster BPTT knows the fp32 format only
        gfmt = L.GATES_F32 if ctx.bptt == "cluster" else dev.gates_fmt()
        h2 = gfmt != L.GATES_F32
        gates = _empty(d, dev.blh_floats(nb, 2 * G4)) if h2 else _empty(d, nb, 32 * 2 * G4)
        # fp16 copies of the weight-gradient GEMM's A operand [xn | h] (ABI v16; written by the two GEMMs that touch these
        # operands anyway): ws_gemm_tnb then loads 32 instead of 56 KB per block and runs ONE MFMA per product
        a16 = gfmt == L.GATES_H2F and any(ctx.needs_input_grad) and tnb_a16()
        xn16 = _empty(d, dev.blh_floats(nb, N)) if a16 else None
        hcat16 = _empty(d, dev.blh_floats(nb, 2 * H)) if a16 else None
        if dev.lstm_fuse_ok(seq.nseq, cluster):
            # band view: the recurrence computes x W_ih^T itself from the normalised input (BL(128)): the 16E-byte
            # pre-activation buffer is never written and read back (lstm_fused.hip)
            dev.gemm_p2b(A=z, lda=N, sm=seq, Wpack=None, N=0, C_out=None, A_bl=xn, stats=stats, gamma=norm_w,
                         beta=norm_b, stat_map=smap, A_bl16=xn16)
            hf = dev.lstm_fused_hfmt(gfmt)
            dev.lstm_fwd_fused(gates, cbuf, hcat, xn, W("fused8" if hf & 4 else "fused16" if hf else "fused"), bcat, seq, gfmt=gfmt,
                               hfmt=hf)
        elif cluster and h2 and dev.lstm_cluster2_on():
            # time view, 2-byte formats (round 5): the cluster kernel computes x W_ih^T itself from the normalised input
            # (lstm_cluster2.hip) -- ws_gemm_p2b only normalises (reads E, writes E (+ E / 2 for the fp16 copy) instead of
            # 17 E), the fp32 pre-activations exist only inside the predicated fall-back behind the launch (the streaming
            # pair, the whole layer again after a time-out: never NaN, no host round trip)
            dev.gemm_p2b(A=z, lda=N, sm=seq, Wpack=None, N=0, C_out=None, A_bl=xn, stats=stats, gamma=norm_w,
                         beta=norm_b, stat_map=smap, A_bl16=xn16)
            tw = dev.lstm_fwd_cluster2(gates, cbuf, hcat, xn, wcat, bcat, whf, whr, seq, dbg=_cluster_dbg())
            pre = dev.fallback_scratch(d, nb * 32 * 2 * G4)       # (untouched after a clean launch; one buffer per stream)
            dev.gemm_p2b(A=z, lda=N, sm=seq, Wpack=W("wih"), N=2 * G4, C_out=pre, bias=bcat, stats=stats,
                         gamma=norm_w, beta=norm_b, stat_map=smap, run_if=tw)
            dev.lstm_fwd(gates, cbuf, hcat, W("hh")[0], seq, lmode, run_if=tw, gfmt=gfmt, gates_in=pre)
            del pre
        else:
            # pre-activations: in `gates` itself with the fp32 format (one buffer, three lives); with the 2-byte formats a
            # scratch buffer that dies with this forward (the recurrences read it and write the unorm16 gates next to it)
            pre = _empty(d, nb, 32 * 2 * G4) if h2 else gates
            xproj = dict(A=z, lda=N, sm=seq, Wpack=W("wih"), N=2 * G4, C_out=pre, bias=bcat, A_bl=xn,
                         stats=stats, gamma=norm_w, beta=norm_b, stat_map=smap, A_bl16=xn16)
            rec = dict(gfmt=gfmt, gates_in=pre) if h2 else {}
            dev.gemm_p2b(**xproj)
            if _h2_probe() & 768 and not torch.cuda.is_available():
                # NUMERICS PROBE (CPU emulation only): the time view's pre-activations in 2 bytes -- fp16 (bit 256) / bf16 (512)
                pre.copy_(pre.half().float() if _h2_probe() & 256 else pre.bfloat16().float())
            if cluster:
                # weight-stationary cluster kernel; behind it the streaming pair predicated on the launch's timeout
                # word: two empty launches after a clean run, the whole layer again if the cluster's workgroups
                # were not co-resident (another stream / process on the GPU) -- never NaN (wesep_hip.h).  The 2-byte
                # formats leave the pre-activations intact: only the recurrence is repeated
                tw = dev.lstm_fwd_cluster(gates, cbuf, hcat, whf, whr, seq, dbg=_cluster_dbg(), **rec)
                if not h2:
                    dev.gemm_p2b(run_if=tw, **xproj)
                dev.lstm_fwd(gates, cbuf, hcat, W("hh")[0], seq, lmode, run_if=tw, **rec)
            else:
                dev.lstm_fwd(gates, cbuf, hcat, W("hh")[0], seq, lmode, **rec)
            del pre
        pw = W("pw")
        out = torch.empty_like(z)
        dev.gemm_b2p(A=hcat, K=2 * H, sm=seq, Wpack=W("proj"), C_out=out, ldc=N, bias=proj_b, R=z, a16_out=hcat16)
        if _h2_probe() and not h2:
            if _h2_probe() & 1:
                _probe_round(gates, "f16")
            if _h2_probe() & 8:
                _probe_round(gates, "u16")
            if _h2_probe() & 4:
                _probe_round(cbuf, "f16")
        if any(ctx.needs_input_grad):     # (grad mode itself is always off inside a Function's forward)
            # the backward's packs (transposed projections, BPTT weight stream) are built here, where the GPU has a
            # single stream to serve: built lazily in the backward, these 10 us launches queue behind the side stream's
            # chip-filling weight-gradient GEMMs for up to a millisecond each (round 2 profile: 4 ms per step)
            W("projT")
            bdx = ctx.bptt == "stream" and band_dx(band_rfmt(gfmt, lmode), seq, geo)
            if not bdx:
                W(_wiht_kind(2 if gfmt == L.GATES_H2F else 0))
            if ctx.bptt == "pair":
                W("hhp16" if pair_rfmt(gfmt) else "hhp")
            if (ctx.bptt == "stream" and not band_rfmt(gfmt, lmode)) or (ctx.bptt == "pair" and h2):
                W("hh")     # (the pair BPTT's predicated streaming fall-back of the 2-byte formats)
            if ctx.bptt == "stream" and band_rfmt(gfmt, lmode):
                W("hh8")
                if bdx:
                    W("wx8")
        # (with the fp16 copies the backward never reads the split-pair xn again: its 2-byte copy is saved instead)
        ctx.save_for_backward(z, stats, gates, cbuf, hcat, xn16 if a16 else xn, wcat, norm_w, norm_b, pw, whf, whr, hcat16)
        ctx.a16 = a16
        if view == "time" and box is not None and any(ctx.needs_input_grad):
            key = (d.type, d.index)
            _TIME_LEFT[key] = _TIME_LEFT.get(key, 0) + 1
        ctx.view, ctx.box, ctx.lmode, ctx.cluster, ctx.gfmt = view, box, lmode, cluster, gfmt
        ctx.packs = W
        ctx.consumed = False
        return out

    @staticmethod
    def _weight_grads(gates, xn, hcat, dout_bl, seq, nb, N, g_fmt=0, amax=None, hcat16=None):
        """[dW_ih | dW_hh | db] of both directions in one pass over each direction's dgates, and
        dW_proj / db_proj; launched on the current stream.  Returns them in parameter order.  hcat16 given: `xn` and it are
        the fp16 copies in BLH (ws_gemm_tnb a_fmt = 1; g_fmt 2 only)."""
        d = gates.device
        a_fmt = 1 if hcat16 is not None else 0
        if _h2_probe() & 192 and not torch.cuda.is_available() and not a_fmt:
            # NUMERICS PROBE (CPU emulation only: plain fp32 buffers): the A operand [xn | h] of the weight-gradient GEMMs at
            # fp16 (bit 64) / bf16 (bit 128) -- what a 2-byte A operand of ws_gemm_tnb would cost
            rnd = (lambda t: t.half().float()) if _h2_probe() & 64 else (lambda t: t.bfloat16().float())
            xn, hcat = rnd(xn), rnd(hcat)
        if os.environ.get("WESEP_PROBE_SKIP_WGRAD") == "1":   # measurement only: how much of this is exposed?
            z_ = lambda *s_: torch.zeros(*s_, device=d)
            return [z_(G4, N), z_(G4, H), z_(G4), z_(G4), z_(G4, N), z_(G4, H), z_(G4), z_(G4), z_(N, 2 * H), z_(N)]
        if _h2_probe() & 1024 and not torch.cuda.is_available() and amax is not None:
            # NUMERICS PROBE (CPU emulation only): the proj weight gradient on fp16 operands -- h as fp16, the incoming
            # gradient as fp16 scaled by the d(gates) scale of this backward (L.dgates_scale)
            S = L.dgates_scale(int(amax.reshape(-1)[0]))
            hcat = hcat.half().float()
            dout_bl = (dout_bl * S).half().float() / S
        # dW_proj^T [2H][N] = hcat^T dout (hcat as the streamed-once operand), db_proj = colsum(dout)
        ns, bps = dev.tnb_splits(nb, (2 * H) // 128)
        slab, aslab = _empty(d, ns, 2 * H * N), _empty(d, ns, N)
        dev.gemm_tnb(G=hcat, g_width=2 * H, g_off=0, g_cols=2 * H, A0=dout_bl, a0_width=N, a0_off=0, a0_cols=N,
                     nblk=nb, L_=seq.L, slab=slab, nsplit=ns, blocks_per_split=bps, aslab=aslab)
        dproj_w = _reduce_new(slab, ns, 2 * H * N, (2 * H, N)).t().contiguous()
        dproj_b = _reduce_new(aslab, ns, N, (N,))
        ns, bps = dev.tnb_splits(nb, G4 // 128)
        slab, bslab = _empty(d, ns, G4 * (N + H)), _empty(d, ns, G4)
        dwih, dwhh, db = [], [], []
        for di in (0, 1):
            dev.gemm_tnb(G=gates, g_width=2 * G4, g_off=di * G4, g_cols=G4, A0=xn, a0_width=N, a0_off=0,
                         a0_cols=N, A1=hcat16 if a_fmt else hcat, a1_width=2 * H, a1_off=di * H, a1_cols=H,
                         a1_shift=(-1 if di == 0 else 1), nblk=nb, L_=seq.L, slab=slab, nsplit=ns,
                         blocks_per_split=bps, bslab=bslab, g_fmt=g_fmt, amax=amax, a_fmt=a_fmt)
            dw = _reduce_new(slab, ns, G4 * (N + H), (G4, N + H))
            dwih.append(dw[:, :N].contiguous())
            dwhh.append(dw[:, N:].contiguous())
            db.append(_reduce_new(bslab, ns, G4, (G4,)))
        # b_ih and b_hh receive the same gradient; clone so their .grad never alias
        return [dwih[0], dwhh[0], db[0], db[0].clone(), dwih[1], dwhh[1], db[1], db[1].clone(),
                dproj_w, dproj_b]

    @staticmethod
    def backward(ctx, dout):
        if ctx.consumed:
            # BPTT turns the saved activated gates into d(gates) IN PLACE (and the deferred side-stream job reads
            # them later): a second backward through this node would silently differentiate garbage
            raise L.WesepHipError("ResRNN: second backward through the same graph (retain_graph / multi-loss loops): "
                                  "the blocked path consumes its saved gates in place; run the forward again")
        ctx.consumed = True
        z, stats, gates, cbuf, hcat, xn, wcat, norm_w, norm_b, pw, whf, whr, hcat16 = ctx.saved_tensors   # (xn: fp16 copy if hcat16)
        W = ctx.packs
        dout = dout.contiguous()
        R, K, Tf, N = z.shape
        P = R * K * Tf
        d = z.device
        geo, smap, seq, _ = _view_maps(ctx.view, R, K, Tf, N)
        nb = dev.bl_num_blocks(seq)
        box = ctx.box
        # d(hcat) = dout Wp  (+ dout itself in BL for the weight gradient)
        dh, dout_bl = _empty(d, nb, 32 * 2 * H), _empty(d, nb, 32 * N)
        dxn2 = None
        gfmt = ctx.gfmt
        # WS_GATES_H2F: the d(hcat) GEMM raises max |d(hcat)| of this launch in a device word; the BPTT scales its fp16 d(gates)
        # by the power of two it defines, the two consumers of d(gates) undo it (wesep_hip.h)
        amax = amax_word(d) if gfmt == L.GATES_H2F else None
        dev.gemm_p2b(A=dout, lda=N, sm=seq, Wpack=W("projT"), N=2 * H, C_out=dh, A_bl=dout_bl, amax=amax)
        if _h2_probe() & 16:
            _probe_round(dh, "bf16")
        # BPTT: gates (activated) -> d(pre-activation gates), in place.  A time-view recurrence leaves
        # half of the chip idle: the weight-gradient jobs deferred by the previous layers are released
        # right after it is launched
        ready = mark_wgrads_ready(d) if ctx.view == "time" else None
        # time view: the pair kernel (lstm_pair.hip) -- W_hh's hi plane resident across two workgroups per tile, on HALF
        # of the CUs, so the side stream keeps the other half.  The cluster BPTT (all 256 CUs: it evicts the
        # side-stream weight-gradient GEMMs) stays opt-in (WESEP_LSTM_CLUSTER_BWD=1).  Both work in place without a
        # device-side fall-back: FusedClipAdam.step looks at their status word (asynchronously for the pair kernel)
        g_fmt = {L.GATES_H2: 1, L.GATES_H2F: 2}.get(gfmt, 0)
        if ctx.bptt == "cluster":
            dg = gates
            dev.lstm_bwd_cluster(gates, cbuf, dh, whf, whr, seq)
        elif gfmt == L.GATES_F32:
            # ABI <= 14 format: split-pair d(gates) in place over the fp32 gates; a pair time-out has no device-side repair
            # (FusedClipAdam skips the update on the device and raises)
            dg = gates
            if ctx.bptt == "pair":
                dev.lstm_bwd_pair(gates, cbuf, dh, W("hhp"), seq, dbg=_pair_dbg())
            else:
                dev.lstm_bwd(gates, cbuf, hcat, dh, W("hh")[1], seq, ctx.lmode)
        elif ctx.bptt == "pair":
            # 2-byte formats: d(gates) go to a buffer of their own (bf16 in BLH for H2: the same bytes written as in place),
            # so the saved gates survive the launch and the streaming BPTT can stand behind it, predicated on the launch's
            # time-out word: an empty launch after a clean run, the whole BPTT again if the pair's workgroups were not
            # co-resident (a resident RCCL kernel, another process) -- no NaN reaches a consumer (wesep_hip.h)
            dg = _empty(d, nb, 32 * 2 * G4) if gfmt == L.GATES_H2S else _empty(d, dev.blh_floats(nb, 2 * G4))
            rf = pair_rfmt(gfmt)
            tw = dev.lstm_bwd_pair(gates, cbuf, dh, W("hhp16" if rf else "hhp"), seq, gfmt=gfmt, dgates=dg, repairable=True,
                                   dbg=_pair_dbg(), amax=amax, rfmt=rf, dbg_buf=_pair_stamp_buf(d, seq.L))
            dev.lstm_bwd(gates, cbuf, hcat, dh, W("hh")[1], seq, ctx.lmode, gfmt=gfmt, dgates=dg, run_if=tw, amax=amax)
        else:
            # streaming BPTT (band view): bf16 d(gates) in place over the unorm16 gates (H2) / split pairs to their own
            # buffer (H2S)
            dg = _empty(d, nb, 32 * 2 * G4) if gfmt == L.GATES_H2S else gates
            brf = band_rfmt(gfmt, ctx.lmode)
            if band_dx(brf, seq, geo):
                dxn2 = _empty(d, 2, P, N)        # d(xn) of each direction, written by the BPTT itself
            dev.lstm_bwd(gates, cbuf, hcat, dh, W("hh8") if brf else W("hh")[1], seq, ctx.lmode, gfmt=gfmt,
                         dgates=dg if gfmt == L.GATES_H2S else None, amax=amax, rfmt=brf, dxn=dxn2,
                         wxpack=W("wx8") if dxn2 is not None else None)
        if _h2_probe() & 2 and gfmt == L.GATES_F32:
            _probe_round(gates, "bf16", packed=True)
        if _h2_probe() & 32 and gfmt == L.GATES_F32 and not torch.cuda.is_available():
            # fp16 d(gates) scaled by a power of two taken from max |d(hcat)| of this launch (CPU emulation only: plain fp32)
            amax = float(dh.abs().max())
            S = 2.0 ** (10 - math.floor(math.log2(amax))) if amax > 0 and math.isfinite(amax) else 1.0
            sc = gates * S
            _PROBE_SAT[0] += int((sc.abs() > 65504.0).sum())
            _PROBE_SAT[1] = max(_PROBE_SAT[1], float(sc.abs().max()) / 65504.0)
            gates.copy_(sc.clamp(-65504.0, 65504.0).half().float() / S)
        if ready is not None:
            flush_deferred_wgrads(d, ready)
        del dh
        # weight gradients: a side branch of the graph -> deferred to the side stream when a carrier
        # will deliver them
        if box is not None:
            def job(side, gates=dg, xn=xn, hcat=hcat, dout_bl=dout_bl, seq=seq, nb=nb, N=N, box=box, g_fmt=g_fmt, amax=amax,
                    hcat16=hcat16, prod=torch.cuda.current_stream()):
                box.grads = ResRNNBlkFn._weight_grads(gates, xn, hcat, dout_bl, seq, nb, N, g_fmt, amax, hcat16)
                box.event = torch.cuda.Event()
                box.event.record(side)
                keep_for_side(box, (gates, xn, hcat, dout_bl, amax, hcat16), side, prod)
            defer_wgrad(d, job)
            wg = [None] * 10
            if ctx.view == "time":
                key = (d.type, d.index)
                _TIME_LEFT[key] = _TIME_LEFT.get(key, 1) - 1
                if _TIME_LEFT[key] <= 0 and tail_flush():
                    flush_deferred_wgrads(d)     # the graph's last time-view layer: nothing left to hide its jobs under
        else:
            wg = ResRNNBlkFn._weight_grads(dg, xn, hcat, dout_bl, seq, nb, N, g_fmt, amax, hcat16)
        del dout_bl
        # d(normalised input) = dgates Wcat -> GroupNorm backward (+ residual path)
        if dxn2 is not None:
            dxn, dxn_r = dxn2[0], dxn2[1]
        else:
            dxn, dxn_r = _empty(d, P, N), None
            dev.gemm_b2p(A=dg, K=2 * G4, sm=seq, Wpack=W(_wiht_kind(g_fmt)), C_out=dxn, ldc=N, a_fmt=dxn_fmt(g_fmt), amax=amax)
        dz = torch.empty_like(z)
        # (dgamma, dbeta): summed by the LAST workgroup of the kernel that produced the partials (wesep_hip.h, ABI v15) --
        # a separate ws_reduce_slabs launch on this stream can sit out a whole weight-gradient GEMM of the side stream
        # before its four workgroups get a CU (round 3: 7 ms per step in 62 such launches)
        dgb = _empty(d, 2, N)
        if dev.gn_bwd_fused_ok(geo):
            # band view: 16 032 groups of 16 KB -- one wave per group, x / dxn / dout cross HBM once (norm.hip)
            ns2 = min(1024, -(-geo.ngroups // 4))
            pslab = _empty(d, ns2 + dev.tree_groups(ns2), 2, N)
            dev.gn_bwd_fused(z, dxn, stats, geo, norm_w, dz, ns2, pslab, res=dout, pout=dgb,
                             counter=zero_words(d, 1 + dev.tree_groups(ns2)), dxn2=dxn_r)
        elif dev.gn_bwd_apply_pg_ok(geo):
            # time view: 1 024 groups of 256 KB: the group means first, then apply + parameter sums in ONE pass over x / dxn
            ab = _empty(d, geo.ngroups, 2)
            dev.gn_bwd_reduce(z, dxn, stats, geo, ab, gamma=norm_w)
            pslab = _empty(d, geo.ngroups + dev.tree_groups(geo.ngroups), 2, N)
            dev.gn_bwd_apply_pg(z, dxn, stats, ab, geo, dz, norm_w, pslab, dgb, zero_words(d, 1 + dev.tree_groups(geo.ngroups)),
                                res=dout)
        else:
            ab = _empty(d, geo.ngroups, 2)
            dev.gn_bwd_reduce(z, dxn, stats, geo, ab, gamma=norm_w)
            ns2 = min(1024, geo.ngroups)
            pslab = _empty(d, ns2, 2, N)
            dev.gn_param_grad(z, dxn, stats, geo, ns2, pslab)
            dev.gn_bwd_apply(z, dxn, stats, ab, geo, dz, gamma=norm_w, res=dout)
            dgb = _reduce_new(pslab, ns2, 2 * N, (2, N))
        gd = torch.zeros((), device=d) if box is not None else None
        return (dz, gd, None, None, None, dgb[0], dgb[1]) + tuple(wg)


def _bptt_kind(seq, device, cluster) -> str:
    """Which BPTT kernel a blocked-layout ResRNN runs: 'cluster' (opt-in), 'pair' (views with few long sequences: the
    time view), 'stream' (lstm_bf16*.hip)."""
    if cluster and os.environ.get("WESEP_LSTM_CLUSTER_BWD", "0") == "1":
        return "cluster"
    return "pair" if dev.lstm_pair_ok(seq, device) else "stream"


def _pair_dbg() -> int:
    """WESEP_PAIR_FORCE_TIMEOUT=1 (tests): every pair BPTT launch times out in pair 0 at step 2, so the predicated
    streaming fall-back produces the layer's d(gates).  WESEP_PAIR_STAMP=1 (measurement, tools/r06_instep_stamps.py): the
    cycle-stamped build of the kernel (dbg 2048) inside a whole training step."""
    return (8 if os.environ.get("WESEP_PAIR_FORCE_TIMEOUT", "0") == "1" else 0) | (2048 if _pair_stamp_on() else 0)


PAIR_STAMPS = []     # (stamp buffer, steps) of every pair BPTT launched with WESEP_PAIR_STAMP=1, oldest first


def _pair_stamp_on() -> bool:
    return os.environ.get("WESEP_PAIR_STAMP", "0") == "1"


def _pair_stamp_buf(device, steps):
    if not _pair_stamp_on():
        return None
    buf = torch.zeros(steps * 2 * 8 * 2 + 256 * 4 * 2, device=device)   # step stamps of pair 0 + every workgroup's wall-clock row
    PAIR_STAMPS.append((buf, steps))
    return buf


def _cluster_dbg() -> int:
    """WESEP_CLUSTER_FORCE_TIMEOUT=1 (tests): every forward cluster launch times out in workgroup 0 at step 2, so the
    predicated streaming fall-back produces the layer's result."""
    return 8 if os.environ.get("WESEP_CLUSTER_FORCE_TIMEOUT", "0") == "1" else 0


def _resrnn_packs(cache, sig, lmode, wih_f, whh_f, bih_f, bhh_f, wih_r, whh_r, bih_r, bhh_r, proj_w):
    """W(kind): the derived weight forms of one ResRNN through its PackCache (built on first use, kept until the
    weights change).  kinds: cat (wcat, bcat) | whh (contiguous W_hh pair) | hh (fwd, bwd recurrence packs of mode
    `lmode`) | hhp (pair-BPTT pack) | fused ([W_ih | W_hh] stream of lstm_fused.hip) | wih / wihT (p2b x-projection, b2p d(xn)) |
    pw (contiguous proj.weight) | proj / projT (b2p projection, p2b d(hcat))."""
    d = wih_f.device
    N = wih_f.shape[1]

    def build(kind):
        if kind == "cat":
            wcat, bcat = _empty(d, 2 * G4, N), _empty(d, 2 * G4)
            dev.lstm_cat_ih(wih_f.contiguous(), wih_r.contiguous(), bih_f, bhh_f, bih_r, bhh_r, N, wcat, bcat)
            return wcat, bcat
        if kind == "whh":
            return whh_f.contiguous(), whh_r.contiguous()
        if kind == "hh":
            pack_f, pack_b = _empty(d, L.LSTM_PACK_FLOATS), _empty(d, L.LSTM_PACK_FLOATS)
            dev.lstm_pack(*W("whh"), pack_f, pack_b, lmode)
            return pack_f, pack_b
        if kind == "hh8":                 # BPTT pack of the streaming kernel's rfmt 2 (fp16 hi + scaled-FP8 lo of 256 w)
            pack = _empty(d, L.LSTM_PACK_FLOATS)
            dev.lstm_pack_bwd_f8(*W("whh"), pack)
            return pack
        if kind == "wx8":                 # W_ih^T stream of the BPTT's own d(xn) (ws_lstm_args.dxn): fp16 hi + scaled-FP8 lo
            pack = _empty(d, L.LSTM_DX_PACK_FLOATS)
            dev.lstm_pack_dx_f8(W("cat")[0], pack)
            return pack
        if kind in ("hhp", "hhp16"):      # hhp16: fp16 hi + fp16 / FP8 lo of 256 w (the rfmt = 1 / 2 pair BPTT)
            pack = _empty(d, L.LSTM_PACK_FLOATS)
            dev.lstm_pack_pair(*W("whh"), pack, f16=pair_rfmt(L.GATES_H2F) if kind == "hhp16" else 0)
            return pack
        if kind in ("fused", "fused16", "fused8"):  # fused16: the hfmt 1 pack (256 w; W_hh part as fp16 hi / lo); fused8: hfmt 5
            fpack = _empty(d, L.LSTM_FUSED_PACK_FLOATS)        # (fp16 hi + one FP8 fragment per k-step, ABI v20)
            dev.lstm_pack_fused(wih_f.contiguous(), wih_r.contiguous(), *W("whh"), fpack,
                                hfmt={"fused": 0, "fused16": 1, "fused8": 5}[kind])
            return fpack
        if kind == "wih":
            out = _empty(d, 2 * G4 * N)
            dev.pack_w(W("cat")[0], 2 * G4, N, N, out, order=0)
            return out
        if kind in ("wihT", "wihT16", "wihT8"):   # wihT16: fp16 hi / lo (ws_pack_w_f16): d(xn) from scaled-fp16 d(gates), WS_GATES_H2F
            out = _empty(d, N * 2 * G4)              # wihT8: fp16 hi + FP8 lo fragments (ws_pack_w_f16f8, a_fmt 3)
            dev.pack_w(W("cat")[0], N, 2 * G4, N, out, trans=True, order=1, f16={"wihT": 0, "wihT16": 1, "wihT8": 2}[kind])
            return out
        if kind == "pw":
            return proj_w.contiguous()
        if kind == "proj":
            out = _empty(d, N * 2 * H)
            dev.pack_w(W("pw"), N, 2 * H, 2 * H, out, order=1)
            return out
        if kind == "projT":
            out = _empty(d, 2 * H * N)
            dev.pack_w(W("pw"), 2 * H, N, 2 * H, out, trans=True, order=0)
            return out
        raise KeyError(kind)

    def W(kind):
        # (the env-selected arithmetic is part of the key: toggling WESEP_PAIR_RF in one process -- A/B benches, tests -- must not
        #  hand an fp16-lo pack to the FP8 kernel)
        key = (kind, lmode) if kind == "hh" else (kind, pair_rfmt(L.GATES_H2F)) if kind == "hhp16" else kind
        cache.note(kind)
        return cache.get(sig, key, lambda: build(kind))

    cache.lmode = lmode
    return W


def pack_prefetch() -> bool:
    """Derived weight forms of every ResRNN built AHEAD on the side stream at the start of a training forward (opt-in:
    WESEP_PACK_PREFETCH=1; default: each is built on the main stream when it is first asked for).  The weights change every
    step, so every step rebuilds ~90 packs -- launches of 5 us, each with the 6 us gap of a dependent launch in front of it:
    about 1 ms per step of the main queue on paper, in the forward, where the side stream has nothing to do.  Measured
    (profiles/r06_summary.md): -0.27 ms in one alternating A/B, 0.0 in the next -- the main queue loses 84 launches per step and
    the forward recurrences run 0.2 ms longer beside the pack kernels: overlap on this chip is close to zero-sum once more.
    Bit-identical either way (tests/test_bsrnn_gpu.py); off by default because it buys nothing measurable."""
    return os.environ.get("WESEP_PACK_PREFETCH", "0") == "1"


def prefetch_packs(layers):
    """layers: (PackCache, the ten LSTM / proj tensors in ResRNNFn order) of every ResRNN, in forward order.  Builds, on the side
    stream and behind everything enqueued so far on the current one (the optimizer's update of these weights), the packs each
    cache was asked for under its previous signature.  The consumer waits for the layer's event when it first asks (PackCache.get).
    Allocation: the packs come from the side stream's pool and go back to it when the signature moves -- at the next prefetch,
    which again stands behind an event of the consumer stream recorded after the consumer's last use."""
    layers = [(c, p) for c, p in layers if c is not None and c.lmode is not None and p[0].is_cuda]
    if not layers or not pack_prefetch() or not torch.cuda.is_available():
        return
    d = layers[0][1][0].device
    todo = []
    for cache, params in layers:
        sig = cache.begin(params)
        if cache.kinds_prev and not cache.items:
            todo.append((cache, sig, params))
    if not todo:
        return
    side = _side_stream(d)
    gate = torch.cuda.Event()
    gate.record(torch.cuda.current_stream())
    with torch.no_grad(), torch.cuda.stream(side):
        side.wait_event(gate)
        for cache, sig, params in todo:
            W = _resrnn_packs(cache, sig, cache.lmode, *(p.detach() for p in params[:9]))
            for kind in list(cache.kinds_prev):
                W(kind)
            ready = torch.cuda.Event()
            ready.record(side)
            cache.ready, cache.waited = ready, {side.cuda_stream}


def make_wgrad_carrier(params, blocked=None):
    """(dummy, box) for one ResRNN, or None when the side-stream hand-over does not apply.  `params`:
    the ten LSTM / proj tensors in ResRNNFn order (TF-GridNet's BlstmLinearBlkFn: its eight padded weight tensors, in
    the order of its own arguments; blocked=True there -- the caller has checked its own path).  Must be called BEFORE
    the forward of every ResRNN of the step (see WGradCarrierFn)."""
    if blocked is None:
        blocked = resrnn_mode() == "blocked"
    if not (wgrad_overlap() and blocked and torch.is_grad_enabled()
            and params[0].is_cuda and all(p.requires_grad for p in params)):
        return None
    box = WGradBox()
    return WGradCarrierFn.apply(box, *params), box


def resrnn(z, view, norm_w, norm_b, *params, carrier=None, cache=None):
    """ResRNN forward (autograd-aware) on the path selected by WESEP_RESRNN.  cache: the owning module's PackCache."""
    if resrnn_mode() != "blocked":
        return ResRNNFn.apply(z, view, norm_w, norm_b, *params)
    if carrier is None:
        return ResRNNBlkFn.apply(z, None, None, cache, view, norm_w, norm_b, *params)
    dummy, box = carrier
    return ResRNNBlkFn.apply(z, dummy, box, cache, view, norm_w, norm_b, *(p.detach() for p in params))


# ---------------------------------------------------------------------------------------------
# small dense layers on [R, *] (speaker embedding side): y = x W^T + b
# ---------------------------------------------------------------------------------------------
def _w2d(w):
    return w.reshape(w.shape[0], -1).contiguous()


def _lin_fwd(x, W, b, act=0, w_off=0, ldw=None, K=None):
    M = x.shape[0]
    Nout = W.shape[0]
    K = K if K is not None else W.shape[1]
    ldw = ldw if ldw is not None else W.shape[1]
    y = _empty(x.device, M, Nout)
    vec = 3 if (K % 4 == 0 and ldw % 4 == 0 and w_off % 4 == 0 and x.shape[1] % 4 == 0) else 0
    dev.gemm_nt(A=x, a_rows=flat(x.shape[1]), M=M, N=Nout, K=K, W=W, ldw=ldw, bias=b, C_out=y,
                c_rows=flat(Nout), act=act, vec=vec, w_off=w_off)
    return y


def _lin_bwd_w(dy, x, with_bias=True):
    """dW [Nout, K] = dy^T x, db = colsum(dy); single split (M = R is tiny)."""
    M, Nout = dy.shape
    K = x.shape[1]
    pad = (-Nout) % 4                      # the kernel wants 16-byte G rows (e.g. 251 speaker classes)
    if pad:
        dyp = torch.zeros(M, Nout + pad, device=dy.device, dtype=torch.float32)
        dyp[:, :Nout] = dy
        dy = dyp
    Np = Nout + pad
    rps = -(-M // 32) * 32
    dW = _empty(dy.device, Np, K)
    db = _empty(dy.device, Np) if with_bias else None
    dev.gemm_tn(G=dy, g_rows=flat(Np), A=x, a_rows=flat(K), M=M, Nn=Np, Kk=K, slab=dW,
                slab_stride=Np * K, bslab=db, bslab_stride=Np, nsplit=1, rows_per_split=rps,
                vec=1 if K % 4 == 0 else 0)
    if pad:
        dW = dW[:Nout].contiguous()
        db = db[:Nout].contiguous() if with_bias else None
    return dW, db


def _lin_bwd_x(dy, W, T=None, src_off=0, rows=None, cols=None, lds=None):
    """dx = dy W (optionally * (1 - T^2)); W [rows, cols] with leading dim lds."""
    rows = rows if rows is not None else W.shape[0]
    cols = cols if cols is not None else W.shape[1]
    lds = lds if lds is not None else W.shape[1]
    WT = _empty(dy.device, cols, rows)
    dev.transpose(W, rows, cols, lds, WT, src_off=src_off)
    dx = _empty(dy.device, dy.shape[0], cols)
    vec = 3 if rows % 4 == 0 else 0
    dev.gemm_nt(A=dy, a_rows=flat(rows), M=dy.shape[0], N=cols, K=rows, W=WT, ldw=rows, C_out=dx,
                c_rows=flat(cols), T=T, vec=vec)
    return dx


class LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, W, b):
        _need_cuda(x, "Linear")
        x, W2 = x.contiguous(), _w2d(W)
        ctx.save_for_backward(x, W2)
        ctx.wshape = W.shape
        return _lin_fwd(x, W2, b)

    @staticmethod
    def backward(ctx, dy):
        x, W2 = ctx.saved_tensors
        dy = dy.contiguous()
        dW, db = _lin_bwd_w(dy, x)
        dx = _lin_bwd_x(dy, W2) if ctx.needs_input_grad[0] else None
        return dx, dW.view(ctx.wshape), db


class SpkTransformFn(torch.autograd.Function):
    """SpeakerTransform (speaker.py:26-49): Conv1d(k=1) 256->128, 128->128 + Tanh, 128->256."""

    @staticmethod
    def forward(ctx, e, w0, b0, w1, b1, w3, b3):
        _need_cuda(e, "SpeakerTransform")
        e = e.contiguous()
        W0, W1, W3 = _w2d(w0), _w2d(w1), _w2d(w3)
        h0 = _lin_fwd(e, W0, b0)
        h1 = _lin_fwd(h0, W1, b1, act=1)
        y = _lin_fwd(h1, W3, b3)
        ctx.save_for_backward(e, h0, h1, W0, W1, W3)
        ctx.shapes = (w0.shape, w1.shape, w3.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        e, h0, h1, W0, W1, W3 = ctx.saved_tensors
        dy = dy.contiguous()
        dW3, db3 = _lin_bwd_w(dy, h1)
        dp1 = _lin_bwd_x(dy, W3, T=h1)           # d(pre-tanh) of layer 1
        dW1, db1 = _lin_bwd_w(dp1, h0)
        dh0 = _lin_bwd_x(dp1, W1)
        dW0, db0 = _lin_bwd_w(dh0, e)
        de = _lin_bwd_x(dh0, W0) if ctx.needs_input_grad[0] else None
        s0, s1, s3 = ctx.shapes
        return de, dW0.view(s0), db0, dW1.view(s1), db1, dW3.view(s3), db3


# ---------------------------------------------------------------------------------------------
# speaker fusion on Z: out = z * (a0 + a[r]) + b[r]      (speaker.py:81-125, norm.py:118-139)
# ---------------------------------------------------------------------------------------------
def _affine_splits(rows_per_r):
    return max(1, min(64, rows_per_r // 256))


class AffineFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, a, b, a0):
        _need_cuda(z, "SpeakerFuse")
        z = z.contiguous()
        R, K, Tf, N = z.shape
        a = a.contiguous() if a is not None else None
        b = b.contiguous() if b is not None else None
        out = torch.empty_like(z)
        dev.affine_fwd(z, a, b, a0, R * K * Tf, K * Tf, N, out)
        ctx.save_for_backward(z, a)
        ctx.a0, ctx.has_b = a0, b is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        z, a = ctx.saved_tensors
        dout = dout.contiguous()
        R, K, Tf, N = z.shape
        ns = _affine_splits(K * Tf)
        da_slab = _empty(z.device, ns, R, N) if a is not None else None
        db_slab = _empty(z.device, ns, R, N) if ctx.has_b else None
        dz = torch.empty_like(z)
        # the splits are summed by the last workgroup of the launch (no ws_reduce_slabs launches: see ResRNNBlkFn.backward)
        da = _empty(z.device, R, N) if a is not None else None
        db = _empty(z.device, R, N) if ctx.has_b else None
        dev.affine_bwd(dout, z, a, ctx.a0, R * K * Tf, K * Tf, N, ns, dz, da_slab, db_slab, da=da, db=db,
                       counter=zero_words(z.device, R) if (da is not None or db is not None) else None)
        return dz, da, db, None


class ConcatFuseFn(torch.autograd.Function):
    """SpeakerFuseLayer 'concat' (speaker.py:90-102): Linear(cat[x, e]) = x Wx^T + (e We^T + b)."""

    @staticmethod
    def forward(ctx, z, e, W, b):
        _need_cuda(z, "SpeakerFuse(concat)")
        z, e, W = z.contiguous(), e.contiguous(), W.contiguous()
        R, K, Tf, N = z.shape
        E = e.shape[1]
        P = R * K * Tf
        c = _lin_fwd(e, W, b, w_off=N, ldw=N + E, K=E)                     # [R, N]
        out = torch.empty_like(z)
        dev.gemm_nt(A=z, a_rows=flat(N), M=P, N=N, K=N, W=W, ldw=N + E, C_out=out, c_rows=flat(N))
        dev.affine_fwd(out, None, c, 1.0, P, K * Tf, N, out)
        ctx.save_for_backward(z, e, W)
        return out

    @staticmethod
    def backward(ctx, dout):
        z, e, W = ctx.saved_tensors
        dout = dout.contiguous()
        R, K, Tf, N = z.shape
        E = e.shape[1]
        P = R * K * Tf
        d = z.device
        ns = _affine_splits(K * Tf)
        dc_slab = _empty(d, ns, R, N)
        dev.affine_bwd(dout, None, None, 1.0, P, K * Tf, N, ns, None, None, dc_slab)
        dc = _reduce_new(dc_slab, ns, R * N, (R, N))
        dW = _empty(d, N, N + E)
        # dWx = dout^T z  -> columns [0, N)
        nsplit, rps = dev.tn_splits(P)
        slab = _empty(d, nsplit, N * N)
        dev.gemm_tn(G=dout, g_rows=flat(N), A=z, a_rows=flat(N), M=P, Nn=N, Kk=N, slab=slab,
                    slab_stride=N * N, nsplit=nsplit, rows_per_split=rps)
        dev.reduce_slabs(slab, nsplit, N * N, N * N, dW, w=N, ldo=N + E)
        # dWe = dc^T e -> columns [N, N+E);  db = colsum(dc)
        dWe, db = _lin_bwd_w(dc, e)
        dev.reduce_slabs(dWe, 1, N * E, N * E, dW, w=E, ldo=N + E, out_off=N)
        dz = _empty(d, R, K, Tf, N)
        WxT = _empty(d, N, N)
        dev.transpose(W, N, N, N + E, WxT)
        dev.gemm_nt(A=dout, a_rows=flat(N), M=P, N=N, K=N, W=WxT, ldw=N, C_out=dz, c_rows=flat(N))
        de = _lin_bwd_x(dc, W, src_off=N, rows=N, cols=E, lds=N + E) if ctx.needs_input_grad[1] else None
        return dz, de, dW, db


# ---------------------------------------------------------------------------------------------
# per-band plans: device descriptor tables for the grouped (32-band) launches
# ---------------------------------------------------------------------------------------------
def mask_nn() -> bool:
    """The mask MLP's data-gradient GEMMs read the weights as they lie (ws_gemm_nt_args.vec bit 3, split-bf16 kernel) instead of
    transposed copies made every step; WESEP_GEMM_NN=0 restores the copies."""
    return dev.gemm_mode() == "bf16x3" and os.environ.get("WESEP_GEMM_NN", "1") != "0"


class BandPlan:
    """Band tables + cached group descriptors for BN[i] (bsrnn.py:252-258) and mask[i]
    (bsrnn.py:271-282).  Descriptors hold parameter pointers, so they are rebuilt only when a
    parameter's storage moves (e.g. .cuda(), load of a new module) or the batch geometry changes."""

    def __init__(self, band_width, feature_dim, device):
        self.dev = device
        self.bands = dev.BandTables(band_width, device)
        self.bw = [int(b) for b in band_width]
        self.f0 = [int(f) for f in self.bands.f0_host]
        self.K = len(self.bw)
        self.N = feature_dim
        N = feature_dim
        H1 = 4 * N
        # flat-gradient offsets
        self.bn_woff = np.concatenate([[0], np.cumsum([N * 2 * b for b in self.bw])]).astype(np.int64)
        self.m_w3off = np.concatenate([[0], np.cumsum([4 * b * H1 for b in self.bw])]).astype(np.int64)
        self.m_b3off = np.concatenate([[0], np.cumsum([4 * b for b in self.bw])]).astype(np.int64)
        # persistent transposed-weight workspaces (descriptors point into them)
        self.bn_wT = _empty(device, int(self.bn_woff[-1]))
        self.m_w1T = _empty(device, self.K * N * H1)
        self.m_w2T = _empty(device, self.K * H1 * H1)
        self.m_w3T = _empty(device, int(self.m_w3off[-1]))
        self._cache = {}

    def _up(self, arr):
        return L.upload_struct_array(arr, self.dev)

    def _key(self, params, R, Tf):
        return (R, Tf) + tuple(p.data_ptr() for p in params)

    # ---- BN ------------------------------------------------------------------------------
    def bn_desc(self, params, R, Tf):
        key = ("bn",) + self._key(params, R, Tf)
        if key not in self._cache:
            K, N = self.K, self.N
            nt = np.zeros(K, dtype=L.GROUP_NT_DTYPE)
            tn = np.zeros(K, dtype=L.GROUP_TN_DTYPE)
            dx = np.zeros(K, dtype=L.GROUP_NT_DTYPE)
            for g in range(K):
                gw, gb, cw, cb = params[4 * g:4 * g + 4]
                bw2 = 2 * self.bw[g]
                nt[g] = (cw.data_ptr(), cb.data_ptr(), gw.data_ptr(), gb.data_ptr(),
                         2 * self.f0[g], g * Tf * N, g, bw2, N, bw2, 0)
                tn[g] = (gw.data_ptr(), gb.data_ptr(), g * Tf * N, 2 * self.f0[g], g,
                         int(self.bn_woff[g]), g * N, N, bw2, 0, 0)
                dx[g] = (self.bn_wT.data_ptr() + 4 * int(self.bn_woff[g]), 0, 0, 0,
                         g * Tf * N, 2 * self.f0[g], 0, N, bw2, N, 0)
            self._cache = {k: v for k, v in self._cache.items() if k[0] != "bn"}
            self._cache[key] = (self._up(nt), self._up(tn), self._up(dx))
        return self._cache[key]

    # ---- mask ----------------------------------------------------------------------------
    def mask_desc(self, params, R, Tf):
        nn = mask_nn()
        key = ("mask", nn) + self._key(params, R, Tf)
        if key not in self._cache:
            K, N = self.K, self.N
            H1 = 4 * N
            M = R * Tf
            d = {n: np.zeros(K, dtype=L.GROUP_NT_DTYPE) for n in ("l1", "l2", "l3", "dh2", "dh1", "dxn")}
            t = {n: np.zeros(K, dtype=L.GROUP_TN_DTYPE) for n in ("w3", "w2", "w1")}
            gtab = np.zeros(K, dtype=np.uint64)
            for g in range(K):
                gw, gb, w1, b1, w2, b2, w3, b3 = params[8 * g:8 * g + 8]
                bw4 = 4 * self.bw[g]
                hoff = g * M * H1
                zoff = g * Tf * N
                gtab[g] = gw.data_ptr()
                d["l1"][g] = (w1.data_ptr(), b1.data_ptr(), gw.data_ptr(), gb.data_ptr(), zoff, hoff, g, N, H1, N, 0)
                d["l2"][g] = (w2.data_ptr(), b2.data_ptr(), 0, 0, hoff, hoff, 0, H1, H1, H1, 0)
                d["l3"][g] = (w3.data_ptr(), b3.data_ptr(), 0, 0, hoff, 4 * self.f0[g], 0, H1, bw4, H1, 0)
                if nn:    # the data-gradient GEMMs take the weights AS THEY LIE (ws_gemm_nt_args.vec bit 3: W'[n][k] = W[k * ldw + n])
                    d["dh2"][g] = (w3.data_ptr(), 0, 0, 0, 4 * self.f0[g], hoff, 0, bw4, H1, H1, 0)
                    d["dh1"][g] = (w2.data_ptr(), 0, 0, 0, hoff, hoff, 0, H1, H1, H1, 0)
                    d["dxn"][g] = (w1.data_ptr(), 0, 0, 0, hoff, zoff, 0, H1, N, N, 0)
                else:
                    d["dh2"][g] = (self.m_w3T.data_ptr() + 4 * int(self.m_w3off[g]), 0, 0, 0,
                                   4 * self.f0[g], hoff, 0, bw4, H1, bw4, 0)
                    d["dh1"][g] = (self.m_w2T.data_ptr() + 4 * g * H1 * H1, 0, 0, 0, hoff, hoff, 0, H1, H1, H1, 0)
                    d["dxn"][g] = (self.m_w1T.data_ptr() + 4 * g * N * H1, 0, 0, 0, hoff, zoff, 0, H1, N, H1, 0)
                t["w3"][g] = (0, 0, 4 * self.f0[g], hoff, 0, int(self.m_w3off[g]), int(self.m_b3off[g]), bw4, H1, 0, 0)
                t["w2"][g] = (0, 0, hoff, hoff, 0, g * H1 * H1, g * H1, H1, H1, 0, 0)
                t["w1"][g] = (gw.data_ptr(), gb.data_ptr(), hoff, zoff, g, g * H1 * N, g * H1, H1, N, 0, 0)
            self._cache = {k: v for k, v in self._cache.items() if k[0] != "mask"}
            out = {k: self._up(v) for k, v in d.items()}
            out.update({"t" + k: self._up(v) for k, v in t.items()})
            out["gamma_tab"] = torch.from_numpy(gtab.view(np.int64)).to(self.dev)
            self._cache[key] = out
        return self._cache[key]


# ---------------------------------------------------------------------------------------------
# STFT + band split + per-band GroupNorm + 1x1 conv     (bsrnn.py:309-336)
# ---------------------------------------------------------------------------------------------
class BandSplitFn(torch.autograd.Function):
    """inputs: wav [R, T], plan, then per band (gn.weight, gn.bias, conv.weight, conv.bias).
    outputs: z0 [R, K, Tf, N], xbs [R*Tf, 2F] (band-split mixture spectrogram, no grad)."""

    @staticmethod
    def forward(ctx, wav, plan, *params):
        _need_cuda(wav, "BSRNN")
        wav = wav.contiguous()
        R, T = wav.shape
        K, N = plan.K, plan.N
        Tf = 1 + T // HOP
        M = R * Tf
        d = wav.device
        xbs = _empty(d, M, 2 * NBIN)
        dev.stft_bandsplit(wav, plan.bands, xbs)
        geo = Geom(R * K, K, Tf * 2 * NBIN, 0, 2 * NBIN, Tf, 128, K, plan.bands.bw2, plan.bands.off2)
        stats = _empty(d, R * K, 2)
        dev.group_stats(xbs, geo, stats)
        nt, _, _ = plan.bn_desc(params, R, Tf)
        z0 = _empty(d, R, K, Tf, N)
        dev.gemm_nt(A=xbs, a_rows=flat(2 * NBIN), M=M, C_out=z0, c_rows=Rows(Tf, K * Tf * N, N),
                    stats=stats, stat_map=StatMap(Tf, K, 1, 0, 0), groups=nt, ngroups=K, max_n=N, vec=0)
        ctx.save_for_backward(xbs, stats, *params)
        ctx.plan, ctx.dims = plan, (R, T, Tf)
        ctx.mark_non_differentiable(xbs)
        return z0, xbs

    @staticmethod
    def backward(ctx, dz0, _dxbs):
        xbs, stats = ctx.saved_tensors[:2]
        params = ctx.saved_tensors[2:]
        plan = ctx.plan
        R, T, Tf = ctx.dims
        K, N = plan.K, plan.N
        M = R * Tf
        d = xbs.device
        dz0 = dz0.contiguous()
        _, tn, dxd = plan.bn_desc(params, R, Tf)
        geo = Geom(R * K, K, Tf * 2 * NBIN, 0, 2 * NBIN, Tf, 128, K, plan.bands.bw2, plan.bands.off2)
        smap = StatMap(Tf, K, 1, 0, 0)
        # conv weight / bias gradients (A = re-normalised band spectrogram)
        nsplit, rps = dev.tn_splits(M)
        wtot = int(plan.bn_woff[-1])
        slab, bslab = _empty(d, nsplit, wtot), _empty(d, nsplit, K * N)
        dev.gemm_tn(G=dz0, g_rows=Rows(Tf, K * Tf * N, N), A=xbs, a_rows=flat(2 * NBIN), M=M, slab=slab,
                    slab_stride=wtot, bslab=bslab, bslab_stride=K * N, nsplit=nsplit, rows_per_split=rps,
                    stats=stats, stat_map=smap, groups=tn, ngroups=K, max_n=N, max_k=128, vec=0)
        dW = _reduce_new(slab, nsplit, wtot, (wtot,))
        dB = _reduce_new(bslab, nsplit, K * N, (K * N,))
        # d(normalised spectrogram) -> GroupNorm affine gradients (dX itself is never needed)
        for g in range(K):
            bw2 = 2 * plan.bw[g]
            dev.transpose(params[4 * g + 2].reshape(N, bw2), N, bw2, bw2, plan.bn_wT,
                          dst_off=int(plan.bn_woff[g]))
        dxn = _empty(d, M, 2 * NBIN)
        dev.gemm_nt(A=dz0, a_rows=Rows(Tf, K * Tf * N, N), M=M, C_out=dxn, c_rows=flat(2 * NBIN),
                    groups=dxd, ngroups=K, max_n=128, vec=3)
        ns2 = min(64, R)
        pslab = _empty(d, ns2, K, 2, 128)
        dev.gn_param_grad(xbs, dxn, stats, geo, ns2, pslab)
        dgb = _reduce_new(pslab, ns2, K * 2 * 128, (K, 2, 128))
        grads = []
        for g in range(K):
            bw2 = 2 * plan.bw[g]
            o = int(plan.bn_woff[g])
            grads += [dgb[g, 0, :bw2], dgb[g, 1, :bw2], dW[o:o + N * bw2].view(N, bw2, 1), dB[g * N:(g + 1) * N]]
        return (None, None) + tuple(grads)


# ---------------------------------------------------------------------------------------------
# mask MLP + GLU complex mask + iSTFT     (bsrnn.py:366-389)
# ---------------------------------------------------------------------------------------------
class MaskDecodeFn(torch.autograd.Function):
    """inputs: z [R,K,Tf,N], xbs, plan, T, then per band (gn.w, gn.b, w1, b1, w2, b2, w3, b3).
    output: est [R, T]."""

    @staticmethod
    def forward(ctx, z, xbs, plan, T, *params):
        _need_cuda(z, "BSRNN")
        z = z.contiguous()
        R, K, Tf, N = z.shape
        H1 = 4 * N
        M = R * Tf
        d = z.device
        D = plan.mask_desc(params, R, Tf)
        geo = Geom(R * K, 1, Tf * N, 0, N, Tf, N, K)
        stats = _empty(d, R * K, 2)
        dev.group_stats(z, geo, stats)
        h1, h2 = _empty(d, K, M, H1), _empty(d, K, M, H1)
        dev.gemm_nt(A=z, a_rows=Rows(Tf, K * Tf * N, N), M=M, C_out=h1, c_rows=flat(H1), stats=stats,
                    stat_map=StatMap(Tf, K, 1, 0, 0), act=1, groups=D["l1"], ngroups=K, max_n=H1)
        dev.gemm_nt(A=h1, a_rows=flat(H1), M=M, C_out=h2, c_rows=flat(H1), act=1, groups=D["l2"],
                    ngroups=K, max_n=H1)
        m3 = _empty(d, M, 4 * NBIN)
        dev.gemm_nt(A=h2, a_rows=flat(H1), M=M, C_out=m3, c_rows=flat(4 * NBIN), groups=D["l3"],
                    ngroups=K, max_n=4 * max(plan.bw))
        frames = _empty(d, M, 512)
        dev.mask_istft_frames(xbs, m3, R, Tf, plan.bands, frames)
        est = _empty(d, R, T)
        dev.istft_ola(frames, R, Tf, T, est)
        ctx.save_for_backward(z, xbs, stats, h1, h2, m3, *params)
        ctx.plan, ctx.T = plan, T
        return est

    @staticmethod
    def backward(ctx, dest):
        z, xbs, stats, h1, h2, m3 = ctx.saved_tensors[:6]
        params = ctx.saved_tensors[6:]
        plan, T = ctx.plan, ctx.T
        R, K, Tf, N = z.shape
        H1 = 4 * N
        M = R * Tf
        d = z.device
        D = plan.mask_desc(params, R, Tf)
        dest = dest.contiguous()
        dm3 = _empty(d, M, 4 * NBIN)
        dev.mask_istft_bwd(dest, xbs, m3, R, Tf, T, plan.bands, dm3)
        nsplit, rps = dev.tn_splits(M)
        maxb4 = 4 * max(plan.bw)
        # layer 3
        w3tot, b3tot = int(plan.m_w3off[-1]), int(plan.m_b3off[-1])
        slab, bslab = _empty(d, nsplit, w3tot), _empty(d, nsplit, b3tot)
        dev.gemm_tn(G=dm3, g_rows=flat(4 * NBIN), A=h2, a_rows=flat(H1), M=M, slab=slab, slab_stride=w3tot,
                    bslab=bslab, bslab_stride=b3tot, nsplit=nsplit, rows_per_split=rps, groups=D["tw3"],
                    ngroups=K, max_n=maxb4, max_k=H1)
        dW3 = _reduce_new(slab, nsplit, w3tot, (w3tot,))
        dB3 = _reduce_new(bslab, nsplit, b3tot, (b3tot,))
        nn = mask_nn()
        vnn = 3 | (8 if nn else 0)
        if not nn:      # (rounds 1-5: 3 K transposes of the weights per step -- 93 launches of 6 us, each with its launch gap, at the
            for g in range(K):                     # head of the backward; the GEMMs stage W as it lies now, round 6)
                bw4 = 4 * plan.bw[g]
                dev.transpose(params[8 * g + 6].reshape(bw4, H1), bw4, H1, H1, plan.m_w3T,
                              dst_off=int(plan.m_w3off[g]))
                dev.transpose(params[8 * g + 4].reshape(H1, H1), H1, H1, H1, plan.m_w2T, dst_off=g * H1 * H1)
                dev.transpose(params[8 * g + 2].reshape(H1, N), H1, N, N, plan.m_w1T, dst_off=g * N * H1)
        dh2 = _empty(d, K, M, H1)
        dev.gemm_nt(A=dm3, a_rows=flat(4 * NBIN), M=M, C_out=dh2, c_rows=flat(H1), T=h2, groups=D["dh2"],
                    ngroups=K, max_n=H1, vec=vnn)
        # layer 2
        slab, bslab = _empty(d, nsplit, K * H1 * H1), _empty(d, nsplit, K * H1)
        dev.gemm_tn(G=dh2, g_rows=flat(H1), A=h1, a_rows=flat(H1), M=M, slab=slab, slab_stride=K * H1 * H1,
                    bslab=bslab, bslab_stride=K * H1, nsplit=nsplit, rows_per_split=rps, groups=D["tw2"],
                    ngroups=K, max_n=H1, max_k=H1)
        dW2 = _reduce_new(slab, nsplit, K * H1 * H1, (K, H1, H1, 1))
        dB2 = _reduce_new(bslab, nsplit, K * H1, (K, H1))
        dh1 = _empty(d, K, M, H1)
        dev.gemm_nt(A=dh2, a_rows=flat(H1), M=M, C_out=dh1, c_rows=flat(H1), T=h1, groups=D["dh1"],
                    ngroups=K, max_n=H1, vec=vnn)
        del dh2
        # layer 1 (A = re-normalised z band rows)
        slab, bslab = _empty(d, nsplit, K * H1 * N), _empty(d, nsplit, K * H1)
        dev.gemm_tn(G=dh1, g_rows=flat(H1), A=z, a_rows=Rows(Tf, K * Tf * N, N), M=M, slab=slab,
                    slab_stride=K * H1 * N, bslab=bslab, bslab_stride=K * H1, nsplit=nsplit,
                    rows_per_split=rps, stats=stats, stat_map=StatMap(Tf, K, 1, 0, 0), groups=D["tw1"],
                    ngroups=K, max_n=H1, max_k=N)
        dW1 = _reduce_new(slab, nsplit, K * H1 * N, (K, H1, N, 1))
        dB1 = _reduce_new(bslab, nsplit, K * H1, (K, H1))
        del slab, bslab
        dxn = _empty(d, R, K, Tf, N)
        dev.gemm_nt(A=dh1, a_rows=flat(H1), M=M, C_out=dxn, c_rows=Rows(Tf, K * Tf * N, N), groups=D["dxn"],
                    ngroups=K, max_n=N, vec=vnn)
        del dh1
        # GroupNorm backward with per-band gamma
        geo = Geom(R * K, 1, Tf * N, 0, N, Tf, N, K)
        ab = _empty(d, R * K, 2)
        dev.gn_bwd_reduce(z, dxn, stats, geo, ab, gamma_tab=D["gamma_tab"])
        ns2 = min(64, R)
        pslab = _empty(d, ns2, K, 2, N)
        dev.gn_param_grad(z, dxn, stats, geo, ns2, pslab)
        dgb = _reduce_new(pslab, ns2, K * 2 * N, (K, 2, N))
        dz = torch.empty_like(z)
        dev.gn_bwd_apply(z, dxn, stats, ab, geo, dz, gamma_tab=D["gamma_tab"])
        grads = []
        for g in range(K):
            bw4 = 4 * plan.bw[g]
            o3, ob3 = int(plan.m_w3off[g]), int(plan.m_b3off[g])
            grads += [dgb[g, 0], dgb[g, 1], dW1[g], dB1[g], dW2[g], dB2[g],
                      dW3[o3:o3 + bw4 * H1].view(bw4, H1, 1), dB3[ob3:ob3 + bw4]]
        return (dz, None, None, None) + tuple(grads)


# ---------------------------------------------------------------------------------------------
# SI-SDR loss (auraloss.time.SISDRLoss, losses.py:24-25)
# ---------------------------------------------------------------------------------------------
class SISDRFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, est, tgt, eps):
        _need_cuda(est, "SISDRLoss")
        est, tgt = est.contiguous().float(), tgt.contiguous().float()
        R = est.shape[0]
        rowstat = _empty(est.device, R, 8)
        loss = _empty(est.device, 1)
        dev.sisdr_fwd(est, tgt, rowstat, loss, eps)
        ctx.save_for_backward(est, tgt, rowstat)
        return loss.view(())

    @staticmethod
    def backward(ctx, gout):
        est, tgt, rowstat = ctx.saved_tensors
        dest = torch.empty_like(est)
        dev.sisdr_bwd(est, tgt, rowstat, gout.contiguous().view(1).float(), dest)
        return dest, None, None
