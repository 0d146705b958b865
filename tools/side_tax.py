"""What the side stream costs the main stream, from a rocprofv3 kernel trace of the headline step.

    python tools/side_tax.py <..._kernel_trace.csv> [--steps 4]

For every launch of the main queue's large kernels inside the steady-state window (delimited by clip_adam_kernel, as
tools/trace_gaps.py does): its duration and the share of its interval during which a kernel of ANOTHER queue was running, by
the name of that kernel.  Launches are then binned by that share (none / under half / most of the interval) so that the
same kernel at the same shape is compared with and without company: the difference is the tax."""
import argparse
import collections
import csv
import statistics as st

TARGETS = ("lstm_bwd_pair_kernel", "lstm_bwd_bf16_kernel", "lstm_bwd_band", "lstm_fwd_fused64", "lstm_fwd_cluster2_kernel",
           "gemm_b2p_kernel<2>", "gemm_b2p_kernel<0>", "gemm_p2b_kernel", "gn_bwd_fused_kernel", "gn_bwd_apply_pg_kernel")


def short(n):
    n = n.replace("void ", "")
    return n.split("(")[0][:44]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--steps", type=int, default=4)
    a = ap.parse_args()
    rows = []
    with open(a.trace) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]))
    rows.sort()
    adam = [s for s, e, q, n in rows if n.startswith("clip_adam_kernel")]
    if len(adam) < a.steps + 1:
        raise SystemExit(f"only {len(adam)} clip_adam launches in the trace")
    t0, t1 = adam[-a.steps - 1], adam[-1]
    win = [r for r in rows if t0 <= r[0] < t1]
    byq = collections.Counter(r[2] for r in win)
    mainq = byq.most_common(1)[0][0]
    others = [r for r in win if r[2] != mainq]
    print(f"window: {a.steps} steps, {(t1 - t0) / a.steps / 1e6:.2f} ms per step; main queue {mainq} "
          f"({byq[mainq] / a.steps:.0f} launches/step), other queues {sum(v for k, v in byq.items() if k != mainq) / a.steps:.0f}")
    for tgt in TARGETS:
        ls = [r for r in win if r[2] == mainq and tgt in r[3]]
        if not ls:
            continue
        recs = []
        for s, e, _, n in ls:
            ov = collections.Counter()
            for s2, e2, _, n2 in others:
                if e2 <= s or s2 >= e:
                    continue
                ov[short(n2)] += min(e, e2) - max(s, s2)
            recs.append(((e - s) / 1e3, sum(ov.values()) / (e - s), ov))
        durs = [d for d, _, _ in recs]
        print(f"\n{tgt}: {len(ls) / a.steps:.0f} launches/step, mean {st.mean(durs):.0f} us, min {min(durs):.0f}, max {max(durs):.0f}, "
              f"sd {st.pstdev(durs):.0f}; total {sum(durs) / a.steps / 1e3:.2f} ms/step")
        for lo, hi, nm in ((-1, 0.02, "alone (< 2 % overlapped)"), (0.02, 0.5, "2-50 % overlapped"), (0.5, 9e9, "> 50 % overlapped")):
            b = [(d, o) for d, f, o in recs if lo < f <= hi]
            if not b:
                continue
            who = collections.Counter()
            for _, o in b:
                who.update(o)
            tot = sum(who.values()) or 1
            top = ", ".join(f"{k} {100 * v / tot:.0f} %" for k, v in who.most_common(3))
            ds = [d for d, _ in b]
            print(f"   {nm:26s} n = {len(b):3d}  mean {st.mean(ds):7.0f} us  min {min(ds):7.0f}  max {max(ds):7.0f}   beside: {top}")
        if len(recs) > 1:
            lone = [d for d, f, _ in recs if f <= 0.02]
            base = st.mean(lone) if lone else min(durs)
            print(f"   tax vs {'its own lone launches' if lone else 'its fastest launch'}: "
                  f"{(sum(durs) - base * len(durs)) / a.steps / 1e3:.2f} ms/step")


if __name__ == "__main__":
    main()
