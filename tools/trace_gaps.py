"""Idle time on the main stream of the headline step, from a rocprofv3 kernel trace (…_kernel_trace.csv).

    python tools/trace_gaps.py gpurun_out/r03_kernel_trace.csv [--steps 5]

Per hardware queue: busy time (union of kernel intervals), idle time between kernels inside the steady-state window
(the last `steps` occurrences of clip_adam_kernel delimit the steps), and the kernels that precede the longest gaps.
Tells whether the step is bound by the kernels of the main stream or by the host feeding it."""
import argparse
import csv
import collections


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--top", type=int, default=12)
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
    print(f"window: {a.steps} steps, {(t1 - t0) / a.steps / 1e6:.2f} ms per step, {len(win) / a.steps:.0f} launches per step")
    byq = collections.defaultdict(list)
    for r in win:
        byq[r[2]].append(r)
    for q, rs in sorted(byq.items(), key=lambda kv: -len(kv[1])):
        busy, gaps, cur_end, prev = 0, [], None, None
        for s, e, _, n in rs:
            if cur_end is None or s > cur_end:
                if cur_end is not None:
                    gaps.append((s - cur_end, prev, n))
                busy += e - s
                cur_end = e
            else:
                busy += max(0, e - cur_end)
                cur_end = max(cur_end, e)
            prev = n
        idle = sum(g for g, _, _ in gaps)
        print(f"\nqueue {q}: {len(rs) / a.steps:.0f} launches/step, busy {busy / a.steps / 1e6:.2f} ms/step, idle between "
              f"kernels {idle / a.steps / 1e6:.2f} ms/step")
        hist = collections.Counter()
        for g, p, n in gaps:
            hist[(p[:40], n[:40])] += g
        for (p, n), g in hist.most_common(a.top):
            print(f"   {g / a.steps / 1e3:8.1f} us/step idle between  {p:40s} -> {n}")
        small = sum(g for g, _, _ in gaps if g < 20000)
        print(f"   gaps < 20 us: {small / a.steps / 1e6:.2f} ms/step of {len([1 for g, _, _ in gaps if g < 20000]) / a.steps:.0f}")


if __name__ == "__main__":
    main()
