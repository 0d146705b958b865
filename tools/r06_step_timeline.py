"""One steady-state step of the headline bench as a timeline, from a rocprofv3 kernel trace (…_kernel_trace.csv): every launch
with its queue, start (ms since the step's start) and duration, runs of equal kernel names on a queue merged.  Round 6: which
main-queue kernels run while the side queue is EMPTY (mask-estimation backward at the head of the backward pass).

    python tools/r06_step_timeline.py <kernel_trace.csv> [--min-us 150]
"""
import argparse
import csv


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--min-us", type=float, default=150.0)
    a = ap.parse_args()
    rows = []
    with open(a.trace) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]))
    rows.sort()
    adam = [s for s, e, q, n in rows if n.startswith("clip_adam_kernel")]
    t0, t1 = adam[-2], adam[-1]
    win = [r for r in rows if t0 <= r[0] < t1]
    queues = sorted({r[2] for r in win}, key=lambda q: -sum(1 for r in win if r[2] == q))
    qn = {q: ("main" if i == 0 else f"side{i}") for i, q in enumerate(queues)}
    print(f"step {(t1 - t0) / 1e6:.2f} ms, {len(win)} launches; queues: " + ", ".join(f"{qn[q]}={sum(1 for r in win if r[2] == q)}" for q in queues))
    small = 0.0
    for s, e, q, n in win:
        d = (e - s) / 1e3
        if d < a.min_us:
            small += d
            continue
        print(f"{(s - t0) / 1e6:8.3f} ms  {qn[q]:6s} {d:8.1f} us  {n[:100]}")
    print(f"(launches under {a.min_us:.0f} us: {small / 1e3:.2f} ms in total)")


if __name__ == "__main__":
    main()
