"""The pair BPTT's cycle stamps INSIDE a whole pBSRNN training step ("the in-step pair kernel was never
stamped").  Same workload as bench.py (R = 32 x 4 s, FiLM multi-fuse); WESEP_PAIR_STAMP=1 selects the stamped build (dbg 2048) of
ws_lstm_bwd_pair for every time-view layer of ONE step after warm-up, with and without the deferred weight-gradient jobs
(WESEP_PROBE_SKIP_WGRAD=1 empties the side stream).  Compare with profiles/r05_recurrence_step_budget.txt (the kernel alone).

    python tools/r06_instep_stamps.py > gpurun_out/r06_instep_pair_stamps.txt
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from wesep_amd import functional as F  # noqa: E402
from wesep_amd.models import get_model  # noqa: E402
from wesep_amd.optim import FusedClipAdam  # noqa: E402
from wesep_amd.utils.losses import parse_loss  # noqa: E402
from wesep_amd.utils.synthetic import synth_batch  # noqa: E402

NAMES = ["loop top", "cell backward done", "past S1", "MFMA loop done", "X: flagged / O: partial in LDS",
         "X: next loads requested", "X: partner's flag seen", "X: gather arrived"]


def main():
    d = torch.device("cuda:0")
    torch.manual_seed(0)
    model = get_model("BSRNN")(**bench.MODEL_KW).to(d).train()
    opt = FusedClipAdam(model.parameters(), lr=bench.LR0, weight_decay=bench.WD, clip_grad=bench.CLIP)
    crit = parse_loss("SISDR")[0]
    wav, tgt, emb = (t.to(d) for t in synth_batch(32, bench.T, 42))

    def step():
        est, _ = model(wav, emb)
        loss = crit(est, tgt).mean()
        opt.zero_grad()
        loss.backward()
        opt.step()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    for skip in ("0", "1"):
        os.environ["WESEP_PROBE_SKIP_WGRAD"] = skip
        step()                                    # (one step in this mode before the stamped one)
        torch.cuda.synchronize()
        os.environ["WESEP_PAIR_STAMP"] = "1"
        del F.PAIR_STAMPS[:]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        step()
        e1.record()
        torch.cuda.synchronize()
        os.environ["WESEP_PAIR_STAMP"] = "0"
        print(f"=== side stream {'EMPTY (WESEP_PROBE_SKIP_WGRAD=1)' if skip == '1' else 'carrying the weight-gradient jobs'}: "
              f"stamped step {e0.elapsed_time(e1):.1f} ms, {len(F.PAIR_STAMPS)} pair launches (backward order: last layer first)")
        walls = []
        for li, (buf, Ls) in enumerate(F.PAIR_STAMPS):
            raw = buf.view(torch.int64).cpu()
            ts = raw[: Ls * 16].view(Ls, 2, 8).double()
            wt = raw[Ls * 16:].view(256, 4)
            wt = wt[wt[:, 0] > 0].double() / 100.0          # microseconds of the 100 MHz wall clock; rows of live workgroups
            t_in = wt[:, 0].min()
            walls.append(f"    launch {li}: {wt.shape[0]} workgroups | entry spread {float(wt[:, 0].max() - t_in):7.1f} us | prologue "
                         f"{float((wt[:, 1] - wt[:, 0]).mean()):6.1f} (max {float((wt[:, 1] - wt[:, 0]).max()):6.1f}) | loop mean "
                         f"{float((wt[:, 2] - wt[:, 1]).mean()):7.1f} min {float((wt[:, 2] - wt[:, 1]).min()):7.1f} max "
                         f"{float((wt[:, 2] - wt[:, 1]).max()):7.1f} | first entry -> last loop end {float(wt[:, 2].max() - t_in):7.1f} | "
                         f"pair 0 loop {float(wt[0, 2] - wt[0, 1]):7.1f}")
            # s_memtime ticks at 100 MHz on gfx950 (profiles/r05_recurrence_step_budget.txt: 16659 ticks = 7.07 us would be
            # 2.36 GHz -- it is the shader clock there); report ticks and the step's share per phase
            span = float(ts[-1, 0, 0] - ts[5, 0, 0]) / (Ls - 6)
            line = [f"launch {li}: {span:7.0f} ticks/step"]
            for role, rn in ((0, "X"), (1, "O")):
                tt = ts[5:-1, role] - ts[5:-1, role, 0:1]
                ks = range(1, 8) if role == 0 else range(1, 5)
                line.append(rn + " " + " ".join(f"{float(tt[:, k].mean()) / span:.2f}" for k in ks))
            print("  " + " | ".join(line))
        print("  wall clock per launch (every workgroup):")
        print("\n".join(walls))
    print("columns (fractions of the step since the loop top): " + "; ".join(NAMES[1:]))


if __name__ == "__main__":
    main()
