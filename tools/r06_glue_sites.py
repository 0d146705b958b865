"""Round 6: which Python call sites launch the ATen glue of the headline step (fills, copies, small elementwise kernels)?
torch.profiler with stacks over one steady-state step; counts per (op, innermost repo frame).

    python tools/r06_glue_sites.py [--joint]
"""
import argparse
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=32)
    a = ap.parse_args()
    import bench as B
    from wesep_amd.models import get_model
    from wesep_amd.optim import FusedClipAdam
    from wesep_amd.utils.losses import parse_loss
    from wesep_amd.utils.synthetic import synth_batch
    d = torch.device("cuda:0")
    torch.manual_seed(0)
    model = get_model("BSRNN")(**B.MODEL_KW).to(d).train()
    opt = FusedClipAdam(model.parameters(), lr=B.LR0, weight_decay=B.WD, clip_grad=B.CLIP)
    crit = parse_loss("SISDR")[0]
    wav, tgt, emb = (t.to(d) for t in synth_batch(a.rows, B.T, 42))

    def step():
        est, _ = model(wav, emb)
        loss = crit(est, tgt).mean()
        opt.zero_grad()
        loss.backward()
        opt.step()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
        step()
        torch.cuda.synchronize()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    want = ("aten::fill_", "aten::zero_", "aten::copy_", "aten::add", "aten::add_", "aten::mul", "aten::sum", "aten::cat",
            "aten::clone", "aten::contiguous", "aten::div", "aten::sub", "aten::neg", "aten::mean", "aten::ones_like",
            "aten::zeros", "aten::zeros_like", "aten::expand", "aten::_to_copy", "aten::stack")
    sites = collections.Counter()
    for ev in prof.events():
        if ev.name not in want:
            continue
        fr = [s for s in (ev.stack or []) if root in s or "wesep_amd" in s or "bench.py" in s]
        shp = str(ev.input_shapes)[:60] if ev.input_shapes else ""
        sites[(ev.name, fr[0] if fr else ((ev.stack or ["<autograd engine>"])[0]), shp)] += 1
    for (name, site, shp), n in sites.most_common(70):
        print(f"{n:5d}  {name:18s} {site[-110:]:110s} {shp}")


if __name__ == "__main__":
    main()
