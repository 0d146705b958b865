"""Round 6 diagnosis: bench.py's step loop with per-step wall-clock and allocator counters.  Some bench.py runs of one call came out at
344-850 ms per step with every timed kernel class at its usual duration (gpurun_out/r06_c50_full.json, r06_c51_joint_nn1.json);
this prints what the step loop looks like from the host: time per step (with or without a synchronize per step), bytes
allocated / reserved, the allocator's retry / device-malloc / device-free counters per step.

    python tools/r06_diag_steps.py [--joint] [--sync-each] [--steps 8] [--warmup 3]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=32)
    ap.add_argument("--joint", action="store_true")
    ap.add_argument("--sync-each", action="store_true")
    ap.add_argument("--tag", default="")
    a = ap.parse_args()
    import bench as B
    from wesep_amd.models import get_model
    from wesep_amd.optim import FusedClipAdam
    from wesep_amd.utils.losses import parse_loss
    from wesep_amd.utils.schedulers import ExponentialDecrease
    from wesep_amd.utils.synthetic import synth_batch
    d = torch.device("cuda:0")
    torch.manual_seed(0)
    kw = dict(B.MODEL_KW)
    if a.joint:
        kw.update(joint_training=True, spk_model="ResNet34", spk_feat=True,
                  spk_args=dict(feat_dim=80, embed_dim=256, pooling_func="TSTP", two_emb_layer=False))
    model = get_model("BSRNN")(**kw)
    with torch.no_grad():
        for mod in model.separator.separation:
            if hasattr(mod, "fc") and hasattr(mod.fc, "gamma_fcs"):
                torch.nn.init.normal_(mod.fc.gamma_fcs[0].weight, std=0.02)
                torch.nn.init.normal_(mod.fc.beta_fcs[0].weight, std=0.02)
    model = model.to(d).train()
    opt = FusedClipAdam(model.parameters(), lr=B.LR0, weight_decay=B.WD, clip_grad=B.CLIP)
    sched = ExponentialDecrease(opt, num_epochs=150, epoch_iter=1000, initial_lr=B.LR0, final_lr=B.LR1, warm_up_epoch=0)
    crit = parse_loss("SISDR")[0]
    wav, tgt, emb = (t.to(d) for t in synth_batch(a.rows, B.T, 42))
    if a.joint:
        fb = torch.randn(a.rows, 398, 80, generator=torch.Generator().manual_seed(43))
        emb = (fb - fb.mean(1, keepdim=True)).to(d)

    def step(i):
        sched.step(i)
        est, _ = model(wav, emb)
        loss = crit(est, tgt).mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss

    def counters():
        s = torch.cuda.memory_stats()
        return {k: s.get(k, 0) for k in ("num_alloc_retries", "num_ooms", "num_device_alloc", "num_device_free",
                                          "allocated_bytes.all.peak", "reserved_bytes.all.current", "num_sync_all_streams")}

    for i in range(a.warmup):
        t0 = time.perf_counter()
        step(i)
        torch.cuda.synchronize()
        print(f"[{a.tag}] warm-up step {i}: {(time.perf_counter() - t0) * 1e3:8.1f} ms  {counters()}", flush=True)
    torch.cuda.synchronize()
    t_all = time.perf_counter()
    for i in range(a.steps):
        t0 = time.perf_counter()
        step(a.warmup + i)
        t1 = time.perf_counter()
        if a.sync_each:
            torch.cuda.synchronize()
        t2 = time.perf_counter()
        c = counters()
        print(f"[{a.tag}] step {i}: enqueue {(t1 - t0) * 1e3:8.1f} ms, with sync {(t2 - t0) * 1e3:8.1f} ms  retries {c['num_alloc_retries']} "
              f"dev_alloc {c['num_device_alloc']} dev_free {c['num_device_free']} reserved {c['reserved_bytes.all.current'] / 1e9:.1f} GB "
              f"peak_alloc {c['allocated_bytes.all.peak'] / 1e9:.1f} GB sync_all {c['num_sync_all_streams']}", flush=True)
    torch.cuda.synchronize()
    print(f"[{a.tag}] {a.steps} steps: {(time.perf_counter() - t_all) / a.steps * 1e3:.1f} ms per step; "
          f"free / total device memory {[x / 1e9 for x in torch.cuda.mem_get_info()]}", flush=True)


if __name__ == "__main__":
    main()
