"""Diagnostic (GPU): does any kernel of the pBSRNN inference / training path read LDS it has not written?
Stream B loops a kernel that leaves NaN in the LDS of every CU; stream A runs the model.  Outputs are compared with
a quiet run: NaN = an uninitialised-LDS read; small finite differences = a timing race (missing barrier); equal = clean.
Usage: python tools/lds_dirt.py [reps]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import bsrnn_oracle as O  # noqa: E402
from wesep_amd import _lib as L  # noqa: E402
from wesep_amd.models import get_model  # noqa: E402

d = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
kw = dict(num_repeat=1, spk_fuse_type="multiply", multi_fuse=False)
cfg = O.BSRNNConfig(**kw)
model = get_model("BSRNN")(use_spk_transform=False, joint_training=False, **kw)
model.load_state_dict(O.synth_params(cfg, 1))
model.to(d).eval()
wav, tgt, emb = (t.to(d) for t in O.synth_batch(2, 24000, 1))
side = torch.cuda.Stream(device=d)


def fwd():
    with torch.no_grad():
        est, _ = model(wav, emb)
    return est


quiet = fwd()
torch.cuda.synchronize()
for value, tag in ((float("nan"), "NaN"), (1e30, "1e30"), (0.0, "0")):
    nan = bad = 0
    worst = 0.0
    for r in range(reps):
        with torch.cuda.stream(side):
            for _ in range(4):
                L.check(L.lib().ws_debug_dirty_lds(C.c_float(value), 2048, 4000, None, C.c_void_p(side.cuda_stream)), "dirty")
        est = fwd()
        busy = not side.query()            # the dirtier must still be running when the forward has been enqueued
        torch.cuda.synchronize()
        overl = locals().get("overl", 0) + int(busy)
        if not torch.isfinite(est).all():
            nan += 1
        elif not torch.equal(est, quiet):
            bad += 1
            worst = max(worst, float((est - quiet).abs().max() / quiet.abs().max()))
    print(f"dirty LDS value {tag}: {reps} forwards beside the dirtier: {nan} non-finite, {bad} finite mismatches "
          f"(worst {worst:.1e} of the peak); dirtier still running after enqueue in {overl} (cumulative)", flush=True)
