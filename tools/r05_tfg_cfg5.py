"""GPU probe (round 5): TF-GridNet at BASELINE config 5's geometry (recipe, 2 rows x 6 s) against the CPU oracle -- computed
ONCE -- with ws_lstm_fwd_cluster2 (fp16 h; WESEP_TFG_CLUSTER2) on / off for the inter-frame path, the pair BPTT's fp16
recurrence on (the first version of this probe, profiles/r05_tfg_cfg5_precision_split.txt, ran all four combinations with
the fp16-input cut of cluster2: the pair BPTT's arithmetic moved no gradient).  Prints waveform / loss differences and the ten worst per-tensor gradients."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def main():
    from oracle import bsrnn_oracle as O
    from oracle import tfgridnet_oracle as TG
    from wesep_amd import dev
    from wesep_amd.models import get_model
    from wesep_amd.utils.losses import parse_loss
    torch.set_num_threads(16)      # (the 256-core GPU hosts run torch's CPU kernels ~10x slower on all cores: bench.py cpu_baseline)
    d = torch.device("cuda:0")
    kw = dict(n_fft=128, stride=64, n_layers=6, lstm_hidden_units=192, attn_n_head=4, attn_approx_qk_dim=512, emb_dim=128,
              emb_ks=1, emb_hs=1, use_spk_transform=False, spk_fuse_type="multiply")
    cfg = TG.TFGridNetConfig(**kw)
    params = TG.synth_params(cfg, 31)
    wav, tgt, emb = O.synth_batch(2, 96000, 31)
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    out = TG.tfgridnet_forward(p, cfg, wav, emb)
    ref = out[0] if isinstance(out, (tuple, list)) else out
    loss_o = O.sisdr_loss(ref, tgt)
    loss_o.backward()
    want = {k: v.grad.double() for k, v in p.items()}
    top = max(float(g.norm()) for g in want.values())
    for c2, rf in (("1", "1"), ("0", "1")):
        os.environ["WESEP_TFG_CLUSTER2"], os.environ["WESEP_PAIR_RF"] = c2, rf      # (WESEP_TFG_CLUSTER2 defaults to 1 since)
        model = get_model("TFGridNet")(**kw, joint_training=False)
        model.load_state_dict(params, strict=True)
        model = model.to(d).train()
        dev.bump_weight_epoch()
        est, _ = model(wav.to(d), emb.to(d))
        loss = parse_loss("SISDR")[0](est, tgt.to(d))
        loss.backward()
        torch.cuda.synchronize()
        errs = []
        for k, prm in model.named_parameters():
            wn = float(want[k].norm())
            if wn < 1e-6 * top:
                continue
            errs.append((float((prm.grad.double().cpu() - want[k]).norm()) / wn, k, wn / top))
        errs.sort(reverse=True)
        print(f"cluster2={c2} pair_rf={rf}: est rel {rel(est, ref):.2e}  dloss {abs(loss.item() - loss_o.item()):.2e} dB  "
              f"median grad err {errs[len(errs) // 2][0]:.2e}", flush=True)
        for e, k, w in errs[:6]:
            print(f"      {e:.2e}  {k}  (|g| / largest |g| = {w:.1e})", flush=True)


if __name__ == "__main__":
    main()
