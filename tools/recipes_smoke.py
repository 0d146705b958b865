"""GPU smoke of the shipped recipes' model_args (examples/librimix/tse/v2/confs/{tfgridnet,dpccn}.yaml): one small
forward + SI-SDR + backward each, joint training with the ResNet34 encoder on fbank enrollment.  Prints one line each."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wesep_amd.models import get_model  # noqa: E402
from wesep_amd.utils.losses import parse_loss  # noqa: E402

d = torch.device("cuda:0")
spk = dict(spk_model="ResNet34", spk_feat=True, joint_training=True,
           spk_args=dict(feat_dim=80, embed_dim=256, pooling_func="TSTP", two_emb_layer=False))
torch.manual_seed(0)
crit = parse_loss("SISDR")[0]
for name, kw, T in (("TFGridNet", dict(n_fft=128, stride=64, n_layers=6, lstm_hidden_units=192, attn_n_head=4,
                                       attn_approx_qk_dim=512, emb_dim=128, emb_ks=1, emb_hs=1, use_spk_transform=False,
                                       spk_fuse_type="multiply", **spk), 8000),
                    ("DPCCN", dict(win=512, stride=128, feature_dim=257, tcn_blocks=10, tcn_layers=2, causal=False,
                                   spk_fuse_type="multiply", use_spk_transform=False, **spk), 8192)):
    model = get_model(name)(**kw).to(d).train()
    wav, tgt = torch.randn(2, T, device=d) * 0.1, torch.randn(2, T, device=d) * 0.1
    fbank = torch.randn(2, 100, 80, device=d)
    est, _ = model(wav, fbank)
    loss = crit(est, tgt)
    loss.backward()
    gn = sum(float(p.grad.norm()) ** 2 for p in model.parameters() if p.grad is not None) ** 0.5
    nograd = [k for k, p in model.named_parameters() if p.grad is None]
    print(f"{name}: est {tuple(est.shape)} finite={bool(torch.isfinite(est).all())} loss {loss.item():.3f} dB "
          f"|grad| {gn:.3e} params {sum(p.numel() for p in model.parameters()) / 1e6:.2f} M no-grad {len(nograd)}", flush=True)
