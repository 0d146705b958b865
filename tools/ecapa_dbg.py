import sys, torch
sys.path.insert(0, '/root/repo')
from oracle import ecapa_oracle as EO
from wesep_amd.models.resnet import get_speaker_model
d = torch.device("cuda:0")
def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))
params = EO.synth_params(21)
model = get_speaker_model("ECAPA_TDNN_GLOB_c512")(feat_dim=80, embed_dim=192, pooling_func="ASTP")
model.load_state_dict(params, strict=True); model = model.to(d).train()
g = torch.Generator().manual_seed(22)
x, probe = torch.randn(32, 64, 80, generator=g), torch.randn(32, 192, generator=g)
emb = model(x.to(d)); (emb * probe.to(d)).sum().backward()
p = {k: (v.clone() if EO.is_buffer(k) else v.clone().requires_grad_(True)) for k, v in params.items()}
ref = EO.ecapa_forward(p, x); (ref * probe).sum().backward()
print("fwd", rel(emb, ref))
for k, prm in list(model.named_parameters())[::-1]:
    print(f"{k:46s} {rel(prm.grad, p[k].grad):.2e}")
