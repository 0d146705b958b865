"""Speaker-conditioning blocks with the reference's names and state_dict keys
(wesep/modules/common/speaker.py:26-125, wesep/modules/common/norm.py:84-139), running on
the HIP path.  The tensor being conditioned is the Z-layout activation [R, K, Tf, N]; the
embedding is [R, E].  Algorithmically each fuse type is a [R,E]x[E,N] product plus a
broadcast affine over Z -- the reference's [R, K, Tf, E] expansion is never materialised.
The nn.Linear / nn.Conv1d objects are parameter containers only."""
import torch.nn as nn

from ... import functional as F_


class SpeakerTransform(nn.Module):
    """Conv1d(k=1) E->hid, (hid->hid, Tanh) x (num_layers-2), hid->E   (speaker.py:26-49)."""

    def __init__(self, embed_dim=256, num_layers=3, hid_dim=128):
        super().__init__()
        if num_layers != 3:
            raise NotImplementedError("SpeakerTransform kernels cover the reference default num_layers=3")
        layers = [nn.Conv1d(embed_dim, hid_dim, 1)]
        for _ in range(num_layers - 2):
            layers += [nn.Conv1d(hid_dim, hid_dim, 1), nn.Tanh()]
        layers.append(nn.Conv1d(hid_dim, embed_dim, 1))
        self.transforms = nn.Sequential(*layers)

    def forward(self, x):
        squeeze = x.dim() == 3
        if squeeze:
            x = x.squeeze(-1)
        t = self.transforms
        y = F_.SpkTransformFn.apply(x, t[0].weight, t[0].bias, t[1].weight, t[1].bias, t[3].weight, t[3].bias)
        return y.unsqueeze(-1) if squeeze else y


class LinearLayer(nn.Module):
    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.linear = nn.Linear(in_features, out_features, bias)

    def forward(self, x, dummy=None):
        return F_.LinearFn.apply(x, self.linear.weight, self.linear.bias)


class FiLM(nn.Module):
    """(1 + gamma(e)) * x + beta(e), gamma/beta zero-initialised (norm.py:84-139)."""

    def __init__(self, feat_size, embed_size, num_film_layers=1, layer_norm=False):
        super().__init__()
        if num_film_layers != 1 or layer_norm:
            raise NotImplementedError("FiLM kernels cover num_film_layers=1, layer_norm=False (reference use)")
        self.feat_size, self.embed_size, self.num_film_layers = feat_size, embed_size, num_film_layers
        self.layer_norm = None
        self.gamma_fcs = nn.ModuleList([nn.Linear(embed_size, feat_size)])
        self.beta_fcs = nn.ModuleList([nn.Linear(embed_size, feat_size)])
        self.init_weights()

    def init_weights(self):
        for fc in list(self.gamma_fcs) + list(self.beta_fcs):
            nn.init.zeros_(fc.weight)
            nn.init.zeros_(fc.bias)

    def forward(self, embed, x):
        g = F_.LinearFn.apply(embed, self.gamma_fcs[0].weight, self.gamma_fcs[0].bias)
        b = F_.LinearFn.apply(embed, self.beta_fcs[0].weight, self.beta_fcs[0].bias)
        return F_.AffineFn.apply(x, g, b, 1.0)


class SpeakerFuseLayer(nn.Module):
    def __init__(self, embed_dim=256, feat_dim=512, fuse_type="concat"):
        super().__init__()
        assert fuse_type in ["concat", "additive", "multiply", "FiLM", "None"]
        self.fuse_type = fuse_type
        if fuse_type == "concat":
            self.fc = LinearLayer(embed_dim + feat_dim, feat_dim)
        elif fuse_type in ("additive", "multiply"):
            self.fc = LinearLayer(embed_dim, feat_dim)
        elif fuse_type == "FiLM":
            self.fc = FiLM(feat_dim, embed_dim)
        else:
            raise ValueError("Fuse type not defined.")

    def forward(self, x, embed):
        """x: Z layout [R, K, Tf, N]; embed: [R, E] (or the reference's [R, 1, E, 1])."""
        if embed.dim() == 4:
            embed = embed[:, 0, :, 0]
        embed = embed.contiguous()
        if self.fuse_type == "concat":
            return F_.ConcatFuseFn.apply(x, embed, self.fc.linear.weight, self.fc.linear.bias)
        if self.fuse_type == "additive":
            return F_.AffineFn.apply(x, None, self.fc(embed), 1.0)
        if self.fuse_type == "multiply":
            return F_.AffineFn.apply(x, self.fc(embed), None, 0.0)
        return self.fc(embed, x)
