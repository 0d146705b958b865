"""CPU: numerics of the TF-GridNet blocked-layout recurrence path (functional_tfgridnet.BlstmLinearBlkFn, opt-in
WESEP_TFGRID_BLOCKED=1) on the blocked-layout emulation (tests/emu_blk.py): against plain torch autograd for the
function itself -- outputs and every gradient, with sequence padding (cluster branch), the 16-sequence branch and the
fused-projection branch -- and the whole recipe-geometry model against its default path."""
import pytest
import torch

from tests import emu_blk, emu_dev


@pytest.fixture
def emu(monkeypatch):
    emu_dev.install(monkeypatch)
    emu_blk.install(monkeypatch)
    monkeypatch.setattr(torch.Tensor, "is_cuda", property(lambda self: True))
    monkeypatch.setenv("WESEP_WGRAD_OVERLAP", "0")


def _reference(y, res, nseq, Lr, lstm, lin):
    out, _ = lstm(y.view(nseq, Lr, 128))
    return res + lin(out.reshape(nseq * Lr, -1))


# storage format of the saved gates / d(gates) (wesep_hip.h WS_GATES_*) -> gradient tolerance on the emulation: exact for the
# fp32 format, ~1e-5 for unorm16 gates, 2^-12 per element of d(gates) for the default's scaled fp16, 2^-9 for bf16 ("h2b")
@pytest.mark.parametrize("fmt,gtol", [("f32", 1e-4), ("h2s", 1e-4), ("h2", 6e-4), ("h2b", 4e-3)])
@pytest.mark.parametrize("nseq,Lr,branch", [(5, 70, "cluster (padded to 64)"), (3, 9, "16-sequence"),
                                            (40, 4, "16-sequence, two tiles"), (4100, 2, "fused projection")])
def test_blstm_linear_blocked_matches_torch(emu, monkeypatch, nseq, Lr, branch, fmt, gtol):
    from wesep_amd import dev
    from wesep_amd import functional_tfgridnet as FG
    monkeypatch.setenv("WESEP_GATES", fmt)
    monkeypatch.setenv("WESEP_TFG_TNB_A16", "1")     # (the fp16 A operand of the weight-gradient GEMMs: opt-in here, h2 only)
    torch.manual_seed(nseq)
    h = 192
    lstm = torch.nn.LSTM(128, h, 1, batch_first=True, bidirectional=True)
    lin = torch.nn.Linear(2 * h, 128)
    y = torch.randn(nseq * Lr, 128, requires_grad=True)
    res = torch.randn(nseq * Lr, 128, requires_grad=True)
    probe = torch.randn(nseq * Lr, 128)
    # which branch the host code takes (the emulation reports 256 CUs)
    ns = nseq + ((-nseq) % 64 if Lr >= 64 else 0)
    seq = dev.SeqMap(ns, dev.BIG, 0, Lr, 1, Lr)
    cluster = dev.lstm_cluster_ok(seq, torch.device("cpu"))
    assert ("cluster" in branch) == cluster and ("fused" in branch) == dev.lstm_fuse_ok(ns, cluster)
    wf, hf, bf = FG.pad_lstm(lstm.weight_ih_l0, lstm.weight_hh_l0, lstm.bias_ih_l0, lstm.bias_hh_l0)
    wr, hr, br = FG.pad_lstm(lstm.weight_ih_l0_reverse, lstm.weight_hh_l0_reverse, lstm.bias_ih_l0_reverse,
                             lstm.bias_hh_l0_reverse)
    out = FG.BlstmLinearBlkFn.apply(y, res, (nseq, Lr), None, None, wf, wr, bf, br, hf, hr, FG.pad_hidden_cols(lin.weight, h),
                                    lin.bias)
    (out * probe).sum().backward()
    got = {"y": y.grad.clone(), "res": res.grad.clone(),
           **{k: p.grad.clone() for k, p in list(lstm.named_parameters()) + [("lin." + k, p) for k, p in lin.named_parameters()]}}
    for t in [y, res] + list(lstm.parameters()) + list(lin.parameters()):
        t.grad = None
    ref = _reference(y, res, nseq, Lr, lstm, lin)
    (ref * probe).sum().backward()
    want = {"y": y.grad, "res": res.grad,
            **{k: p.grad for k, p in list(lstm.named_parameters()) + [("lin." + k, p) for k, p in lin.named_parameters()]}}
    c2 = cluster and fmt != "f32" and dev.lstm_cluster2_on()     # ws_lstm_fwd_cluster2 (round 5): fp16 h, fp16 input copy
    assert float((out - ref).norm() / ref.norm()) < (1e-4 if c2 else 1e-5)
    for k in want:
        assert float((got[k] - want[k]).norm()) <= max(gtol, 6e-4 if c2 else 0.0) * float(want[k].norm()) + 1e-6, k


@pytest.mark.parametrize("B,T,Q", [(2, 70, 5), (1, 9, 3)])     # 10 sequences of 70 steps (cluster, padded to 64 via nvalid) / streaming
def test_blstm_linear_strided_map_equals_transposed_copy(emu, monkeypatch, B, T, Q):
    """The inter-frame path in place: sequences (b, q) over t as a strided row set of the [B, T, Q, C] map (wesep_hip.h
    ws_seqmap with `nvalid`) against the same BLSTM on the transposed, contiguous copy -- output and every gradient."""
    from wesep_amd import functional_tfgridnet as FG
    monkeypatch.setenv("WESEP_GATES", "f32")
    torch.manual_seed(B * 10 + Q)
    h = 192
    lstm = torch.nn.LSTM(128, h, 1, batch_first=True, bidirectional=True)
    lin = torch.nn.Linear(2 * h, 128)
    y = torch.randn(B * T * Q, 128, requires_grad=True)          # rows (b, t, q)
    res = torch.randn(B * T * Q, 128, requires_grad=True)
    probe = torch.randn(B * T * Q, 128)
    def params():       # (the padded forms are part of the autograd graph: once per backward)
        wf, hf, bf = FG.pad_lstm(lstm.weight_ih_l0, lstm.weight_hh_l0, lstm.bias_ih_l0, lstm.bias_hh_l0)
        wr, hr, br = FG.pad_lstm(lstm.weight_ih_l0_reverse, lstm.weight_hh_l0_reverse, lstm.bias_ih_l0_reverse,
                                 lstm.bias_hh_l0_reverse)
        return (wf, wr, bf, br, hf, hr, FG.pad_hidden_cols(lin.weight, h), lin.bias)

    leaves = [y, res] + list(lstm.parameters()) + list(lin.parameters())

    def grads():
        g = [t.grad.clone() for t in leaves]
        for t in leaves:
            t.grad = None
        return g

    out = FG.BlstmLinearBlkFn.apply(y, res, (B * Q, T, Q, T * Q, 1, Q), None, None, *params())
    (out * probe).sum().backward()
    got = grads()
    tr = lambda t: t.view(B, T, Q, 128).transpose(1, 2).reshape(B * Q * T, 128)           # rows (b, q, t)
    ref = FG.BlstmLinearBlkFn.apply(tr(y).contiguous(), tr(res).contiguous(), (B * Q, T), None, None, *params())
    ref = ref.view(B, Q, T, 128).transpose(1, 2).reshape(B * T * Q, 128)
    (ref * probe).sum().backward()
    want = grads()
    assert float((out - ref).norm() / ref.norm()) < 1e-6
    for a, b in zip(got, want):
        assert float((a - b).norm()) <= 1e-5 * float(b.norm()) + 1e-7


def test_recipe_geometry_model_blocked_equals_default(emu, monkeypatch):
    from wesep_amd.models import get_model
    monkeypatch.setenv("WESEP_GATES", "f32")     # the host composition, exactly (the 2-byte formats' numerics: the test above)
    torch.manual_seed(0)
    model = get_model("TFGridNet")(n_fft=128, stride=64, n_layers=1, lstm_hidden_units=192, attn_n_head=4,
                                   attn_approx_qk_dim=512, emb_dim=128, emb_ks=1, emb_hs=1, use_spk_transform=False,
                                   spk_fuse_type="multiply", joint_training=False).train()
    g = torch.Generator().manual_seed(1)
    wav, emb = 0.1 * torch.randn(2, 1280, generator=g), torch.randn(2, 256, generator=g)
    probe = torch.randn(2, 1280, generator=g)
    res = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("WESEP_TFGRID_BLOCKED", flag)
        model.zero_grad(set_to_none=True)
        est, _ = model(wav, emb)
        (est * probe).sum().backward()
        res[flag] = (est.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters()})
    assert float((res["1"][0] - res["0"][0]).norm() / res["0"][0].norm()) < 1e-5
    for k, g0 in res["0"][1].items():
        assert float((res["1"][1][k] - g0).norm()) <= 1e-4 * float(g0.norm()) + 1e-7, k


def test_recipe_geometry_model_deferred_weight_gradients_equal_inline(emu, monkeypatch):
    """Round 4: the BLSTMs' weight gradients computed by deferred jobs on the side stream and delivered through
    functional.WGradCarrierFn (released under the inter-frame BPTTs) -- the host logic on the emulation with stand-in
    streams (tests/emu_streams.py): the same parameter gradients, bit for bit, as the in-line computation; every job
    released, every carrier served."""
    from tests import emu_streams
    from wesep_amd import functional as F0
    from wesep_amd.models import get_model
    emu_streams.install(monkeypatch)
    monkeypatch.setenv("WESEP_GATES", "f32")
    torch.manual_seed(0)
    model = get_model("TFGridNet")(n_fft=128, stride=64, n_layers=2, lstm_hidden_units=192, attn_n_head=4,
                                   attn_approx_qk_dim=512, emb_dim=128, emb_ks=1, emb_hs=1, use_spk_transform=False,
                                   spk_fuse_type="multiply", joint_training=False).train()
    g = torch.Generator().manual_seed(1)
    wav, emb = 0.1 * torch.randn(2, 1280, generator=g), torch.randn(2, 256, generator=g)
    probe = torch.randn(2, 1280, generator=g)
    res, made = {}, []
    real = F0.make_wgrad_carrier
    monkeypatch.setattr(F0, "make_wgrad_carrier", lambda params, blocked=None: made.append(real(params, blocked)) or made[-1])
    for flag in ("0", "1"):
        monkeypatch.setenv("WESEP_WGRAD_OVERLAP", flag)
        model.zero_grad(set_to_none=True)
        del made[:]
        est, _ = model(wav, emb)
        (est * probe).sum().backward()
        assert len(made) == 4 and all((c is not None) == (flag == "1") for c in made)     # 2 blocks x (intra, inter)
        assert not F0._pending(wav.device)                                                   # every job was released
        assert all(c[1].grads is None and c[1].event is not None for c in made if c)        # ... and every box emptied
        res[flag] = (est.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters()})
    assert torch.equal(res["1"][0], res["0"][0])
    for k, g0 in res["0"][1].items():
        assert torch.equal(res["1"][1][k], g0), k

