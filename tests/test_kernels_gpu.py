"""GPU parity of every C-ABI kernel against plain torch fp32 computed on the CPU.
All calls go through the C ABI (wesep_amd.dev -> libwesep_hip.so)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cuda():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def rnd(g, *shape, scale=1.0):
    return (torch.randn(*shape, generator=g) * scale).float()


# ----------------------------------------------------------------------------------------------
# GEMM NT
# ----------------------------------------------------------------------------------------------
MODES = ["f32", "bf16x3"]
TOL = {"f32": 3e-6, "bf16x3": 4e-5}   # bf16x3 drops the lo*lo term: <= 2^-16 per product


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("M,N,K,vec", [(300, 200, 70, 0), (300, 200, 72, 3), (128, 128, 32, 3),
                                       (1000, 2048, 128, 3), (77, 12, 512, 3), (513, 128, 6, 0)])
def test_gemm_nt_plain(M, N, K, vec, mode):
    from wesep_amd import dev
    d = _cuda()
    g = torch.Generator().manual_seed(M + N + K)
    A, W, b = rnd(g, M, K), rnd(g, N, K), rnd(g, N)
    C = torch.full((M, N), float("nan"), device=d)
    dev.gemm_nt(A=A.to(d), a_rows=dev.flat(K), M=M, N=N, K=K, W=W.to(d), ldw=K, bias=b.to(d), C_out=C,
                c_rows=dev.flat(N), vec=vec, mode=mode)
    ref = A.double() @ W.double().t() + b.double()
    assert rel(C, ref) < TOL[mode]


@pytest.mark.parametrize("mode", MODES)
def test_gemm_nt_epilogues_and_norm(mode):
    from wesep_amd import dev
    d = _cuda()
    g = torch.Generator().manual_seed(5)
    M, N, K, S = 260, 136, 128, 13   # rows grouped in S stat groups of 20
    A, W, b = rnd(g, M, K), rnd(g, N, K, scale=0.1), rnd(g, N)
    Rr, Tt = rnd(g, M, N), torch.tanh(rnd(g, M, N))
    stats = torch.stack([rnd(g, S), rnd(g, S).abs() + 0.5], 1).contiguous()
    gamma, beta = rnd(g, K), rnd(g, K)
    C = torch.empty(M, N, device=d)
    dev.gemm_nt(A=A.to(d), a_rows=dev.flat(K), M=M, N=N, K=K, W=W.to(d), ldw=K, bias=b.to(d), C_out=C,
                c_rows=dev.flat(N), R=Rr.to(d), T=Tt.to(d), stats=stats.to(d), gamma=gamma.to(d),
                beta=beta.to(d), stat_map=dev.StatMap(20, 1, 1, 0, 0), act=1, mode=mode)
    s = torch.arange(M) // 20
    An = (A - stats[s, 0:1]) * stats[s, 1:2] * gamma + beta
    ref = torch.tanh(An @ W.t() + b) * (1 - Tt * Tt) + Rr
    assert rel(C, ref) < TOL[mode]


@pytest.mark.parametrize("mode", MODES)
def test_gemm_nt_two_level_rows_and_groups(mode):
    """Grouped launch over 3 'bands' with ragged N/K, 2-level A rows and strided C columns."""
    from wesep_amd import dev, _lib as L
    d = _cuda()
    g = torch.Generator().manual_seed(9)
    Rb, Kb, Tf, N = 3, 3, 20, 128          # Z layout [Rb, Kb, Tf, N]
    M = Rb * Tf
    Z = rnd(g, Rb, Kb, Tf, N)
    widths = [12, 64, 24]
    offs = [0, 12, 76]
    Ws = [rnd(g, w, N, scale=0.1) for w in widths]
    bs = [rnd(g, w) for w in widths]
    Wd, bd = [w.to(d) for w in Ws], [b.to(d) for b in bs]
    C = torch.zeros(M, 100, device=d)
    desc = np.zeros(3, dtype=L.GROUP_NT_DTYPE)
    for k in range(3):
        desc[k] = (Wd[k].data_ptr(), bd[k].data_ptr(), 0, 0, k * Tf * N, offs[k], 0, N, widths[k], N, 0)
    gd = L.upload_struct_array(desc, d)
    dev.gemm_nt(A=Z.to(d), a_rows=dev.Rows(Tf, Kb * Tf * N, N), M=M, C_out=C, c_rows=dev.flat(100),
                groups=gd, ngroups=3, max_n=64, mode=mode)
    ref = torch.zeros(M, 100)
    for k in range(3):
        Ak = Z[:, k].reshape(M, N)
        ref[:, offs[k]:offs[k] + widths[k]] = Ak @ Ws[k].t() + bs[k]
    assert rel(C, ref) < TOL[mode]


# ----------------------------------------------------------------------------------------------
# GEMM TN
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("M,Nn,Kk,vec", [(5000, 200, 72, 1), (300, 12, 512, 1), (4100, 128, 6, 0),
                                         (9000, 1024, 256, 1)])
def test_gemm_tn_plain_and_bias(M, Nn, Kk, vec, mode):
    from wesep_amd import dev
    d = _cuda()
    g = torch.Generator().manual_seed(M + Nn)
    G, A = rnd(g, M, Nn), rnd(g, M, Kk)
    nsplit, rps = dev.tn_splits(M)
    slab = torch.full((nsplit, Nn * Kk), float("nan"), device=d)
    bslab = torch.full((nsplit, Nn), float("nan"), device=d)
    dev.gemm_tn(G=G.to(d), g_rows=dev.flat(Nn), A=A.to(d), a_rows=dev.flat(Kk), M=M, Nn=Nn, Kk=Kk,
                slab=slab, slab_stride=Nn * Kk, bslab=bslab, bslab_stride=Nn, nsplit=nsplit,
                rows_per_split=rps, vec=vec, mode=mode)
    out = torch.empty(Nn, Kk, device=d)
    bo = torch.empty(Nn, device=d)
    dev.reduce_slabs(slab, nsplit, Nn * Kk, Nn * Kk, out)
    dev.reduce_slabs(bslab, nsplit, Nn, Nn, bo)
    assert rel(out, G.double().t() @ A.double()) < TOL[mode]
    assert rel(bo, G.double().sum(0)) < 5e-6


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("seq_div,seq_len,sign", [(1, 25, -1), (1, 25, 1), (10, 4, -1), (10, 4, 1)])
def test_gemm_tn_shift_and_norm(seq_div, seq_len, sign, mode):
    from wesep_amd import dev
    d = _cuda()
    g = torch.Generator().manual_seed(17)
    M, Nn, Kk = 400, 64, 128
    G, A = rnd(g, M, Nn), rnd(g, M, Kk)
    shift = sign * seq_div
    slab = torch.empty(1, Nn * Kk, device=d)
    dev.gemm_tn(G=G.to(d), g_rows=dev.flat(Nn), A=A.to(d), a_rows=dev.flat(Kk), M=M, Nn=Nn, Kk=Kk,
                slab=slab, slab_stride=Nn * Kk, nsplit=1, rows_per_split=416, shift_rows=shift,
                seq_div=seq_div, seq_len=seq_len, mode=mode)
    m = torch.arange(M)
    t = (m // seq_div) % seq_len
    ok = ((t + sign) >= 0) & ((t + sign) < seq_len)
    As = torch.zeros_like(A)
    idx = m[ok] + shift
    As[ok] = A[idx]
    assert rel(slab.view(Nn, Kk), G.double().t() @ As.double()) < TOL[mode]
    # norm prologue
    S = 20
    stats = torch.stack([rnd(g, S), rnd(g, S).abs() + 0.5], 1).contiguous()
    gamma, beta = rnd(g, Kk), rnd(g, Kk)
    dev.gemm_tn(G=G.to(d), g_rows=dev.flat(Nn), A=A.to(d), a_rows=dev.flat(Kk), M=M, Nn=Nn, Kk=Kk,
                slab=slab, slab_stride=Nn * Kk, nsplit=1, rows_per_split=416, stats=stats.to(d),
                gamma=gamma.to(d), beta=beta.to(d), stat_map=dev.StatMap(20, 1, 1, 0, 0), mode=mode)
    s = m // 20
    An = (A - stats[s, 0:1]) * stats[s, 1:2] * gamma + beta
    assert rel(slab.view(Nn, Kk), G.double().t() @ An.double()) < TOL[mode]


def test_transpose_and_reduce_ld():
    from wesep_amd import dev
    d = _cuda()
    g = torch.Generator().manual_seed(3)
    src = rnd(g, 70, 384)
    dst = torch.empty(100, 70, device=d)
    dev.transpose(src.to(d), 70, 100, 384, dst, src_off=128)
    assert torch.equal(dst.cpu(), src[:, 128:228].t().contiguous())
    slab = rnd(g, 3, 50)
    out = torch.zeros(5, 24, device=d)
    dev.reduce_slabs(slab.to(d), 3, 50, 50, out, w=10, ldo=24, out_off=4)
    ref = torch.zeros(5, 24)
    ref[:, 4:14] = slab.sum(0).view(5, 10)
    assert rel(out, ref) < 1e-6
    # the split-lane kernels (4 lanes from 16 splits, 16 lanes from 64 splits: round 6) and the one-workgroup-per-output form:
    # odd split counts, counts that do not fill the last 64-column block, deterministic
    for nsplit, count in ((16, 100), (37, 11520), (64, 777), (342, 11520), (341, 6912), (200, 5)):
        slab = rnd(g, nsplit, count).to(d)
        o1, o2 = torch.full((count,), float("nan"), device=d), torch.full((count,), float("nan"), device=d)
        dev.reduce_slabs(slab, nsplit, count, count, o1)
        dev.reduce_slabs(slab, nsplit, count, count, o2)
        assert torch.equal(o1, o2)
        assert rel(o1, slab.double().sum(0)) < 1e-6, (nsplit, count)


# ----------------------------------------------------------------------------------------------
# GroupNorm pieces
# ----------------------------------------------------------------------------------------------
def _geoms(R, K, Tf, N, d):
    from wesep_amd import dev
    time = dev.Geom(R * K, 1, Tf * N, 0, N, Tf, N)
    band = dev.Geom(R * Tf, Tf, K * Tf * N, N, Tf * N, K, N)
    return time, band


@pytest.mark.parametrize("view", ["time", "band"])
def test_group_stats_and_gn_backward(view):
    from wesep_amd import dev
    d = _cuda()
    g = torch.Generator().manual_seed(21)
    R, K, Tf, N = 2, 5, 9, 128
    z = rnd(g, R, K, Tf, N) + 0.3
    dxn = rnd(g, R, K, Tf, N)
    gamma = rnd(g, N) + 1.0
    res = rnd(g, R, K, Tf, N)
    geo = _geoms(R, K, Tf, N, d)[0 if view == "time" else 1]
    # torch reference: groups as [G, N, L]
    if view == "time":
        x3 = z.reshape(R * K, Tf, N).transpose(1, 2)
        d3 = dxn.reshape(R * K, Tf, N).transpose(1, 2)
    else:
        x3 = z.permute(0, 2, 3, 1).reshape(R * Tf, N, K)
        d3 = dxn.permute(0, 2, 3, 1).reshape(R * Tf, N, K)
    x3 = x3.clone().requires_grad_(True)
    gam = gamma.clone().requires_grad_(True)
    bet = torch.zeros(N, requires_grad=True)
    eps = float(np.finfo(np.float32).eps)
    y = torch.nn.functional.group_norm(x3, 1, gam, bet, eps)
    y.backward(d3)
    mean = x3.detach().mean((1, 2))
    rstd = 1 / torch.sqrt(x3.detach().var((1, 2), unbiased=False) + eps)
    stats = torch.empty(geo.ngroups, 2, device=d)
    zd, dd = z.to(d), dxn.to(d)
    dev.group_stats(zd, geo, stats)
    assert rel(stats[:, 0], mean) < 1e-5 and rel(stats[:, 1], rstd) < 1e-5
    ab = torch.empty(geo.ngroups, 2, device=d)
    dev.gn_bwd_reduce(zd, dd, stats, geo, ab, gamma=gamma.to(d))
    dz = torch.empty_like(zd)
    dev.gn_bwd_apply(zd, dd, stats, ab, geo, dz, gamma=gamma.to(d), res=res.to(d))
    if view == "time":
        gref = x3.grad.transpose(1, 2).reshape(R, K, Tf, N)
    else:
        gref = x3.grad.reshape(R, Tf, N, K).permute(0, 3, 1, 2)
    assert rel(dz, gref + res) < 1e-5
    ns = 3
    slab = torch.empty(ns, 1, 2, N, device=d)
    dev.gn_param_grad(zd, dd, stats, geo, ns, slab)
    out = slab.sum(0)[0]
    assert rel(out[0], gam.grad) < 1e-5 and rel(out[1], bet.grad) < 1e-5
    # pass 2 WITH the parameter sums, the last workgroup adding up the per-group partials (ABI v15): same dx bits as the
    # plain apply pass, (dgamma, dbeta) without a reduction launch; the counter word comes back as zero (used twice)
    assert dev.gn_bwd_apply_pg_ok(geo)
    counter = torch.zeros(1 + dev.tree_groups(geo.ngroups), device=d, dtype=torch.int32)
    pouts = []
    for _ in range(2):
        dz2 = torch.full_like(zd, float("nan"))
        pslab = torch.full((geo.ngroups + dev.tree_groups(geo.ngroups), 2, N), float("nan"), device=d)
        pout = torch.full((2, N), float("nan"), device=d)
        dev.gn_bwd_apply_pg(zd, dd, stats, ab, geo, dz2, gamma.to(d), pslab, pout, counter, res=res.to(d))
        assert torch.equal(dz2, dz)
        assert rel(pout[0], gam.grad) < 1e-5 and rel(pout[1], bet.grad) < 1e-5
        pouts.append(pout)
    assert torch.equal(pouts[0], pouts[1]) and int(counter.abs().sum().item()) == 0


@pytest.mark.parametrize("R,K,Tf", [(2, 32, 37), (3, 6, 11), (1, 2, 300)])
def test_gn_backward_fused_small_groups(R, K, Tf):
    """norm.hip gn_bwd_fused_kernel (one wave per band-view group: reduce + apply + dgamma / dbeta in one pass) against
    torch's GroupNorm backward and against the three-kernel form."""
    from wesep_amd import dev
    d = _cuda()
    g = torch.Generator().manual_seed(R * 100 + K)
    N = 128
    z = rnd(g, R, K, Tf, N) + 0.3
    dxn = rnd(g, R, K, Tf, N)
    gamma = rnd(g, N) + 1.0
    res = rnd(g, R, K, Tf, N)
    geo = _geoms(R, K, Tf, N, d)[1]
    assert dev.gn_bwd_fused_ok(geo)
    x3 = z.permute(0, 2, 3, 1).reshape(R * Tf, N, K).clone().requires_grad_(True)
    d3 = dxn.permute(0, 2, 3, 1).reshape(R * Tf, N, K)
    gam = gamma.clone().requires_grad_(True)
    bet = torch.zeros(N, requires_grad=True)
    eps = float(np.finfo(np.float32).eps)
    torch.nn.functional.group_norm(x3, 1, gam, bet, eps).backward(d3)
    gref = x3.grad.reshape(R, Tf, N, K).permute(0, 3, 1, 2) + res
    zd, dd = z.to(d), dxn.to(d)
    stats = torch.empty(geo.ngroups, 2, device=d)
    dev.group_stats(zd, geo, stats)
    outs = []
    for nwg in (1, 7, min(1024, -(-geo.ngroups // 4))):
        dz = torch.full_like(zd, float("nan"))
        pslab = torch.full((nwg, 2, N), float("nan"), device=d)
        dev.gn_bwd_fused(zd, dd, stats, geo, gamma.to(d), dz, nwg, pslab, res=res.to(d))
        assert rel(dz, gref) < 1e-5, nwg
        out = pslab.sum(0)
        assert rel(out[0], gam.grad) < 1e-5 and rel(out[1], bet.grad) < 1e-5, nwg
        outs.append(dz)
    assert torch.equal(outs[0], outs[2])                      # dx does not depend on the launch geometry
    ab = torch.empty(geo.ngroups, 2, device=d)
    dev.gn_bwd_reduce(zd, dd, stats, geo, ab, gamma=gamma.to(d))
    dz3 = torch.empty_like(zd)
    dev.gn_bwd_apply(zd, dd, stats, ab, geo, dz3, gamma=gamma.to(d), res=res.to(d))
    assert rel(outs[0], dz3) < 1e-6
    dzn = torch.empty_like(zd)                                # without the residual term
    dev.gn_bwd_fused(zd, dd, stats, geo, gamma.to(d), dzn, 5, torch.empty(5, 2, N, device=d))
    assert rel(dzn, gref - res) < 1e-5
    # the last workgroup adds the per-workgroup shares up (pout; ABI v15): the sum of pslab, in workgroup order
    for nwg in (1, 7, 40, min(1024, -(-geo.ngroups // 4))):
        counter = torch.zeros(1 + dev.tree_groups(nwg), device=d, dtype=torch.int32)
        pouts = []
        for _ in range(2):
            pslab = torch.full((nwg + dev.tree_groups(nwg), 2, N), float("nan"), device=d)
            pout = torch.full((2, N), float("nan"), device=d)
            dzp = torch.empty_like(zd)
            dev.gn_bwd_fused(zd, dd, stats, geo, gamma.to(d), dzp, nwg, pslab, res=res.to(d), pout=pout, counter=counter)
            assert torch.equal(dzp, outs[0])
            assert rel(pout, pslab[:nwg].double().sum(0)) < 1e-6
            assert rel(pout[0], gam.grad) < 1e-5 and rel(pout[1], bet.grad) < 1e-5, nwg
            pouts.append(pout)
        assert torch.equal(pouts[0], pouts[1]) and int(counter.abs().sum().item()) == 0


# ----------------------------------------------------------------------------------------------
# LSTM recurrence
# ----------------------------------------------------------------------------------------------
LSTM_TOL = {1: (1e-5, 2e-5), 2: (1e-5, 2e-5), 3: (4e-5, 8e-5), 4: (4e-5, 8e-5), 5: (4e-5, 8e-5)}   # 3/4 = split-bf16 (drops lo*lo)


@pytest.mark.parametrize("dims", [(2, 5, 11), (3, 7, 37)])
@pytest.mark.parametrize("view,mt", [("time", 1), ("band", 1), ("time", 2), ("band", 2), ("time", 3), ("band", 3),
                                     ("time", 4), ("band", 4), ("time", 5), ("band", 5)])
def test_lstm_fwd_bwd_vs_torch(view, mt, dims):
    from wesep_amd import dev, _lib as L
    from wesep_amd.functional import _view_maps
    d = _cuda()
    g = torch.Generator().manual_seed(31)
    (R, K, Tf), N, H = dims, 128, 256
    P = R * K * Tf
    tol_f, tol_b = LSTM_TOL[mt]
    geo, smap, seq, _ = _view_maps(view, R, K, Tf, N)
    lstm = torch.nn.LSTM(N, H, 1, batch_first=True, bidirectional=True)
    x = rnd(g, R, K, Tf, N)
    if view == "time":
        xs = x.reshape(R * K, Tf, N)
    else:
        xs = x.permute(0, 2, 1, 3).reshape(R * Tf, K, N)
    xs = xs.clone().requires_grad_(True)
    out, _ = lstm(xs)
    dout_seq = rnd(g, *out.shape)
    out.backward(dout_seq)
    # gates_x = x W_ih^T + b_ih + b_hh computed on the CPU, laid out [P][2][4H]
    with torch.no_grad():
        gx = []
        for sfx in ("", "_reverse"):
            w, bi, bh = (getattr(lstm, n + sfx) for n in ("weight_ih_l0", "bias_ih_l0", "bias_hh_l0"))
            gx.append(x.reshape(P, N) @ w.t() + bi + bh)
        gates = torch.stack(gx, 1).contiguous().to(d)           # [P, 2, 4H]
    pf, pb = torch.empty(L.LSTM_PACK_FLOATS, device=d), torch.empty(L.LSTM_PACK_FLOATS, device=d)
    dev.lstm_pack(lstm.weight_hh_l0.detach().to(d).contiguous(),
                  lstm.weight_hh_l0_reverse.detach().to(d).contiguous(), pf, pb, mt)
    cbuf, hcat = torch.zeros(P, 2 * H, device=d), torch.zeros(P, 2 * H, device=d)
    blocked = mt >= 4            # modes 4/5: the blocked layout BL (32- / 16-sequence workgroups)
    if blocked:
        gates = dev.to_blocked(gates.view(P, 8 * H), seq)
        nb = dev.bl_num_blocks(seq)
        cbuf, hcat = torch.zeros(nb, 2 * H // 4, 32, 4, device=d), torch.zeros(nb, 2 * H // 4, 32, 4, device=d)
    dev.lstm_fwd(gates, cbuf, hcat, pf, seq, mt)
    if blocked:
        hcat_b, hcat = hcat, dev.from_blocked(hcat, seq, P, split=True)   # h leaves the blocked kernels as BLS
    if view == "time":
        href = out.detach().reshape(R, K, Tf, 2 * H)
        dref = dout_seq.reshape(R, K, Tf, 2 * H)
    else:
        href = out.detach().reshape(R, Tf, K, 2 * H).permute(0, 2, 1, 3)
        dref = dout_seq.reshape(R, Tf, K, 2 * H).permute(0, 2, 1, 3)
    assert rel(hcat.view(R, K, Tf, 2 * H), href) < tol_f
    dh = dref.contiguous().reshape(P, 2 * H).to(d)
    if blocked:
        dev.lstm_bwd(gates, cbuf, hcat_b, dev.to_blocked(dh, seq), pb, seq, mt)
        gates = dev.from_blocked(gates, seq, P, split=True).view(P, 2, 4 * H)   # d(gates): BLS
    else:
        dev.lstm_bwd(gates, cbuf, hcat, dh, pb, seq, mt)
    # d gates_x -> dx = dgates @ W_ih (both dirs), dW_hh via autograd comparisons
    dg = gates.cpu()
    dx = dg[:, 0] @ lstm.weight_ih_l0.detach() + dg[:, 1] @ lstm.weight_ih_l0_reverse.detach()
    if view == "time":
        dxref = xs.grad.reshape(R, K, Tf, N)
    else:
        dxref = xs.grad.reshape(R, Tf, K, N).permute(0, 2, 1, 3)
    assert rel(dx.view(R, K, Tf, N), dxref) < tol_b
    assert rel(dg[:, 0].sum(0), lstm.bias_ih_l0.grad) < tol_b
    assert rel(dg[:, 1].sum(0), lstm.bias_ih_l0_reverse.grad) < tol_b


# ----------------------------------------------------------------------------------------------
# STFT / iSTFT
# ----------------------------------------------------------------------------------------------
def _bands(d):
    from wesep_amd import dev
    from oracle.bsrnn_oracle import band_widths
    bw = band_widths(16000, 512)
    return bw, dev.BandTables(bw, d)


@pytest.mark.parametrize("T", [4000, 3000, 777])
def test_stft_bandsplit_vs_torch(T):
    from wesep_amd import dev
    d = _cuda()
    g = torch.Generator().manual_seed(T)
    R = 3
    wav = rnd(g, R, T, scale=0.1)
    bw, bt = _bands(d)
    Tf = 1 + T // 128
    xbs = torch.full((R * Tf, 514), float("nan"), device=d)
    dev.stft_bandsplit(wav.to(d), bt, xbs)
    spec = torch.stft(wav, 512, 128, window=torch.hann_window(512), return_complex=True)  # [R,257,Tf]
    ref = torch.empty(R, Tf, 514)
    f0 = 0
    for b in bw:
        ref[:, :, 2 * f0:2 * f0 + b] = spec.real[:, f0:f0 + b].transpose(1, 2)
        ref[:, :, 2 * f0 + b:2 * f0 + 2 * b] = spec.imag[:, f0:f0 + b].transpose(1, 2)
        f0 += b
    assert rel(xbs.view(R, Tf, 514), ref) < 2e-6


@pytest.mark.parametrize("T", [4000, 3000])
def test_mask_istft_fwd_bwd_vs_torch(T):
    from wesep_amd import dev
    d = _cuda()
    g = torch.Generator().manual_seed(T + 1)
    R = 2
    bw, bt = _bands(d)
    Tf = 1 + T // 128
    wav = rnd(g, R, T, scale=0.1)
    spec = torch.stft(wav, 512, 128, window=torch.hann_window(512), return_complex=True)
    m3 = rnd(g, R * Tf, 1028).requires_grad_(True)
    # torch reference of mask apply + istft
    est_bands, f0 = [], 0
    m3v = m3.view(R, Tf, 1028)
    for b in bw:
        o = m3v[:, :, 4 * f0:4 * f0 + 4 * b].reshape(R, Tf, 2, 2, b).permute(0, 2, 3, 4, 1)  # R,2,2,b,Tf
        m = o[:, 0] * torch.sigmoid(o[:, 1])
        xb = spec[:, f0:f0 + b]
        est_bands.append(torch.complex(xb.real * m[:, 0] - xb.imag * m[:, 1],
                                       xb.real * m[:, 1] + xb.imag * m[:, 0]))
        f0 += b
    est_ref = torch.istft(torch.cat(est_bands, 1), 512, 128, window=torch.hann_window(512), length=T)
    dwav = rnd(g, R, T)
    est_ref.backward(dwav)
    xbs = torch.empty(R * Tf, 514, device=d)
    dev.stft_bandsplit(wav.to(d), bt, xbs)
    m3d = m3.detach().to(d)
    frames = torch.empty(R * Tf, 512, device=d)
    dev.mask_istft_frames(xbs, m3d, R, Tf, bt, frames)
    est = torch.empty(R, T, device=d)
    dev.istft_ola(frames, R, Tf, T, est)
    assert rel(est, est_ref) < 5e-6
    dm3 = torch.full((R * Tf, 1028), float("nan"), device=d)
    dev.mask_istft_bwd(dwav.to(d), xbs, m3d, R, Tf, T, bt, dm3)
    assert rel(dm3, m3.grad) < 5e-6


# ----------------------------------------------------------------------------------------------
# elementwise: affine, SI-SDR, clip + Adam
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N", [128, 100])   # full and partial column quads of the 16-byte backward (ws_affine_fwd needs N % 4 == 0)
@pytest.mark.parametrize("K", [4, 32])      # 1 split / 8 splits of the rows (the last workgroup adds the splits up)
def test_affine_fwd_bwd(K, N):
    from wesep_amd.functional import AffineFn
    d = _cuda()
    g = torch.Generator().manual_seed(41)
    R, Tf = 3, 71
    z, a, b, go = rnd(g, R, K, Tf, N), rnd(g, R, N), rnd(g, R, N), rnd(g, R, K, Tf, N)
    for a0, use_a, use_b in ((0.0, True, False), (1.0, False, True), (1.0, True, True)):
        zc, ac, bc = (t.clone().requires_grad_(True) for t in (z, a, b))
        ref = zc * (a0 + (ac[:, None, None, :] if use_a else 0)) + (bc[:, None, None, :] if use_b else 0)
        ref.backward(go)
        zd = z.to(d).requires_grad_(True)
        ad = a.to(d).requires_grad_(True) if use_a else None
        bd = b.to(d).requires_grad_(True) if use_b else None
        out = AffineFn.apply(zd, ad, bd, a0)
        out.backward(go.to(d))
        assert rel(out, ref) < 1e-6 and rel(zd.grad, zc.grad) < 1e-6
        if use_a:
            assert rel(ad.grad, ac.grad) < 1e-5
        if use_b:
            assert rel(bd.grad, bc.grad) < 1e-5


@pytest.mark.parametrize("snr_db", [-5.0, 10.0, 30.0])
def test_sisdr_fwd_bwd(snr_db):
    from oracle import bsrnn_oracle as O
    from wesep_amd.functional import SISDRFn
    d = _cuda()
    g = torch.Generator().manual_seed(7)
    R, T = 4, 64000
    t = rnd(g, R, T, scale=0.1) + 0.01
    x = (0.7 * t + rnd(g, R, T, scale=0.1 * 10 ** (-snr_db / 20)) + 0.02).requires_grad_(True)
    ref = O.sisdr_loss(x, t)
    ref.backward()
    xd = x.detach().to(d).requires_grad_(True)
    loss = SISDRFn.apply(xd, t.to(d), 1e-8)
    loss.backward()
    assert abs(loss.item() - ref.item()) < 1e-2       # dB (north_star tolerance)
    assert abs(loss.item() - ref.item()) < 2e-4
    assert rel(xd.grad, x.grad) < 1e-4


def test_clip_adam_matches_reference_semantics():
    from oracle import bsrnn_oracle as O
    from wesep_amd.optim import FusedClipAdam
    d = _cuda()
    g = torch.Generator().manual_seed(11)
    shapes = [(1024, 128), (7,), (300, 5), (128,)]
    ps = [rnd(g, *s) for s in shapes]
    pd = [torch.nn.Parameter(p.clone().to(d)) for p in ps]
    opt = FusedClipAdam(pd, lr=1e-3, weight_decay=1e-4, clip_grad=5.0)
    pc = [p.clone() for p in ps]
    ms, vs = [torch.zeros_like(p) for p in ps], [torch.zeros_like(p) for p in ps]
    for step in range(1, 4):
        gs = [rnd(g, *s, scale=(3.0 if i == 0 else 0.1)) for i, s in enumerate(shapes)]
        for p, gr in zip(pd, gs):
            p.grad = gr.clone().to(d)
        lr = 1e-3 * (0.9 ** step)
        for grp in opt.param_groups:
            grp["lr"] = lr
        opt.step()
        norms = opt.last_grad_norms()
        gdict = {i: gr.clone() for i, gr in enumerate(gs)}
        nref = O.clip_gradients_(gdict, 5.0)
        for i in range(len(ps)):
            assert abs(norms[i] - nref[i]) < 1e-4 * max(1.0, nref[i])
            O.adam_l2_step_(pc[i], gdict[i], ms[i], vs[i], step, lr, weight_decay=1e-4)
            assert rel(pd[i], pc[i]) < 1e-6, (step, i)
            assert rel(pd[i].grad, gdict[i]) < 1e-6   # clipped in place like funcs.py:86-87


# ----------------------------------------------------------------------------------------------
# GEMMs between the plain Z layout and the blocked layout BL (gemm_blk.hip)
# ----------------------------------------------------------------------------------------------
BLK_DIMS = [(2, 5, 11), (3, 32, 37)]   # (R, K, Tf): padded tiles in both views / full tiles in the time view


def _blk_setup(view, dims):
    from wesep_amd.functional import _view_maps
    R, K, Tf = dims
    N = 128
    geo, smap, seq, _ = _view_maps(view, R, K, Tf, N)
    return R, K, Tf, N, R * K * Tf, geo, smap, seq


@pytest.mark.parametrize("dims", BLK_DIMS)
@pytest.mark.parametrize("view", ["time", "band"])
@pytest.mark.parametrize("norm", [False, True])
def test_gemm_p2b_vs_torch(view, dims, norm):
    from wesep_amd import dev
    d = _cuda()
    R, K, Tf, N, P, geo, smap, seq = _blk_setup(view, dims)
    g = torch.Generator().manual_seed(3)
    Nout = 192
    A, W, b = rnd(g, P, N), rnd(g, Nout, N, scale=0.1), rnd(g, Nout)
    kw, An = {}, A
    if norm:
        stats = torch.stack([rnd(g, geo.ngroups) * 0.1, rnd(g, geo.ngroups).abs() + 0.5], 1).contiguous()
        gamma, beta = rnd(g, N), rnd(g, N)
        kw = dict(stats=stats.to(d), gamma=gamma.to(d), beta=beta.to(d), stat_map=smap)
        m = torch.arange(P)
        sidx = (m // smap.div1) * smap.m1 + (m % smap.div2) * smap.m2 + smap.base
        An = (A - stats[sidx, 0:1]) * stats[sidx, 1:2] * gamma + beta
    wp = torch.empty(Nout * N, device=d)
    dev.pack_w(W.to(d), Nout, N, N, wp, order=0)
    nb = dev.bl_num_blocks(seq)
    outs = []
    for _ in range(2):
        C = torch.full((nb, 32 * Nout), float("nan"), device=d)
        Ab = torch.full((nb, 32 * N), float("nan"), device=d)
        dev.gemm_p2b(A=A.to(d), lda=N, sm=seq, Wpack=wp, N=Nout, C_out=C, bias=b.to(d), A_bl=Ab, **kw)
        outs.append((C, Ab))
    assert torch.equal(outs[0][0], outs[1][0])                      # deterministic
    C, Ab = outs[0]
    assert not torch.isnan(C).any() and not torch.isnan(Ab).any()   # padded slots are written (zeros)
    ref = An.double() @ W.double().t() + b.double()
    assert rel(dev.from_blocked(C.view(nb, Nout // 4, 32, 4), seq, P), ref) < 4e-5
    assert rel(dev.from_blocked(Ab.view(nb, N // 4, 32, 4), seq, P, split=True), An) < 1e-5   # BLS: 2^-17
    assert torch.equal(C.view(nb, Nout // 4, 32, 4), dev.to_blocked(dev.from_blocked(C.view(nb, Nout // 4, 32, 4), seq, P), seq))


@pytest.mark.parametrize("dims", BLK_DIMS)
@pytest.mark.parametrize("view", ["time", "band"])
@pytest.mark.parametrize("Kd", [512, 2048])
def test_gemm_b2p_vs_torch(view, dims, Kd):
    from wesep_amd import dev
    d = _cuda()
    R, K, Tf, N, P, geo, smap, seq = _blk_setup(view, dims)
    g = torch.Generator().manual_seed(4)
    A, W, b, Rr = rnd(g, P, Kd), rnd(g, N, Kd, scale=0.05), rnd(g, N), rnd(g, P, N)
    wp = torch.empty(N * Kd, device=d)
    dev.pack_w(W.to(d), N, Kd, Kd, wp, order=1)
    Ab = dev.to_blocked(A.to(d), seq, split=True)
    outs = []
    for _ in range(2):
        C = torch.full((P, N), float("nan"), device=d)
        dev.gemm_b2p(A=Ab, K=Kd, sm=seq, Wpack=wp, C_out=C, ldc=N, bias=b.to(d), R=Rr.to(d))
        outs.append(C)
    assert torch.equal(outs[0], outs[1])
    ref = A.double() @ W.double().t() + b.double() + Rr.double()
    assert rel(outs[0], ref) < 4e-5
    # transposed packing (the data-gradient form): W'[n][k] = W2[k][n]
    W2 = W.t().contiguous()                       # [Kd][N]
    dev.pack_w(W2.to(d), N, Kd, N, wp, trans=True, order=1)
    C = torch.empty(P, N, device=d)
    dev.gemm_b2p(A=Ab, K=Kd, sm=seq, Wpack=wp, C_out=C, ldc=N)
    assert rel(C, A.double() @ W.double().t()) < 4e-5


@pytest.mark.parametrize("dims", BLK_DIMS)
@pytest.mark.parametrize("view", ["time", "band"])
def test_gemm_tnb_vs_torch(view, dims):
    """[dW_ih | dW_hh] form: G column range, two A sources, the second shifted by one step."""
    from wesep_amd import dev
    d = _cuda()
    R, K, Tf, N, P, geo, smap, seq = _blk_setup(view, dims)
    g = torch.Generator().manual_seed(5)
    GW, g_off, g_cols = 512, 256, 256
    G, A0, A1 = rnd(g, P, GW), rnd(g, P, N), rnd(g, P, 512)
    Gb, A0b, A1b = (dev.to_blocked(t.to(d), seq, split=True) for t in (G, A0, A1))
    nb = dev.bl_num_blocks(seq)
    for shift in (-1, 1):
        ns, bps = dev.tnb_splits(nb, g_cols // 128)
        slab, bslab = torch.empty(ns, g_cols * 384, device=d), torch.empty(ns, g_cols, device=d)
        aslab = torch.empty(ns, 384, device=d)
        dev.gemm_tnb(G=Gb, g_width=GW, g_off=g_off, g_cols=g_cols, A0=A0b, a0_width=N, a0_off=0, a0_cols=N,
                     A1=A1b, a1_width=512, a1_off=256, a1_cols=256, a1_shift=shift, nblk=nb, L_=seq.L,
                     slab=slab, nsplit=ns, blocks_per_split=bps, bslab=bslab, aslab=aslab)
        out = slab.sum(0).view(g_cols, 384)
        # reference: shift the A1 rows by one step inside each sequence
        pos, valid = dev.bl_positions(seq, torch.device("cpu"))
        nt = -(-seq.nseq // 32)
        posv = pos.view(nt, seq.L, 32)
        val = valid.view(nt, seq.L, 32)
        A1s = torch.zeros(P, 256)
        src = torch.roll(posv, shifts=-shift, dims=1)          # position of step + shift
        ok = val.clone()
        if shift == -1:
            ok[:, 0] = False
        else:
            ok[:, -1] = False
        A1s[posv[ok]] = A1[src[ok]][:, 256:512]
        Gs = G[:, g_off:g_off + g_cols].double()
        ref = torch.cat([Gs.t() @ A0.double(), Gs.t() @ A1s.double()], 1)
        assert rel(out, ref) < 4e-5
        assert rel(bslab.sum(0), Gs.sum(0)) < 1e-5
        assert rel(aslab.sum(0), torch.cat([A0.double().sum(0), A1s.double().sum(0)])) < 1e-5
    # single A tile (dW_proj^T form): G = A1 (512 columns), A = A0
    ns, bps = dev.tnb_splits(nb, 4)
    slab, aslab = torch.empty(ns, 512 * 128, device=d), torch.empty(ns, 128, device=d)
    dev.gemm_tnb(G=A1b, g_width=512, g_off=0, g_cols=512, A0=A0b, a0_width=N, a0_off=0, a0_cols=N, nblk=nb,
                 L_=seq.L, slab=slab, nsplit=ns, blocks_per_split=bps, aslab=aslab)
    assert rel(slab.sum(0).view(512, 128), A1.double().t() @ A0.double()) < 4e-5
    assert rel(aslab.sum(0), A0.double().sum(0)) < 1e-5


@pytest.mark.parametrize("view,dims", [("time", (2, 32, 70)), ("time", (4, 32, 11)), ("band", (4, 9, 16))])
def test_lstm_cluster_fwd_bwd_vs_torch(view, dims):
    """Weight-stationary cluster recurrence (lstm_cluster.hip): same contract as the blocked forward;
    checked against torch's LSTM, for run-to-run identity, and against the streaming kernel."""
    from wesep_amd import dev, _lib as L
    from wesep_amd.functional import _view_maps
    d = _cuda()
    g = torch.Generator().manual_seed(41)
    (R, K, Tf), N, H = dims, 128, 256
    P = R * K * Tf
    geo, smap, seq, _ = _view_maps(view, R, K, Tf, N)
    assert seq.nseq % 64 == 0
    lstm = torch.nn.LSTM(N, H, 1, batch_first=True, bidirectional=True)
    x = rnd(g, R, K, Tf, N)
    xs = x.reshape(R * K, Tf, N) if view == "time" else x.permute(0, 2, 1, 3).reshape(R * Tf, K, N)
    xs = xs.clone().requires_grad_(True)
    out, _ = lstm(xs)
    dout_seq = rnd(g, *out.shape)
    out.backward(dout_seq)
    out = out.detach()
    with torch.no_grad():
        gx = []
        for sfx in ("", "_reverse"):
            w, bi, bh = (getattr(lstm, n + sfx) for n in ("weight_ih_l0", "bias_ih_l0", "bias_hh_l0"))
            gx.append(x.reshape(P, N) @ w.t() + bi + bh)
    gates0 = dev.to_blocked(torch.stack(gx, 1).reshape(P, 8 * H).to(d), seq)
    nb = dev.bl_num_blocks(seq)
    whf, whr = lstm.weight_hh_l0.detach().to(d).contiguous(), lstm.weight_hh_l0_reverse.detach().to(d).contiguous()
    res = []
    status = torch.zeros(1, device=d, dtype=torch.int32)
    for _ in range(2):
        gates = gates0.clone()
        cbuf, hcat = torch.zeros(nb, 2 * H // 4, 32, 4, device=d), torch.zeros(nb, 2 * H // 4, 32, 4, device=d)
        dev.lstm_fwd_cluster(gates, cbuf, hcat, whf, whr, seq, status=status)
        res.append((gates, cbuf, hcat))
    assert int(status.item()) == 0                                   # no bounded wait timed out
    assert all(torch.equal(a, b) for a, b in zip(res[0], res[1]))    # deterministic
    href = out.reshape(R, K, Tf, 2 * H) if view == "time" else out.reshape(R, Tf, K, 2 * H).permute(0, 2, 1, 3)
    assert rel(dev.from_blocked(res[0][2], seq, P, split=True).view(R, K, Tf, 2 * H), href) < 4e-5
    # the streaming kernel computes the same thing (different MFMA order: compare, do not equate)
    pf, pb = torch.empty(L.LSTM_PACK_FLOATS, device=d), torch.empty(L.LSTM_PACK_FLOATS, device=d)
    dev.lstm_pack(whf, whr, pf, pb, L.LSTM_BF16X3_BLK)
    gates = gates0.clone()
    cbuf, hcat = torch.zeros_like(res[0][1]), torch.zeros_like(res[0][2])
    dev.lstm_fwd(gates, cbuf, hcat, pf, seq, L.LSTM_BF16X3_BLK)
    for i, (a, b) in enumerate(zip(res[0], (gates, cbuf, hcat))):
        assert rel(dev.from_blocked(a, seq, P, split=i == 2), dev.from_blocked(b, seq, P, split=i == 2)) < 4e-5
    # ---- BPTT over the same clusters --------------------------------------------------------------
    dref = dout_seq.reshape(R, K, Tf, 2 * H) if view == "time" else dout_seq.reshape(R, Tf, K, 2 * H).permute(0, 2, 1, 3)
    dh = dev.to_blocked(dref.contiguous().reshape(P, 2 * H).to(d), seq)
    outs = []
    for _ in range(2):
        g2 = res[0][0].clone()
        dev.lstm_bwd_cluster(g2, res[0][1], dh, whf, whr, seq, status=status)
        outs.append(g2)
    assert int(status.item()) == 0
    assert torch.equal(outs[0], outs[1])
    dg = dev.from_blocked(outs[0], seq, P, split=True).view(P, 2, 4 * H).cpu()
    dx = dg[:, 0] @ lstm.weight_ih_l0.detach() + dg[:, 1] @ lstm.weight_ih_l0_reverse.detach()
    dxref = xs.grad.reshape(R, K, Tf, N) if view == "time" else xs.grad.reshape(R, Tf, K, N).permute(0, 2, 1, 3)
    assert rel(dx.view(R, K, Tf, N), dxref) < 8e-5
    assert rel(dg[:, 0].sum(0), lstm.bias_ih_l0.grad) < 8e-5
    assert rel(dg[:, 1].sum(0), lstm.bias_ih_l0_reverse.grad) < 8e-5


@pytest.mark.parametrize("view,dims", [("time", (2, 32, 70)), ("time", (3, 7, 37)), ("time", (32, 32, 9)),
                                       ("band", (4, 9, 16))])
def test_lstm_pair_bwd_vs_torch(view, dims):
    """Pair BPTT (lstm_pair.hip: two workgroups per (tile, direction), W_hh split by gate rows, hi plane resident, the
    partner's half of the partial dh exchanged every step): same contract as the blocked streaming BPTT -- checked
    against torch's LSTM autograd, against the streaming kernel on the same forward state, for run-to-run identity,
    with padded tiles ((3, 7, 37): 21 sequences) and at the headline launch geometry (1024 sequences = 128
    workgroups); then a forced timeout must poison d(gates) and raise the launch's word and the status word."""
    from wesep_amd import dev, _lib as L
    from wesep_amd.functional import _view_maps
    d = _cuda()
    g = torch.Generator().manual_seed(43)
    (R, K, Tf), N, H = dims, 128, 256
    P = R * K * Tf
    geo, smap, seq, _ = _view_maps(view, R, K, Tf, N)
    lstm = torch.nn.LSTM(N, H, 1, batch_first=True, bidirectional=True)
    x = rnd(g, R, K, Tf, N)
    xs = x.reshape(R * K, Tf, N) if view == "time" else x.permute(0, 2, 1, 3).reshape(R * Tf, K, N)
    xs = xs.clone().requires_grad_(True)
    out, _ = lstm(xs)
    dout_seq = rnd(g, *out.shape)
    out.backward(dout_seq)
    with torch.no_grad():
        gx = []
        for sfx in ("", "_reverse"):
            w, bi, bh = (getattr(lstm, n + sfx) for n in ("weight_ih_l0", "bias_ih_l0", "bias_hh_l0"))
            gx.append(x.reshape(P, N) @ w.t() + bi + bh)
    gates = dev.to_blocked(torch.stack(gx, 1).reshape(P, 8 * H).to(d), seq)
    nb = dev.bl_num_blocks(seq)
    whf, whr = lstm.weight_hh_l0.detach().to(d).contiguous(), lstm.weight_hh_l0_reverse.detach().to(d).contiguous()
    pf, pb = torch.empty(L.LSTM_PACK_FLOATS, device=d), torch.empty(L.LSTM_PACK_FLOATS, device=d)
    dev.lstm_pack(whf, whr, pf, pb, L.LSTM_BF16X3_BLK)
    cbuf, hcat = torch.zeros(nb, 2 * H // 4, 32, 4, device=d), torch.zeros(nb, 2 * H // 4, 32, 4, device=d)
    dev.lstm_fwd(gates, cbuf, hcat, pf, seq, L.LSTM_BF16X3_BLK)
    dref = dout_seq.reshape(R, K, Tf, 2 * H) if view == "time" else dout_seq.reshape(R, Tf, K, 2 * H).permute(0, 2, 1, 3)
    dh = dev.to_blocked(dref.contiguous().reshape(P, 2 * H).to(d), seq)
    g_stream = gates.clone()
    dev.lstm_bwd(g_stream, cbuf, hcat, dh, pb, seq, L.LSTM_BF16X3_BLK)
    pp = torch.empty(L.LSTM_PACK_FLOATS, device=d)
    dev.lstm_pack_pair(whf, whr, pp)
    status = torch.zeros(1, device=d, dtype=torch.int32)
    outs = []
    for _ in range(3):
        g2 = gates.clone()
        tw = dev.lstm_bwd_pair(g2, cbuf, dh, pp, seq, status=status)
        assert int(tw.item()) == 0
        outs.append(g2)
    assert int(status.item()) == 0
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])      # deterministic
    _, valid = dev.bl_positions(seq, d)
    if not bool(valid.all()):                                                   # padded slots stay zero
        rows = dev.bls_unpack(outs[0]).view(nb, -1, 32, 4).permute(0, 2, 1, 3).reshape(nb * 32, -1)
        assert float(rows[~valid].abs().max()) == 0.0
    dg_pair, dg_stream = (dev.from_blocked(t, seq, P, split=True) for t in (outs[0], g_stream))
    assert rel(dg_pair, dg_stream) < 2e-5
    dg = dg_pair.view(P, 2, 4 * H).cpu()
    dx = dg[:, 0] @ lstm.weight_ih_l0.detach() + dg[:, 1] @ lstm.weight_ih_l0_reverse.detach()
    dxref = xs.grad.reshape(R, K, Tf, N) if view == "time" else xs.grad.reshape(R, Tf, K, N).permute(0, 2, 1, 3)
    assert rel(dx.view(R, K, Tf, N), dxref) < 8e-5
    assert rel(dg[:, 0].sum(0), lstm.bias_ih_l0.grad) < 8e-5
    assert rel(dg[:, 1].sum(0), lstm.bias_ih_l0_reverse.grad) < 8e-5
    if seq.L > 3:
        # a wait that times out (test build: pair 0, member 0, wave 0 at step 2): NaN from there on + both words
        g3 = gates.clone()
        tw = dev.lstm_bwd_pair(g3, cbuf, dh, pp, seq, status=status, dbg=8)
        assert int(tw.item()) == 1 and int(status.item()) == 1
        assert torch.isnan(g3).any()


@pytest.mark.parametrize("N", [16, 24, 32, 48, 64, 96])
def test_generic_bf16_gemms_narrow_column_tiles(N):
    """Ungrouped split-bf16 launches with few output columns run narrow tiles (32 / 64 columns per workgroup in
    ws_gemm_nt, 32 gradient columns in ws_gemm_tn): same results as fp64, run-to-run identical, partial row tiles."""
    from wesep_amd import dev
    d = _cuda()
    g = torch.Generator().manual_seed(90 + N)
    M, K = 1000 + N, 100
    A, W, b, Rr = rnd(g, M, K), rnd(g, N, K, scale=0.1), rnd(g, N), rnd(g, M, N)
    outs = []
    for _ in range(2):
        C = torch.full((M, N), float("nan"), device=d)
        dev.gemm_nt(A=A.to(d), a_rows=dev.flat(K), M=M, N=N, K=K, W=W.to(d), ldw=K, bias=b.to(d), C_out=C,
                    c_rows=dev.flat(N), R=Rr.to(d), mode="bf16x3")
        outs.append(C)
    assert torch.equal(outs[0], outs[1])
    assert rel(outs[0], A.double() @ W.double().t() + b.double() + Rr.double()) < 4e-5
    G = rnd(g, M, N)
    ns, rps = 3, 384
    outs = []
    for _ in range(2):
        slab, bslab = torch.full((ns, N * K), float("nan"), device=d), torch.full((ns, N), float("nan"), device=d)
        dev.gemm_tn(G=G.to(d), g_rows=dev.flat(N), A=A.to(d), a_rows=dev.flat(K), M=M, Nn=N, Kk=K, slab=slab,
                    slab_stride=N * K, nsplit=ns, rows_per_split=rps, bslab=bslab, bslab_stride=N, mode="bf16x3")
        outs.append((slab, bslab))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert rel(outs[0][0].sum(0).view(N, K), G.double().t() @ A.double()) < 4e-5
    assert rel(outs[0][1].sum(0), G.double().sum(0)) < 1e-5


@pytest.mark.parametrize("kind", ["nt", "tn"])
def test_generic_bf16_gemms_have_no_outliers_at_scale(kind):
    """The split-bf16 generic GEMMs (mask MLP / speaker path) at a many-workgroup size: element-wise
    error bound (a relative-L2 check would hide a handful of corrupted elements) and run-to-run
    identity.  (The norm-on-load variant failed exactly this and is routed to the fp32 kernel.)"""
    from wesep_amd import dev
    d = _cuda()
    torch.manual_seed(7)
    M, N, K = 131072, 512, 512
    A = torch.randn(M, K, device=d)
    if kind == "nt":
        W = torch.randn(N, K, device=d) * 0.05
        outs = []
        for _ in range(2):
            C = torch.empty(M, N, device=d)
            dev.gemm_nt(A=A, a_rows=dev.flat(K), M=M, N=N, K=K, W=W, ldw=K, C_out=C, c_rows=dev.flat(N), mode="bf16x3")
            outs.append(C)
        ref = torch.empty(M, N, device=d)
        dev.gemm_nt(A=A, a_rows=dev.flat(K), M=M, N=N, K=K, W=W, ldw=K, C_out=ref, c_rows=dev.flat(N), mode="f32")
    else:
        G = torch.randn(M, N, device=d) * 0.05
        ns, rps = dev.tn_splits(M)
        outs = []
        for mode in ("bf16x3", "bf16x3", "f32"):
            slab = torch.empty(ns, N * K, device=d)
            dev.gemm_tn(G=G, g_rows=dev.flat(N), A=A, a_rows=dev.flat(K), M=M, Nn=N, Kk=K, slab=slab,
                        slab_stride=N * K, nsplit=ns, rows_per_split=rps, mode=mode)
            outs.append(slab.sum(0))
        ref = outs.pop()
    assert torch.equal(outs[0], outs[1])
    assert float((outs[0] - ref).abs().max()) < 2e-4 * float(ref.abs().max())


@pytest.mark.parametrize("kind", ["nt", "tn"])
def test_generic_bf16_norm_on_load_gemms_at_scale(kind):
    """GroupNorm-on-load variants of the generic split-bf16 GEMMs at a many-workgroup size: run-to-run
    identity and an element-wise bound against the exact-fp32 kernel (the first, branchy version of
    the NT kernel failed this with sparse O(1) errors)."""
    from wesep_amd import dev
    d = _cuda()
    torch.manual_seed(11)
    M, N, K, L_ = 120240, 512, 128, 501          # 240 norm groups of 501 rows
    A = torch.randn(M, K, device=d)
    stats = torch.stack([torch.randn(M // L_, device=d) * 0.1, torch.rand(M // L_, device=d) + 0.5], 1).contiguous()
    kw = dict(stats=stats, gamma=torch.randn(K, device=d), beta=torch.randn(K, device=d),
              stat_map=dev.StatMap(L_, 1, 1, 0, 0))
    outs = []
    if kind == "nt":
        W = torch.randn(N, K, device=d) * 0.05
        for mode in ("bf16x3", "bf16x3", "f32"):
            C = torch.empty(M, N, device=d)
            dev.gemm_nt(A=A, a_rows=dev.flat(K), M=M, N=N, K=K, W=W, ldw=K, C_out=C, c_rows=dev.flat(N), mode=mode, **kw)
            outs.append(C)
    else:
        G = torch.randn(M, N, device=d) * 0.05
        ns, rps = dev.tn_splits(M)
        for mode in ("bf16x3", "bf16x3", "f32"):
            slab = torch.empty(ns, N * K, device=d)
            dev.gemm_tn(G=G, g_rows=dev.flat(N), A=A, a_rows=dev.flat(K), M=M, Nn=N, Kk=K, slab=slab,
                        slab_stride=N * K, nsplit=ns, rows_per_split=rps, mode=mode, **kw)
            outs.append(slab.sum(0))
    ref = outs.pop()
    assert torch.equal(outs[0], outs[1])
    assert float((outs[0] - ref).abs().max()) < 3e-4 * float(ref.abs().max())


# ----------------------------------------------------------------------------------------------
# one-pass row LayerNorm for short rows (norm.hip ws_rowln_*; TF-GridNet's per-position LayerNorm)
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,W", [(1000, 128), (777, 48), (513, 256), (9, 4), (4099, 200), (70000, 128)])
def test_rowln_fwd_bwd_vs_torch(M, W):
    from wesep_amd import dev
    d = _cuda()
    g = torch.Generator().manual_seed(M + W)
    x = (rnd(g, M, W) * 2.0 + 0.5).to(d)
    gamma, beta, dy, res = rnd(g, W).to(d), rnd(g, W).to(d), rnd(g, M, W).to(d), rnd(g, M, W).to(d)
    y, st = torch.full((M, W), float("nan"), device=d), torch.full((M, 2), float("nan"), device=d)
    dev.rowln_fwd(x, gamma, beta, M, W, y, st)
    xr = x.double().requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xr, (W,), gr, br, 1e-5)
    assert rel(y, ref.detach()) < 2e-6
    assert rel(st[:, 0], x.double().mean(1)) < 1e-6
    ref.backward(dy.double())
    outs = []
    for _ in range(2):
        dx = torch.full((M, W), float("nan"), device=d)
        tot = dev.rowln_bwd(x, dy, st, gamma, M, W, dx, res=res)
        outs.append((dx, tot))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])     # reproducible
    dx, tot = outs[0]
    assert rel(dx - res, xr.grad) < 5e-6
    assert rel(tot[0], br.grad) < 5e-6 and rel(tot[1], gr.grad) < 5e-6
    dx2 = dy.clone()                                                                        # in place, no residual
    dev.rowln_bwd(x, dx2, st, gamma, M, W, dx2)
    assert rel(dx2, xr.grad) < 5e-6
    with pytest.raises(Exception, match="ws_rowln_fwd"):
        dev.rowln_fwd(x, gamma, beta, M, 260, y, st)                                        # wider rows are refused
