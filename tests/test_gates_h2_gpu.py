"""GPU: the 2-byte storage formats of the saved recurrence state (include/wesep_hip.h WS_GATES_H2 / WS_GATES_H2S, ABI v15)
against the fp32 format of the same kernels -- which tests/test_kernels_gpu.py holds to torch's LSTM.

The formats change what is STORED, not what is computed, so the statements are exact wherever the arithmetic allows:
  * forward (streaming 32 / 16 sequences, fused projection 32 / 64 sequences, cluster): cell state and h bit-identical to
    the fp32-format launch; the unorm16 gate codes decode to the fp32 gates within half a code step;
  * BPTT (streaming 32 / 16, pair) on gates that sit on the unorm16 grid: WS_GATES_H2S d(gates) bit-identical to the fp32
    format's split pairs, WS_GATES_H2 d(gates) bit-identical to their hi terms (bf16, round to nearest even), WS_GATES_H2F
    d(gates) = fp16 of the scaled value (half an fp16 ulp + the split pair's 2^-17 from the fp32 format's);
  * ws_gemm_b2p (a_fmt 1 / 2) / ws_gemm_tnb (g_fmt 1 / 2) on bf16 / scaled-fp16 operands: bit-identical to the split-pair
    kernels fed the same values (zero lo term / the exact bf16 hi + lo split of an fp16 value; the power-of-two scale is
    exact to undo), and within the split-product tolerance of fp64."""
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


def rnd(g, *shape, scale=1.0):
    return torch.randn(*shape, generator=g) * scale


def _setup(view, dims, seed, d):
    from wesep_amd import dev
    from wesep_amd.functional import _view_maps
    g = torch.Generator().manual_seed(seed)
    R, K, Tf = dims
    P = R * K * Tf
    geo, smap, seq, _ = _view_maps(view, R, K, Tf, N)
    nb = dev.bl_num_blocks(seq)
    whf, whr = rnd(g, 4 * H, H, scale=0.06).to(d), rnd(g, 4 * H, H, scale=0.06).to(d)
    pre = dev.to_blocked(rnd(g, P, 8 * H).to(d), seq)            # x-projection + biases, BL(2 * 4H) fp32
    dh = dev.to_blocked(rnd(g, P, 2 * H).to(d), seq)
    return seq, nb, P, whf, whr, pre, dh, g


def _state(nb, d):
    return (torch.full((nb, 2 * H // 4, 32, 4), float("nan"), device=d), torch.full((nb, 2 * H // 4, 32, 4), float("nan"), device=d))


def _gates_h(nb, d):
    from wesep_amd import dev
    return torch.zeros(dev.blh_floats(nb, 8 * H), device=d)


def _check_codes(gh, g32, nb):
    """unorm16 codes decode to the fp32 gates within half a code step (+ one ulp of the fma)."""
    from wesep_amd import dev
    dec = dev.blh_gates_unpack(gh, nb)
    Cq = 8 * H // 4
    is_g = ((torch.arange(Cq, device=gh.device) >> 6) & 3 == 2).view(1, Cq, 1, 1)
    step = torch.where(is_g, torch.tensor(1.0 / 32767.5, device=gh.device), torch.tensor(1.0 / 65535.0, device=gh.device))
    err = ((dec - g32.view(nb, Cq, 32, 4)).abs() / step).max()
    assert float(err) <= 0.5 + 1e-2, float(err)
    return dec


FWD_KINDS = ["blk32", "blk16", "fused32", "fused64", "cluster"]


@pytest.mark.parametrize("kind", FWD_KINDS)
@pytest.mark.parametrize("view,dims", [("time", (2, 32, 70)), ("band", (3, 7, 37))])
def test_forward_kernels_store_unorm16_gates(kind, view, dims, monkeypatch):
    from wesep_amd import _lib as L
    from wesep_amd import dev
    d = _cuda()
    seq, nb, P, whf, whr, pre, dh, g = _setup(view, dims, 7, d)
    if kind == "cluster" and not dev.lstm_cluster_ok(seq, d):
        pytest.skip("cluster geometry")
    c0, h0 = _state(nb, d)
    c1, h1 = _state(nb, d)
    g32, gh = pre.clone(), _gates_h(nb, d)
    if kind in ("blk32", "blk16"):
        mode = L.LSTM_BF16X3_BLK if kind == "blk32" else L.LSTM_BF16X3_BLK16
        pf, pb = torch.empty(L.LSTM_PACK_FLOATS, device=d), torch.empty(L.LSTM_PACK_FLOATS, device=d)
        dev.lstm_pack(whf, whr, pf, pb, mode)
        dev.lstm_fwd(g32, c0, h0, pf, seq, mode)
        dev.lstm_fwd(gh, c1, h1, pf, seq, mode, gfmt=L.GATES_H2, gates_in=pre)
    elif kind == "cluster":
        st = torch.zeros(1, device=d, dtype=torch.int32)
        dev.lstm_fwd_cluster(g32, c0, h0, whf, whr, seq, status=st)
        dev.lstm_fwd_cluster(gh, c1, h1, whf, whr, seq, status=st, gfmt=L.GATES_H2, gates_in=pre)
        assert int(st.item()) == 0
    else:
        monkeypatch.setenv("WS_FUSED_SEQS", "64" if kind == "fused64" else "32")
        wih_f, wih_r = rnd(g, 4 * H, N, scale=0.06).to(d), rnd(g, 4 * H, N, scale=0.06).to(d)
        bias = rnd(g, 2 * 4 * H).to(d)
        xn = dev.to_blocked(rnd(g, P, N).to(d), seq, split=True)
        fp = torch.empty(L.LSTM_FUSED_PACK_FLOATS, device=d)
        dev.lstm_pack_fused(wih_f, wih_r, whf, whr, fp)
        g32 = torch.full_like(pre, float("nan"))
        dev.lstm_fwd_fused(g32, c0, h0, xn, fp, bias, seq)
        dev.lstm_fwd_fused(gh, c1, h1, xn, fp, bias, seq, gfmt=L.GATES_H2)
    torch.cuda.synchronize()
    assert not torch.isnan(c0).any() and not torch.isnan(h0).any()
    assert torch.equal(c0, c1) and torch.equal(h0, h1)          # the arithmetic is the fp32 format's
    _check_codes(gh, g32, nb)


@pytest.mark.parametrize("kind", ["blk32", "blk16", "pair"])
@pytest.mark.parametrize("view,dims", [("time", (2, 32, 70)), ("time", (3, 7, 37)), ("band", (4, 9, 16))])
def test_bptt_kernels_read_unorm16_gates_and_store_bf16(kind, view, dims):
    from wesep_amd import _lib as L
    from wesep_amd import dev
    d = _cuda()
    seq, nb, P, whf, whr, pre, dh, g = _setup(view, dims, 11, d)
    if kind == "pair" and not dev.lstm_pair_ok(seq, d):
        pytest.skip("pair geometry")
    mode = L.LSTM_BF16X3_BLK16 if kind == "blk16" else L.LSTM_BF16X3_BLK
    pf, pb = torch.empty(L.LSTM_PACK_FLOATS, device=d), torch.empty(L.LSTM_PACK_FLOATS, device=d)
    dev.lstm_pack(whf, whr, pf, pb, mode)
    cbuf, hcat = _state(nb, d)
    gh = _gates_h(nb, d)
    dev.lstm_fwd(gh, cbuf, hcat, pf, seq, mode, gfmt=L.GATES_H2, gates_in=pre)
    gq = dev.blh_gates_unpack(gh, nb).view_as(pre).contiguous()   # the saved gates, on the unorm16 grid, as fp32
    pp = torch.empty(L.LSTM_PACK_FLOATS, device=d)
    dev.lstm_pack_pair(whf, whr, pp)
    st = torch.zeros(1, device=d, dtype=torch.int32)

    def bptt(gates, gfmt, dgates=None):
        if kind == "pair":
            tw = dev.lstm_bwd_pair(gates, cbuf, dh, pp, seq, status=st, gfmt=gfmt, dgates=dgates)
            assert int(tw.item()) == 0
        else:
            dev.lstm_bwd(gates, cbuf, hcat, dh, pb, seq, mode, gfmt=gfmt, dgates=dgates)

    def bits(t):                                                  # (codes read as fp32 hold NaN patterns: compare bits)
        return t.contiguous().view(torch.int32)

    ref = gq.clone()
    bptt(ref, L.GATES_F32)                                        # split pairs, in place over fp32 gates
    h2s_g, h2s_d = gh.clone(), torch.full_like(pre, float("nan"))
    bptt(h2s_g, L.GATES_H2S, h2s_d)
    h2 = gh.clone()
    bptt(h2, L.GATES_H2)
    h2b = gh.clone()
    bptt(h2b, L.GATES_H2)
    h2o_g, h2o_d = gh.clone(), torch.zeros_like(gh)               # H2, out of place
    bptt(h2o_g, L.GATES_H2, h2o_d)
    torch.cuda.synchronize()
    assert int(st.item()) == 0
    assert torch.equal(bits(h2s_g), bits(gh))                     # H2S leaves the saved gates alone
    assert torch.equal(bits(h2s_d), bits(ref))
    hi = (bits(ref) >> 16).to(torch.int16)                        # bf16 hi terms of the split pairs
    got = h2.reshape(-1)[: ref.numel() // 2].view(torch.int16).view(hi.shape)
    assert torch.equal(got, hi)
    assert torch.equal(bits(h2), bits(h2b))                       # deterministic
    assert torch.equal(bits(h2o_g), bits(gh)) and torch.equal(bits(h2o_d), bits(h2))
    dg = dev.bls_unpack(ref)
    assert float(dg.abs().max()) > 0.0
    # ---- H2F: fp16 of d(gates) scaled by the power of two that max |d(hcat)| defines ------------------------------------
    amax = dh.abs().max().reshape(1).view(torch.int32).clone()    # what ws_gemm_p2b's atomic max leaves behind
    e = (int(amax.item()) >> 23) & 0xFF
    S = L.dgates_scale(int(amax.item()))
    f_in, f_out_g, f_out_d = gh.clone(), gh.clone(), torch.zeros_like(gh)

    def bptt_f(gates, dgates=None):
        if kind == "pair":
            dev.lstm_bwd_pair(gates, cbuf, dh, pp, seq, status=st, gfmt=L.GATES_H2F, dgates=dgates, amax=amax)
        else:
            dev.lstm_bwd(gates, cbuf, hcat, dh, pb, seq, mode, gfmt=L.GATES_H2F, dgates=dgates, amax=amax)

    bptt_f(f_in)
    bptt_f(f_out_g, f_out_d)
    torch.cuda.synchronize()
    assert torch.equal(bits(f_out_g), bits(gh)) and torch.equal(bits(f_out_d), bits(f_in))
    got = f_in.reshape(-1)[: ref.numel() // 2].view(torch.float16).view(dg.shape).float() / S
    assert float((dg.abs() * S).max()) < 65504.0                  # nothing near the clamp
    err = (got - dg).abs()
    bound = dg.abs() * (2.0 ** -11 + 2.0 ** -16) + (2.0 ** -24) / S      # half an fp16 ulp (+ subnormal step) + the pair's 2^-17
    assert bool((err <= bound).all()), float((err / bound).max())
    assert float((got - dg).norm() / dg.norm()) < 3e-4
    if kind == "blk32":
        # ---- ws_lstm_args.rfmt = 2 (ABI v18): the streaming BPTT's recurrent product on the stored fp16 d(gates) x W_hh as fp16
        # hi + scaled-FP8 lo (ws_lstm_pack_bwd_f8): the pair kernel's rfmt 2 arithmetic in the stream-fed kernel; same statements
        pb8 = torch.zeros(L.LSTM_PACK_FLOATS, device=d)
        dev.lstm_pack_bwd_f8(whf, whr, pb8)
        s_in, s_in2, s_out_g, s_out_d = gh.clone(), gh.clone(), gh.clone(), torch.zeros_like(gh)
        for gates_, dg_ in ((s_in, None), (s_in2, None), (s_out_g, s_out_d)):
            dev.lstm_bwd(gates_, cbuf, hcat, dh, pb8, seq, mode, gfmt=L.GATES_H2F, dgates=dg_, amax=amax, rfmt=2)
        torch.cuda.synchronize()
        assert torch.equal(bits(s_in), bits(s_in2))
        assert torch.equal(bits(s_out_g), bits(gh)) and torch.equal(bits(s_out_d), bits(s_in))
        got2 = s_in.reshape(-1)[: ref.numel() // 2].view(torch.float16).view(dg.shape).float() / S
        assert bool(torch.isfinite(got2).all())
        e2, emax = float((got2 - dg).norm() / dg.norm()), float((got2 - dg).abs().max() / dg.abs().max())
        print(f"streaming BPTT rfmt 2: d(gates) vs the three-term kernel rel-L2 {e2:.2e}, max {emax:.2e}")
        assert e2 < 6e-4 and emax < 2e-3, (e2, emax)
        # rfmt = 3 (ABI v20): the same pack, the lo term on the block-scaled FP8 matrix instruction (codes x e4m3 of d(gates) / 256):
        # the same statements, and next to rfmt 2 (the term is 2^-12 of the product and keeps 2^-4 of itself)
        t_in, t_in2, t_out_g, t_out_d = gh.clone(), gh.clone(), gh.clone(), torch.zeros_like(gh)
        for gates_, dg_ in ((t_in, None), (t_in2, None), (t_out_g, t_out_d)):
            dev.lstm_bwd(gates_, cbuf, hcat, dh, pb8, seq, mode, gfmt=L.GATES_H2F, dgates=dg_, amax=amax, rfmt=3)
        torch.cuda.synchronize()
        assert torch.equal(bits(t_in), bits(t_in2))
        assert torch.equal(bits(t_out_g), bits(gh)) and torch.equal(bits(t_out_d), bits(t_in))
        got3 = t_in.reshape(-1)[: ref.numel() // 2].view(torch.float16).view(dg.shape).float() / S
        assert bool(torch.isfinite(got3).all())
        e3, e3max = float((got3 - dg).norm() / dg.norm()), float((got3 - dg).abs().max() / dg.abs().max())
        e32 = float((got3 - got2).norm() / got2.norm())
        print(f"streaming BPTT rfmt 3: d(gates) vs the three-term kernel rel-L2 {e3:.2e}, max {e3max:.2e}; vs rfmt 2 {e32:.2e}")
        assert e3 < 6e-4 and e3max < 2e-3 and e32 < 4e-4, (e3, e3max, e32)
        # the pack: hi plane = fp16(256 w), codes finite and <= 256, scales bracket each group's maximum
        raw = pb8.view(torch.uint8).reshape(16, 131072).cpu()
        codes = torch.stack([raw[:, c * 6144 + 4096:(c + 1) * 6144] for c in range(16)], 1).contiguous().view(torch.float8_e4m3fn).float()
        assert bool(torch.isfinite(codes).all()) and float(codes.abs().max()) <= 256.0
        Sg = raw[:, 98304:98304 + 32].contiguous().view(torch.float32)                     # [16 (d, w)][8 groups]
        hi8 = torch.stack([raw[:, c * 6144:c * 6144 + 4096] for c in range(16)], 1).contiguous().view(torch.float16).float()
        m = hi8.reshape(16, 8, -1).abs().amax(2)                                           # two chunks per group
        assert bool(((m >= Sg * 2.0 ** 19 * (1 - 2.0 ** -10)) & (m <= Sg * 2.0 ** 20)).all())
        # ---- ws_lstm_args.dxn (ABI v19): the same launch also writes d(xn) = d(gates) W_ih per direction -- from the d(gates)
        # image in LDS, W_ih^T as fp16 hi + scaled-FP8 lo on v_mfma_f32_16x16x32_f16 -- as plain rows at the sequence map's
        # positions.  d(gates) must not change (bit-identical to the launch without it); d(xn) against fp64 on the STORED d(gates)
        # and against ws_gemm_b2p(a_fmt 2), whose job it takes over
        wcat = (0.06 * torch.randn(2, 4 * H, N, generator=g)).to(d)
        px = torch.zeros(L.LSTM_DX_PACK_FLOATS, device=d)
        dev.lstm_pack_dx_f8(wcat, px)
        x_in, x_in2 = gh.clone(), gh.clone()
        dxn, dxn_b = torch.full((2, P, N), float("nan"), device=d), torch.full((2, P, N), float("nan"), device=d)
        dev.lstm_bwd(x_in, cbuf, hcat, dh, pb8, seq, mode, gfmt=L.GATES_H2F, amax=amax, rfmt=2, dxn=dxn, wxpack=px)
        dev.lstm_bwd(x_in2, cbuf, hcat, dh, pb8, seq, mode, gfmt=L.GATES_H2F, amax=amax, rfmt=2, dxn=dxn_b, wxpack=px)
        torch.cuda.synchronize()
        assert bool(torch.isfinite(dxn).all())                     # every row of both directions was written
        assert torch.equal(dxn, dxn_b)
        # (one accumulator chain instead of two, half of the waves walk the k-steps pairwise swapped: the same products in another
        #  summation order -- last-bit differences of d(h) that flip fp16 roundings of the stored d(gates): ulps, not errors)
        gx = x_in.reshape(-1)[: ref.numel() // 2].view(torch.float16).view(dg.shape).float() / S
        e_x = float((gx - got2).norm() / got2.norm())
        print(f"d(gates) of the launch that also writes d(xn) vs the plain rfmt 2 launch: rel-L2 {e_x:.2e}")
        assert e_x < 2e-4 and float((gx - dg).norm() / dg.norm()) < 6e-4
        wt16 = torch.empty(N * 2 * 4 * H, device=d)
        dev.pack_w(wcat.reshape(2 * 4 * H, N), N, 2 * 4 * H, N, wt16, trans=True, order=1, f16=True)
        c_ref = torch.full((P, N), float("nan"), device=d)
        dev.gemm_b2p(A=x_in, K=2 * 4 * H, sm=seq, Wpack=wt16, C_out=c_ref, ldc=N, a_fmt=2, amax=amax)
        torch.cuda.synchronize()
        rows = dev.from_blocked(dev.blh_f16_unpack(x_in, nb, 2 * 4 * H) / S, seq, P).double()     # the stored d(gates), plain rows
        for di in (0, 1):
            want = rows[:, di * 4 * H:(di + 1) * 4 * H] @ wcat[di].double()
            e_d = float((dxn[di].double() - want).norm() / want.norm())
            print(f"d(xn) of direction {di} vs fp64 on the stored d(gates): rel-L2 {e_d:.2e}")
            assert e_d < 2e-5, (di, e_d)                           # 16-bit weights (2^-17 of the group maximum), fp32 accumulation
        e_sum = float(((dxn[0] + dxn[1]) - c_ref).norm() / c_ref.norm())
        print(f"d(xn) inside the BPTT vs ws_gemm_b2p(a_fmt 2) on the same d(gates): rel-L2 {e_sum:.2e}")
        assert e_sum < 3e-5, e_sum                                 # weights 2^-17 apart (FP8 vs fp16 remainder), fp32 accumulation
    if kind != "pair":
        return
    # ---- rfmt = 1 (ABI v17): the recurrent product takes the STORED fp16 d(gates) (one operand of the fp16 MFMA, W_hh as
    # fp16 hi / lo of 256 w, two terms).  Every step's rounding (2^-12 relative, random) now travels down the recurrence, so
    # the statement is a tolerance against the three-term kernel, plus determinism and in-place == out-of-place
    # rfmt = 2 (ABI v18): the same with the lo plane of W_hh as block-scaled FP8 (16 instead of 22 bits of every weight;
    # all of W_hh stays on the CU): the same bounds, and next to rfmt = 1 (same B operand, weights 2^-16 apart)
    # rfmt = 3 (ABI v20): rfmt 2's codes as operands of the block-scaled FP8 matrix instruction against e4m3 of d(gates) / 256 --
    # the lo term is 2^-12 of the product and keeps 2^-4 of itself: next to rfmt 2
    by_rfmt = {}
    for rfmt in (1, 2, 3):
        pp16 = torch.empty(L.LSTM_PACK_FLOATS, device=d)
        dev.lstm_pack_pair(whf, whr, pp16, f16=rfmt)
        r_in, r_in2, r_out_g, r_out_d = gh.clone(), gh.clone(), gh.clone(), torch.zeros_like(gh)
        for gates_, dg_ in ((r_in, None), (r_in2, None), (r_out_g, r_out_d)):
            tw = dev.lstm_bwd_pair(gates_, cbuf, dh, pp16, seq, status=st, gfmt=L.GATES_H2F, dgates=dg_, amax=amax, rfmt=rfmt)
            assert int(tw.item()) == 0
        torch.cuda.synchronize()
        assert torch.equal(bits(r_in), bits(r_in2))
        assert torch.equal(bits(r_out_g), bits(gh)) and torch.equal(bits(r_out_d), bits(r_in))
        got1 = r_in.reshape(-1)[: ref.numel() // 2].view(torch.float16).view(dg.shape).float() / S
        assert bool(torch.isfinite(got1).all())
        e2, emax = float((got1 - dg).norm() / dg.norm()), float((got1 - dg).abs().max() / dg.abs().max())
        print(f"pair rfmt {rfmt}: d(gates) vs the three-term kernel rel-L2 {e2:.2e}, max {emax:.2e}")
        assert e2 < 6e-4 and emax < 2e-3, (rfmt, e2, emax)
        by_rfmt[rfmt] = got1
    e12 = float((by_rfmt[2] - by_rfmt[1]).norm() / by_rfmt[1].norm())
    e32 = float((by_rfmt[3] - by_rfmt[2]).norm() / by_rfmt[2].norm())
    print(f"pair rfmt 2 vs 1: rel-L2 {e12:.2e}; 3 vs 2: {e32:.2e}")
    assert e12 < 4e-4 and e32 < 4e-4, (e12, e32)


def test_pair_pack_fp8_lo_plane_reconstructs_the_weights():
    """ws_lstm_pack_pair_f8: per (direction, half, wave) block of 64 KB -- fp16 hi of 256 w (32 KB, the fp16 pack's order),
    e4m3 codes of (256 w - hi) / S (16 KB), S = 2^(e - 20) for the block's max |256 w| in [2^(e-1), 2^e) (one float at
    48 KB).  Decoded on the host: hi + S * lo = 256 w to 2^-15 of each weight (plus the block's code floor), no code is
    NaN, the largest code magnitude is <= 256."""
    from wesep_amd import dev
    d = _cuda()
    g = torch.Generator().manual_seed(5)
    for scale in (0.06, 0.9, 3e-3):
        whf, whr = (scale * torch.randn(4 * H, H, generator=g)).to(d), (scale * torch.randn(4 * H, H, generator=g)).to(d)
        whf[7, 9] = 0.0
        p16, p8 = torch.zeros(L.LSTM_PACK_FLOATS, device=d), torch.zeros(L.LSTM_PACK_FLOATS, device=d)
        dev.lstm_pack_pair(whf, whr, p16, f16=1)
        dev.lstm_pack_pair(whf, whr, p8, f16=2)
        torch.cuda.synchronize()
        b16 = p16.view(torch.uint8).reshape(32, 65536).cpu()
        b8 = p8.view(torch.uint8).reshape(32, 65536).cpu()
        assert torch.equal(b8[:, :32768], b16[:, :32768])                      # the hi plane IS the fp16 pack's
        hi = b8[:, :32768].contiguous().view(torch.float16).float().reshape(32, 32, 64, 8)
        lo16 = b16[:, 32768:].contiguous().view(torch.float16).float().reshape(32, 32, 64, 8)
        codes = b8[:, 32768:49152].contiguous().view(torch.float8_e4m3fn).float().reshape(32, 32, 64, 8)
        Sb = b8[:, 49152:49156].contiguous().view(torch.float32).reshape(32)
        assert bool(torch.isfinite(codes).all()) and float(codes.abs().max()) <= 256.0
        w256 = hi + lo16                                                       # 22 bits of 256 w
        m = w256.abs().reshape(32, -1).amax(1)
        assert bool(((m >= Sb * 2.0 ** 19) & (m < Sb * 2.0 ** 20)).all()), (m, Sb)
        err = (hi + codes * Sb.view(32, 1, 1, 1) - w256).abs()
        # (+ 2^-24: the REFERENCE's lo is an fp16, a subnormal one for small weights -- the codes come from the fp32 remainder)
        bound = w256.abs() * 2.0 ** -14 + Sb.view(32, 1, 1, 1) * 2.0 ** -10 + lo16.abs() * 2.0 ** -10 + 2.0 ** -24
        assert bool((err <= bound).all()), float((err / bound).max())
        print(f"fp8 lo plane, |w| ~ {scale}: worst error {float((err / w256.abs().clamp_min(1e-30)).max()):.2e} of the weight; "
              f"rel-L2 {float(err.norm() / w256.norm()):.2e}; scales 2^{int(torch.log2(Sb.min()))}..2^{int(torch.log2(Sb.max()))}")


@pytest.mark.parametrize("view,dims", [("time", (2, 5, 11)), ("band", (3, 32, 37))])
def test_gemm_b2p_scaled_fp16_operand(view, dims):
    """a_fmt = 2: A = fp16(x * S) feeds v_mfma_f32_32x32x16_f16 as it is; the weights come as fp16 hi + lo of 256 w
    (ws_pack_w_f16: 22 bits), the epilogue undoes 256 S exactly.  Against fp64 on the stored values, and against the
    split-pair kernel fed the same values (different weight split: compare, do not equate)."""
    from wesep_amd import dev
    from wesep_amd.functional import _view_maps
    d = _cuda()
    g = torch.Generator().manual_seed(14)
    R, K, Tf = dims
    P, Kd = R * K * Tf, 2048
    geo, smap, seq, _ = _view_maps(view, R, K, Tf, N)
    amax = torch.tensor([3.1e-6], dtype=torch.float32).view(torch.int32).to(d)      # max |d(hcat)| of a training step
    S = L.dgates_scale(int(amax.item()))
    Ah = (rnd(g, P, Kd) * 1e-6 * S).to(torch.float16)             # what the BPTT would have stored
    A = Ah.float() / S                                             # the values it stands for (exact)
    W = rnd(g, N, Kd, scale=0.05)
    wp, wp16 = torch.empty(N * Kd, device=d), torch.empty(N * Kd, device=d)
    dev.pack_w(W.t().contiguous().to(d), N, Kd, N, wp, trans=True, order=1)
    dev.pack_w(W.t().contiguous().to(d), N, Kd, N, wp16, trans=True, order=1, f16=True)
    Abl = dev.to_blocked(A.to(d), seq)
    Ahbl = dev.to_blocked(Ah.float().to(d), seq).to(torch.float16).contiguous().view(torch.float32)
    outs = []
    for _ in range(2):
        C_new = torch.full((P, N), float("nan"), device=d)
        dev.gemm_b2p(A=Ahbl, K=Kd, sm=seq, Wpack=wp16, C_out=C_new, ldc=N, a_fmt=2, amax=amax)
        outs.append(C_new)
    C_ref = torch.full((P, N), float("nan"), device=d)
    dev.gemm_b2p(A=dev.bls_pack(Abl), K=Kd, sm=seq, Wpack=wp, C_out=C_ref, ldc=N)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1])
    ref = A.double() @ W.double().t()
    assert rel(outs[0], ref) < 1e-5                                # 11-bit operand (exact here) x 22-bit weights, fp32 accumulation
    assert rel(C_ref, ref) < 4e-5 and rel(outs[0], C_ref) < 4e-5
    # a_fmt = 3 (ABI v20): the lo term on the block-scaled FP8 matrix instruction -- fp16 hi + e4m3 lo fragments of the weights
    # (ws_pack_w_f16f8), e4m3 of A / 256 built in registers: the term is 2^-12 of the product and keeps 2^-4 of itself
    wp8 = torch.empty(N * Kd, device=d)
    dev.pack_w(W.t().contiguous().to(d), N, Kd, N, wp8, trans=True, order=1, f16=2)
    o3 = []
    for _ in range(2):
        C3 = torch.full((P, N), float("nan"), device=d)
        dev.gemm_b2p(A=Ahbl, K=Kd, sm=seq, Wpack=wp8, C_out=C3, ldc=N, a_fmt=3, amax=amax)
        o3.append(C3)
    torch.cuda.synchronize()
    assert torch.equal(o3[0], o3[1])
    print(f"gemm_b2p a_fmt 3 ({view}): vs fp64 {rel(o3[0], ref):.2e} (a_fmt 2: {rel(outs[0], ref):.2e}); vs a_fmt 2 {rel(o3[0], outs[0]):.2e}")
    assert rel(o3[0], ref) < 4e-5 and rel(o3[0], outs[0]) < 4e-5


@pytest.mark.parametrize("view,dims", [("time", (2, 5, 11)), ("band", (3, 32, 37))])
def test_gemm_b2p_bf16_operand(view, dims):
    from wesep_amd import dev
    from wesep_amd.functional import _view_maps
    d = _cuda()
    g = torch.Generator().manual_seed(4)
    R, K, Tf = dims
    P, Kd = R * K * Tf, 2048
    geo, smap, seq, _ = _view_maps(view, R, K, Tf, N)
    nb = dev.bl_num_blocks(seq)
    A = rnd(g, P, Kd).to(torch.bfloat16).float()                  # values on the bf16 grid
    W = rnd(g, N, Kd, scale=0.05)
    wp = torch.empty(N * Kd, device=d)
    dev.pack_w(W.t().contiguous().to(d), N, Kd, N, wp, trans=True, order=1)
    Abl = dev.to_blocked(A.to(d), seq)                            # BL-shaped fp32 (exact bf16 values)
    C_ref, C_new = torch.full((P, N), float("nan"), device=d), torch.full((P, N), float("nan"), device=d)
    dev.gemm_b2p(A=dev.bls_pack(Abl), K=Kd, sm=seq, Wpack=wp, C_out=C_ref, ldc=N)          # lo terms are zero
    dev.gemm_b2p(A=dev.blh_bf16_pack(Abl), K=Kd, sm=seq, Wpack=wp, C_out=C_new, ldc=N, a_fmt=1)
    torch.cuda.synchronize()
    assert torch.equal(C_ref, C_new)
    assert rel(C_new, A.double() @ W.double().t()) < 4e-5


@pytest.mark.parametrize("view,dims", [("time", (2, 5, 11)), ("band", (3, 32, 37)), ("time", (2, 32, 70))])
def test_gemm_tnb_bf16_g_operand(view, dims, monkeypatch):
    """[dW_ih | dW_hh | db] form with G = d(gates) as bf16 (BLH): both directions' column ranges, A1 shifted by one step."""
    from wesep_amd import dev
    from wesep_amd.functional import _view_maps
    d = _cuda()
    g = torch.Generator().manual_seed(5)
    R, K, Tf = dims
    P, GW = R * K * Tf, 2048
    geo, smap, seq, _ = _view_maps(view, R, K, Tf, N)
    nb = dev.bl_num_blocks(seq)
    G = rnd(g, P, GW).to(torch.bfloat16).float()
    A0, A1 = rnd(g, P, N), rnd(g, P, 2 * H)
    Gbl = dev.to_blocked(G.to(d), seq)
    A0b, A1b = dev.to_blocked(A0.to(d), seq, split=True), dev.to_blocked(A1.to(d), seq, split=True)
    G_pairs, G_bf16 = dev.bls_pack(Gbl), dev.blh_bf16_pack(Gbl)
    # g_fmt = 2: the same matrix, were it tiny like a gradient, as scaled fp16.  bf16-grid values are fp16-grid values too.
    amax = torch.tensor([2.7e-6], dtype=torch.float32).view(torch.int32).to(d)
    S = L.dgates_scale(int(amax.item()))
    tiny = 2.0 ** -20
    G_f16 = (Gbl * (tiny * S)).to(torch.float16).contiguous().view(torch.float32)
    assert torch.equal((Gbl * (tiny * S)).to(torch.float16).float(), Gbl * (tiny * S))
    for di, shift in ((0, -1), (1, 1)):
        ns, bps = dev.tnb_splits(nb, 8)
        outs = []
        for Gbuf, fmt, f16mm in ((G_pairs, 0, None), (G_bf16, 1, None), (G_bf16, 1, None), (G_f16, 2, "0"), (G_f16, 2, "1")):
            if f16mm is not None:
                monkeypatch.setenv("WS_TNB_F16", f16mm)   # 0: scaled fp16 on the bf16 instruction (3 terms); 1: on the fp16 one (2)
            slab, bslab = torch.full((ns, 1024 * 384), float("nan"), device=d), torch.full((ns, 1024), float("nan"), device=d)
            dev.gemm_tnb(G=Gbuf, g_width=GW, g_off=di * 1024, g_cols=1024, A0=A0b, a0_width=N, a0_off=0, a0_cols=N,
                         A1=A1b, a1_width=2 * H, a1_off=di * H, a1_cols=H, a1_shift=shift, nblk=nb, L_=seq.L,
                         slab=slab, nsplit=ns, blocks_per_split=bps, bslab=bslab, g_fmt=fmt, amax=amax if fmt == 2 else None)
            outs.append((slab, bslab))
        torch.cuda.synchronize()
        assert torch.equal(outs[0][0], outs[1][0])                # same products, same order: the zero lo terms add nothing
        assert torch.equal(outs[1][0], outs[2][0]) and torch.equal(outs[1][1], outs[2][1])
        assert rel(outs[1][1].sum(0), outs[0][1].sum(0)) < 1e-6   # column sums: pairs of slots per dot2 instead of (hi, lo)
        assert torch.equal(outs[3][0], outs[0][0] * tiny)         # scaled fp16: the same products times an exact power of two
        assert rel(outs[3][1].sum(0), outs[0][1].sum(0) * tiny) < 1e-6
        # fp16 instruction (the default): G A_hi + G A_lo with A lifted by 2^6 -- the same products up to the fp16 denormal
        # range of the lo terms, another summation inside the instruction
        assert rel(outs[4][0].sum(0), outs[0][0].sum(0) * tiny) < 2e-6
        assert rel(outs[4][1].sum(0), outs[0][1].sum(0) * tiny) < 1e-6
        assert rel(outs[1][1].sum(0), G[:, di * 1024:(di + 1) * 1024].double().sum(0)) < 1e-5
        # against fp64 with the step shift of A1 (as in tests/test_kernels_gpu.py::test_gemm_tnb_vs_torch)
        pos, valid = dev.bl_positions(seq, torch.device("cpu"))
        nt = -(-seq.nseq // 32)
        posv, val = pos.view(nt, seq.L, 32), valid.view(nt, seq.L, 32)
        A1s = torch.zeros(P, H)
        src = torch.roll(posv, shifts=-shift, dims=1)
        ok = val.clone()
        if shift == -1:
            ok[:, 0] = False
        else:
            ok[:, -1] = False
        A1s[posv[ok]] = A1[src[ok]][:, di * H:(di + 1) * H]
        Gs = G[:, di * 1024:(di + 1) * 1024].double()
        ref = torch.cat([Gs.t() @ A0.double(), Gs.t() @ A1s.double()], 1)
        assert rel(outs[1][0].sum(0).view(1024, 384), ref) < 4e-5


@pytest.mark.parametrize("fmt", ["f32", "h2s", "h2", "h2b"])
@pytest.mark.parametrize("view", ["time", "band"])
def test_resrnn_formats_agree(view, fmt, monkeypatch):
    """One ResRNN (functional.ResRNNBlkFn) at a geometry that takes the production kernels (time view: cluster forward +
    pair BPTT; band view: fused forward + streaming BPTT) in each storage format: same output bits; gradients of the 2-byte
    formats against the fp32 format."""
    from wesep_amd.models.bsrnn import ResRNN
    d = _cuda()
    torch.manual_seed(5)
    R, K, Tf = (2, 32, 70) if view == "time" else (4, 4, 530)      # band: 2 120 sequences -> the fused 32-sequence forward
    blk = ResRNN(N, 2 * N).to(d)
    z = torch.randn(R, K, Tf, N, device=d)
    go = torch.randn(R, K, Tf, N, device=d)
    from wesep_amd import dev
    res = {}

    def run(f, c2):
        monkeypatch.setenv("WESEP_GATES", f)
        monkeypatch.setenv("WESEP_LSTM_CLUSTER2", c2)
        monkeypatch.setenv("WESEP_FUSED_H16", c2)      # (the band view's counterpart since round 6: fp16 h in the fused forward)
        for p in blk.parameters():
            p.grad = None
        dev.bump_weight_epoch()
        zd = z.clone().requires_grad_(True)
        out = blk(zd, view)
        out.backward(go)
        torch.cuda.synchronize()
        return out.detach().clone(), zd.grad.clone(), {k: p.grad.clone() for k, p in blk.named_parameters()}

    a, b = run("f32", "0"), run(fmt, "0")
    assert torch.equal(a[0], b[0])                                  # the formats change what is STORED, not what is computed
    tol = {"f32": 0.0, "h2s": 5e-5, "h2": 4e-4, "h2b": 3e-3}[fmt]   # unorm16 gates; + scaled-fp16 (default) / bf16 d(gates)
    assert rel(b[1], a[1]) <= tol
    for k in a[2]:
        assert rel(b[2][k], a[2][k]) <= tol, k
    if fmt != "f32":
        # rounds 5 / 6: the 2-byte formats run fp16 h in the recurrent product of the forward by default (time view:
        # ws_lstm_fwd_cluster2; band view: ws_lstm_fused_args.hfmt = 1) -- that changes the forward arithmetic itself (2^-12 on
        # h), so bit-equality becomes a tolerance; everything else as above
        c = run(fmt, "1")
        assert rel(c[0], a[0]) <= 5e-5, rel(c[0], a[0])
        assert rel(c[1], a[1]) <= tol + 2e-4
        for k in a[2]:
            assert rel(c[2][k], a[2][k]) <= tol + 2e-4, k


# ---- ABI v16: fp16 copies of the weight-gradient GEMM's A operand [xn | h] -----------------------------------------------
@pytest.mark.parametrize("view,dims", [("time", (2, 5, 11)), ("band", (3, 32, 37))])
def test_fp16_operand_copies_of_p2b_and_b2p(view, dims):
    """ws_gemm_p2b's A_bl16 = fp16 of the (GroupNorm-on-load) operand it also emits as split pairs; ws_gemm_b2p's a16_out =
    fp16 of its split-pair A operand, bit for bit; neither changes the GEMM's own result."""
    from wesep_amd import dev
    from wesep_amd.functional import _view_maps
    d = _cuda()
    g = torch.Generator().manual_seed(8)
    R, K, Tf = dims
    P = R * K * Tf
    geo, smap, seq, _ = _view_maps(view, R, K, Tf, N)
    nb = dev.bl_num_blocks(seq)
    z = rnd(g, P, N).to(d)
    stats = torch.empty(geo.ngroups, 2, device=d)
    dev.group_stats(z.view(R, K, Tf, N), geo, stats)
    gamma, beta = (1.0 + 0.1 * rnd(g, N)).to(d), (0.1 * rnd(g, N)).to(d)
    W = rnd(g, 256, N, scale=0.05).to(d)
    wp = torch.empty(256 * N, device=d)
    dev.pack_w(W, 256, N, N, wp, order=0)
    outs = []
    for with16 in (False, True):
        C_, xn = torch.full((nb, 32 * 256), float("nan"), device=d), torch.full((nb, 32 * N), float("nan"), device=d)
        xn16 = torch.full((dev.blh_floats(nb, N),), float("nan"), device=d) if with16 else None
        dev.gemm_p2b(A=z, lda=N, sm=seq, Wpack=wp, N=256, C_out=C_, A_bl=xn, stats=stats, gamma=gamma, beta=beta, stat_map=smap,
                     A_bl16=xn16)
        outs.append((C_, xn, xn16))
    torch.cuda.synchronize()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1].view(torch.int32), outs[1][1].view(torch.int32))
    v = dev.bls_unpack(outs[1][1]).view(nb, N // 4, 32, 4)              # the operand as the kernel emitted it (hi + lo)
    a16 = dev.blh_f16_unpack(outs[1][2], nb, N)
    # fp16 of the fp32 operand; the split pair carries 16 of its 24 bits, so compare to rounding, not to the bit
    assert float((a16 - v).abs().max()) <= 2.0 ** -11 * float(v.abs().max()) + 1e-7
    assert float((a16 - v.half().float()).abs().max()) <= 2.0 ** -10 * float(v.abs().max())
    # b2p: hcat-shaped operand (K = 512), its fp16 copy bit for bit
    h = torch.tanh(rnd(g, P, 2 * H)).to(d)
    hb = dev.to_blocked(h, seq, split=True)
    Wp_ = rnd(g, N, 2 * H, scale=0.05).to(d)
    wpp = torch.empty(N * 2 * H, device=d)
    dev.pack_w(Wp_, N, 2 * H, 2 * H, wpp, order=1)
    res = []
    for with16 in (False, True):
        out = torch.full((P, N), float("nan"), device=d)
        h16 = torch.full((dev.blh_floats(nb, 2 * H),), float("nan"), device=d) if with16 else None
        dev.gemm_b2p(A=hb, K=2 * H, sm=seq, Wpack=wpp, C_out=out, ldc=N, a16_out=h16)
        res.append((out, h16))
    torch.cuda.synchronize()
    assert torch.equal(res[0][0], res[1][0])
    want = dev.blh_f16_pack(dev.bls_unpack(hb).view(nb, 2 * H // 4, 32, 4))
    assert torch.equal(res[1][1].view(torch.int32)[: want.numel()], want.reshape(-1).view(torch.int32))


@pytest.mark.parametrize("view,dims", [("time", (2, 5, 11)), ("band", (3, 32, 37)), ("time", (2, 32, 70))])
def test_gemm_tnb_fp16_a_operand(view, dims):
    """ws_gemm_tnb with a_fmt = 1: scaled-fp16 G times fp16 A0 / A1 (one MFMA per product) against the same values as
    split pairs (a_fmt 0: the values are on the fp16 grid, so both forms hold them exactly) and against fp64."""
    from wesep_amd import dev
    from wesep_amd.functional import _view_maps
    d = _cuda()
    g = torch.Generator().manual_seed(9)
    R, K, Tf = dims
    P, GW = R * K * Tf, 2048
    geo, smap, seq, _ = _view_maps(view, R, K, Tf, N)
    nb = dev.bl_num_blocks(seq)
    G = rnd(g, P, GW).to(torch.bfloat16).float()
    A0, A1 = rnd(g, P, N).half().float(), torch.tanh(rnd(g, P, 2 * H)).half().float()
    Gbl = dev.to_blocked(G.to(d), seq)
    A0bl, A1bl = dev.to_blocked(A0.to(d), seq), dev.to_blocked(A1.to(d), seq)
    amax = torch.tensor([2.7e-6], dtype=torch.float32).view(torch.int32).to(d)
    S = L.dgates_scale(int(amax.item()))
    tiny = 2.0 ** -20
    G_f16 = (Gbl * (tiny * S)).to(torch.float16).contiguous().view(torch.float32)
    forms = {0: (dev.bls_pack(A0bl), dev.bls_pack(A1bl)), 1: (dev.blh_f16_pack(A0bl), dev.blh_f16_pack(A1bl))}
    for di, shift in ((0, -1), (1, 1)):
        ns, bps = dev.tnb_splits(nb, 8)
        outs = {}
        for a_fmt in (0, 1, 1):
            slab, bslab = torch.full((ns, 1024 * 384), float("nan"), device=d), torch.full((ns, 1024), float("nan"), device=d)
            dev.gemm_tnb(G=G_f16, g_width=GW, g_off=di * 1024, g_cols=1024, A0=forms[a_fmt][0], a0_width=N, a0_off=0, a0_cols=N,
                         A1=forms[a_fmt][1], a1_width=2 * H, a1_off=di * H, a1_cols=H, a1_shift=shift, nblk=nb, L_=seq.L,
                         slab=slab, nsplit=ns, blocks_per_split=bps, bslab=bslab, g_fmt=2, amax=amax, a_fmt=a_fmt)
            if a_fmt in outs:
                assert torch.equal(outs[a_fmt][0], slab) and torch.equal(outs[a_fmt][1], bslab)     # reproducible
            outs[a_fmt] = (slab, bslab)
        torch.cuda.synchronize()
        assert rel(outs[1][0].sum(0), outs[0][0].sum(0)) < 2e-6
        assert rel(outs[1][1].sum(0), outs[0][1].sum(0)) < 1e-6
        pos, valid = dev.bl_positions(seq, torch.device("cpu"))
        nt = -(-seq.nseq // 32)
        posv, val = pos.view(nt, seq.L, 32), valid.view(nt, seq.L, 32)
        A1s = torch.zeros(P, H)
        src = torch.roll(posv, shifts=-shift, dims=1)
        ok = val.clone()
        if shift == -1:
            ok[:, 0] = False
        else:
            ok[:, -1] = False
        A1s[posv[ok]] = A1[src[ok]][:, di * H:(di + 1) * H]
        Gs = G[:, di * 1024:(di + 1) * 1024].double() * tiny
        ref = torch.cat([Gs.t() @ A0.double(), Gs.t() @ A1s.double()], 1)
        assert rel(outs[1][0].sum(0).view(1024, 384), ref) < 4e-5


def test_gemm_tnb_fp16_operands_vs_fp64_at_the_headline_geometry():
    """the weight-gradient GEMM on fp16 operands (ws_gemm_tnb g_fmt = 2, a_fmt = 1: ONE
    v_mfma_f32_32x32x16_f16 per product, 11-bit G x 11-bit A) at the headline's own contraction length -- the time view of
    R = 32 rows x 4 s: K = P = 513 024 positions -- against the fp64 product of the UNROUNDED fp32 operands, computed on the
    device.  Random operands are the worst case for a rounded product (the roundings of 513 k terms add incoherently and so
    does the sum itself: the relative error does not shrink with K); the bound is 2^-11, the measured figures are printed."""
    from wesep_amd import dev
    from wesep_amd.functional import _view_maps
    d = _cuda()
    torch.manual_seed(21)
    R, K, Tf = 32, 32, 501
    P, GW = R * K * Tf, 2048
    geo, smap, seq, _ = _view_maps("time", R, K, Tf, N)
    nb = dev.bl_num_blocks(seq)
    G32 = torch.randn(P, GW, device=d) * torch.exp(torch.randn(P, 1, device=d)) * 1e-6     # gradient-sized, heavy-tailed rows
    A0 = torch.randn(P, N, device=d)                                                       # the normalised input
    A1 = torch.tanh(torch.randn(P, 2 * H, device=d))                                       # h in (-1, 1)
    amax = (G32.abs().max() / 4).reshape(1).view(torch.int32).clone()                      # d(gates) reach a few times max |d(hcat)|
    S = L.dgates_scale(int(amax.item()))
    assert float(G32.abs().max()) * S < 65504.0
    G_f16 = (dev.to_blocked(G32, seq) * S).to(torch.float16).contiguous().view(torch.float32)
    A0h, A1h = dev.blh_f16_pack(dev.to_blocked(A0, seq)), dev.blh_f16_pack(dev.to_blocked(A1, seq))
    pos, valid = dev.bl_positions(seq, d)
    nt = -(-seq.nseq // 32)
    posv, val = pos.view(nt, seq.L, 32), valid.view(nt, seq.L, 32)
    worst = 0.0
    for di, shift in ((0, -1), (1, 1)):
        ns, bps = dev.tnb_splits(nb, 8)
        slab, bslab = torch.full((ns, 1024 * 384), float("nan"), device=d), torch.full((ns, 1024), float("nan"), device=d)
        dev.gemm_tnb(G=G_f16, g_width=GW, g_off=di * 1024, g_cols=1024, A0=A0h, a0_width=N, a0_off=0, a0_cols=N,
                     A1=A1h, a1_width=2 * H, a1_off=di * H, a1_cols=H, a1_shift=shift, nblk=nb, L_=seq.L,
                     slab=slab, nsplit=ns, blocks_per_split=bps, bslab=bslab, g_fmt=2, amax=amax, a_fmt=1)
        got, gotb = slab.sum(0).view(1024, 384).double(), bslab.sum(0).double()
        A1s = torch.zeros(P, H, device=d)                              # h of the previous step of the same sequence
        src = torch.roll(posv, shifts=-shift, dims=1)
        ok = val.clone()
        if shift == -1:
            ok[:, 0] = False
        else:
            ok[:, -1] = False
        A1s[posv[ok]] = A1[src[ok]][:, di * H:(di + 1) * H]
        ref = torch.zeros(1024, 384, device=d, dtype=torch.float64)
        refb = torch.zeros(1024, device=d, dtype=torch.float64)
        for lo in range(0, P, 32768):                                  # fp64 on the device, in slices of the contraction
            g64 = G32[lo:lo + 32768, di * 1024:(di + 1) * 1024].double()
            ref += g64.t() @ torch.cat([A0[lo:lo + 32768], A1s[lo:lo + 32768]], 1).double()
            refb += g64.sum(0)
        e_ih, e_hh, e_b = rel(got[:, :N], ref[:, :N]), rel(got[:, N:], ref[:, N:]), rel(gotb, refb)
        e_max = float((got - ref).abs().max() / ref.abs().max())
        print(f"gemm_tnb fp16 x fp16 at K = {P}, direction {di}: rel-L2 dW_ih {e_ih:.2e}  dW_hh {e_hh:.2e}  db {e_b:.2e}; "
              f"largest element error / largest element {e_max:.2e}")
        worst = max(worst, e_ih, e_hh, e_b)
        del g64, ref
    assert worst < 2.0 ** -11, worst
