"""Pins the fence for the cross-stream disturbance of profiles/r02_kernel_race.md / r03_kernel_race.md.

Round 3 root cause (standalone reproducer tools/cbench/race_repro.hip): on the MI355X a packed FP32 instruction whose
src1 selects the other half (v_pk_{add,mul,fma}_f32 ... op_sel:[0,1]) returns wrong LOW halves while gemm_b2p runs on
the same CU from another stream.  The library is therefore built without packed FP32 arithmetic
(wesep_amd/build.py NO_PACKED_FP32).  These tests run the kernels round 2 found to be victims -- stft_bandsplit, the
fp32 gemm_nt, mask_istft_frames -- beside gemm_b2p on a second stream and require bit-identical results: with the
round-2 build they fail in 20-29 of 30 trials (profiles/r02_kernel_race.md), with the fence they must never."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
TRIALS = 30


def _cuda():
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    return torch.device("cuda:0")


def _aggressor(d):
    """gemm_b2p at the band-view shape of the training step (8192 sequences x 41 steps, K = 256): ~0.18 ms a launch."""
    from wesep_amd import dev
    nseq, L, K, N = 8192, 41, 256, 128
    g = torch.Generator().manual_seed(1)
    A = (torch.randn(nseq * L * K, generator=g) * 0.5).to(d)       # any bit pattern is a valid BLS operand
    W = (torch.randn(N, K, generator=g) * 0.05).to(d)
    wp = torch.empty(N * K, device=d)
    dev.pack_w(W, N, K, K, wp, order=1)
    C = torch.empty(nseq * L, N, device=d)
    seq = dev.SeqMap(nseq=nseq, div=1 << 30, s1=0, s2=L, step_rows=1, L=L)

    def launch():
        dev.gemm_b2p(A=A, K=K, sm=seq, Wpack=wp, C_out=C, ldc=N)
    launch()
    torch.cuda.synchronize()
    return launch, C, C.clone()          # (arbitrary operand bits: C may hold NaNs, compare it as integers)


def _victims(d):
    from oracle.bsrnn_oracle import band_widths
    from wesep_amd import dev
    bw = band_widths(16000, 512)
    bt = dev.BandTables(bw, d)
    g = torch.Generator().manual_seed(2)
    R, T = 16, 64000
    Tf = 1 + T // 128
    wav = (torch.randn(R, T, generator=g) * 0.1).to(d)
    xbs = torch.empty(R * Tf, 514, device=d)
    out = {}

    def stft():
        dev.stft_bandsplit(wav, bt, xbs)
        return xbs
    out["stft_bandsplit"] = stft
    M, Nn, Kk = 8192, 128, 512
    A = torch.randn(M, Kk, generator=g).to(d)
    W = (torch.randn(Nn, Kk, generator=g) * 0.05).to(d)
    Cc = torch.empty(M, Nn, device=d)

    def gemm():
        dev.gemm_nt(A=A, a_rows=dev.flat(Kk), M=M, N=Nn, K=Kk, W=W, ldw=Kk, C_out=Cc, c_rows=dev.flat(Nn), vec=3,
                    mode="f32")
        return Cc
    out["gemm_nt_f32"] = gemm
    return out


def test_library_has_no_packed_fp32_arithmetic():
    """The fence itself: not one v_pk_{mul,fma,add}_f32 in the code objects this process loaded."""
    from wesep_amd import _lib
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_no_packed_fp32.py"), _lib.LIB_PATH],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("victim", ["stft_bandsplit", "gemm_nt_f32"])
def test_round2_victims_are_bit_identical_beside_gemm_b2p(victim):
    d = _cuda()
    aggr, C, Cref = _aggressor(d)
    run = _victims(d)[victim]
    ref = run().clone()
    torch.cuda.synchronize()
    assert torch.equal(run(), ref)                                  # alone: reproducible
    s0, s1 = torch.cuda.Stream(device=d), torch.cuda.Stream(device=d)
    bad = 0
    for _ in range(TRIALS):
        with torch.cuda.stream(s0):
            for _ in range(4):
                aggr()
        with torch.cuda.stream(s1):
            got = run()
        torch.cuda.synchronize()
        bad += int(not torch.equal(got, ref))
    assert bad == 0, f"{victim}: {bad} of {TRIALS} launches beside gemm_b2p differ from the launch alone"
    assert torch.equal(C.view(torch.int32), Cref.view(torch.int32))


def test_standalone_reproducer_still_shows_the_hardware_behaviour():
    """Documents (does not fence) the platform behaviour: the asm victim with op_sel on src1 is disturbed, the same
    instruction without op_sel is not.  Skipped when the reproducer binary was not built."""
    _cuda()
    exe = os.path.join(ROOT, "tools", "cbench", "race_repro")
    if not os.path.exists(exe):
        pytest.skip("tools/cbench/race_repro not built")
    r = subprocess.run([exe, "--trials", "4", "--victims", "o_pk_add,a_pk_add,ring_step"], capture_output=True, text=True,
                       timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = {ln.split()[0]: ln for ln in r.stdout.splitlines() if ln and not ln.startswith("#")}
    assert " 0 of 4 trials" in lines["a_pk_add"] and " 0 of 4 trials" in lines["ring_step"], r.stdout
    if " 0 of 4 trials" in lines["o_pk_add"]:
        pytest.xfail("op_sel victim clean on this box: the platform behaviour did not reproduce here")


def test_standalone_aggressor_pair_differs_by_one_wait_count():
    """Documents round 4's finding on the aggressor side (profiles/r04_kernel_race.md): two from-scratch kernels of the
    reproducer that differ only in the order of three MFMAs -- own35 waits lgkmcnt(0) before the first MFMA of a group,
    own99 issues it while the second ds_read_b128 is still in flight -- leave the op_sel victim intact / disturb it.
    Documentary: xfail when the platform behaviour does not show on a box."""
    _cuda()
    exe = os.path.join(ROOT, "tools", "cbench", "race_repro")
    if not os.path.exists(exe):
        pytest.skip("tools/cbench/race_repro not built")
    out = {}
    for aggr in ("own35", "own99"):
        r = subprocess.run([exe, "--trials", "4", "--aggr", aggr, "--victims", "o_pk_add"], capture_output=True, text=True,
                           timeout=120)
        assert r.returncode == 0, r.stderr[-2000:]
        out[aggr] = next(ln for ln in r.stdout.splitlines() if ln.startswith("o_pk_add"))
    # documentary on both sides: hardware behaviour, not a property of this library -- never a hard failure of the suite
    if " 0 of 4 trials" not in out["own35"]:
        pytest.xfail("own35 (MFMAs issued with no LDS read in flight) disturbed the victim on this box: " + out["own35"][:200])
    if " 0 of 4 trials" in out["own99"]:
        pytest.xfail("own99 did not disturb the op_sel victim on this box")
