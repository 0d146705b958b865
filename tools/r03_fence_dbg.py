"""Which victim sizes / aggressor repeat counts make the round-2 victims fail beside gemm_b2p (packed build)?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_cross_stream_gpu as X
from wesep_amd import dev, _lib
d = torch.device("cuda:0")
print("lib", _lib.LIB_PATH)
aggr, C, Cref = X._aggressor(d)
from oracle.bsrnn_oracle import band_widths
bt = dev.BandTables(band_widths(16000, 512), d)
s0, s1 = torch.cuda.Stream(device=d), torch.cuda.Stream(device=d)
for R, T in ((2, 24000), (16, 64000), (64, 64000)):
    Tf = 1 + T // 128
    wav = (torch.randn(R, T) * 0.1).to(d)
    xbs = torch.empty(R * Tf, 514, device=d)
    dev.stft_bandsplit(wav, bt, xbs); torch.cuda.synchronize()
    ref = xbs.clone()
    for nag, nvic in ((1, 1), (4, 1), (4, 8), (16, 32)):
        bad = 0
        for _ in range(30):
            with torch.cuda.stream(s0):
                for _ in range(nag): aggr()
            with torch.cuda.stream(s1):
                for _ in range(nvic): dev.stft_bandsplit(wav, bt, xbs)
            torch.cuda.synchronize()
            bad += int(not torch.equal(xbs, ref))
        print(f"stft R={R} T={T}: {nag} aggressor x {nvic} victim launches: {bad}/30 differ", flush=True)
