// STFT / iSTFT for pBSRNN on gfx950 (torch.stft / torch.istft at wesep/models/bsrnn.py:309-316,
// 382-389: n_fft 512, hop 128, periodic Hann, center=True with reflect padding), fused with
// the band split (bsrnn.py:319-328) on the way in and with the GLU complex-mask apply
// (bsrnn.py:366-381) on the way out.  HBM-bound byte shuffling: one WAVE owns one frame,
// 8 points per lane, three radix-8 passes (512 = 8*8*8) exchanged through LDS, twiddles from
// an LDS table built once per workgroup.  Frame loads are 64 consecutive floats per
// instruction (coalesced); every sample is re-read by 4 overlapping frames out of L2.
#include "common.h"

#define NFFT 512
#define HOP 128
#define NBIN 257
#define FR_PER_WG 16  // 4 waves x 4 frames

struct cpx {
  float x, y;
};
__device__ __forceinline__ cpx cadd(cpx a, cpx b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cpx csub(cpx a, cpx b) { return {a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ cpx cmul(cpx a, cpx b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ cpx mul_mi(cpx a) { return {a.y, -a.x}; }  // a * (-i)

// forward 8-point DFT in place: v[k] = sum_n v[n] e^{-2 pi i n k / 8}
__device__ __forceinline__ void dft8(cpx (&v)[8]) {
  const float h = 0.70710678118654752440f;
  cpx a0 = cadd(v[0], v[4]), a4 = csub(v[0], v[4]);
  cpx a1 = cadd(v[1], v[5]), a5 = csub(v[1], v[5]);
  cpx a2 = cadd(v[2], v[6]), a6 = csub(v[2], v[6]);
  cpx a3 = cadd(v[3], v[7]), a7 = csub(v[3], v[7]);
  a5 = cmul(a5, cpx{h, -h});   // W8^1
  a6 = mul_mi(a6);             // W8^2
  a7 = cmul(a7, cpx{-h, -h});  // W8^3
  cpx b0 = cadd(a0, a2), b2 = csub(a0, a2), b1 = cadd(a1, a3), b3 = mul_mi(csub(a1, a3));
  cpx c0 = cadd(a4, a6), c2 = csub(a4, a6), c1 = cadd(a5, a7), c3 = mul_mi(csub(a5, a7));
  v[0] = cadd(b0, b1);
  v[4] = csub(b0, b1);
  v[2] = cadd(b2, b3);
  v[6] = csub(b2, b3);
  v[1] = cadd(c0, c1);
  v[5] = csub(c0, c1);
  v[3] = cadd(c2, c3);
  v[7] = csub(c2, c3);
}

// LDS per wave: S1 [8][72] + S2 [64][9] complex
#define S1_LD 72
#define S2_LD 9
#define WAVE_SCRATCH (8 * S1_LD + 64 * S2_LD)

// One wave: in  v[n1] = x[64*n1 + lane], out v[k3] = X[lane + 64*k3], X[k] = sum_n x[n] e^{-2 pi i n k/512}.
// Contains two __syncthreads(): every wave of the workgroup must call it together.
__device__ __forceinline__ void fft512(cpx (&v)[8], cpx* sc, const cpx* tw, int lane) {
  cpx* S1 = sc;
  cpx* S2 = sc + 8 * S1_LD;
  // pass 1: over n1, lane = 8*n2 + n3; twiddle W512^{(8 n2 + n3) k1}
  dft8(v);
#pragma unroll
  for (int k1 = 1; k1 < 8; ++k1) v[k1] = cmul(v[k1], tw[lane * k1]);
#pragma unroll
  for (int k1 = 0; k1 < 8; ++k1) S1[k1 * S1_LD + lane] = v[k1];
  __syncthreads();
  // pass 2: lane = 8*k1 + n3, over n2; twiddle W64^{n3 k2} = W512^{8 n3 k2}
  const int k1 = lane >> 3, n3 = lane & 7;
#pragma unroll
  for (int n2 = 0; n2 < 8; ++n2) v[n2] = S1[k1 * S1_LD + n2 * 8 + n3];
  dft8(v);
#pragma unroll
  for (int k2 = 1; k2 < 8; ++k2) v[k2] = cmul(v[k2], tw[8 * n3 * k2]);
#pragma unroll
  for (int k2 = 0; k2 < 8; ++k2) S2[(k1 + 8 * k2) * S2_LD + n3] = v[k2];
  __syncthreads();
  // pass 3: lane = k1 + 8*k2, over n3 -> X[lane + 64*k3]
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = S2[lane * S2_LD + j];
  dft8(v);
}

__device__ __forceinline__ void build_twiddles(cpx* tw) {
  for (int j = threadIdx.x; j < NFFT; j += blockDim.x) {
    float s, c;
    sincospif((float)j * (1.0f / 256.0f), &s, &c);  // angle 2 pi j / 512
    tw[j] = cpx{c, -s};
  }
}
__device__ __forceinline__ float hann(const cpx* tw, int n) { return 0.5f - 0.5f * tw[n].x; }

// window-envelope of torch.istft at padded coordinate q (= sample + 256)
__device__ __forceinline__ float ola_envelope(const cpx* tw, int q, int Tf) {
  const int t_hi = min(q / HOP, Tf - 1);
  const int t_lo = q >= NFFT ? (q - (NFFT - HOP)) / HOP : 0;
  float e = 0.f;
  for (int t = t_lo; t <= t_hi; ++t) {
    const float w = hann(tw, q - HOP * t);
    e += w * w;
  }
  return e;
}

// -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stft_bandsplit_kernel(const float* __restrict__ wav, int R,
                                                             int T, int Tf, const ws_bands b,
                                                             float* __restrict__ xbs) {
  __shared__ cpx tw[NFFT];
  __shared__ cpx scratch[4 * WAVE_SCRATCH];
  build_twiddles(tw);
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  cpx* sc = scratch + wave * WAVE_SCRATCH;
  const int nframes = R * Tf;
  for (int it = 0; it < FR_PER_WG / 4; ++it) {
    const int f = blockIdx.x * FR_PER_WG + it * 4 + wave;
    const bool active = f < nframes;
    const int ff = active ? f : 0;
    const int r = ff / Tf, t = ff - r * Tf;
    cpx v[8];
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1) {
      const int n = 64 * n1 + lane;
      int pos = t * HOP + n - NFFT / 2;
      if (pos < 0) pos = -pos;
      if (pos >= T) pos = 2 * (T - 1) - pos;
      v[n1] = cpx{wav[(long long)r * T + pos] * hann(tw, n), 0.f};
    }
    fft512(v, sc, tw, lane);
    if (active) {
      float* row = xbs + (long long)f * (2 * NBIN);
#pragma unroll
      for (int k3 = 0; k3 < 5; ++k3) {
        const int bin = lane + 64 * k3;
        if (bin < NBIN) {
          const int g = b.band_of_bin[bin];
          const int f0 = b.band_f0[g], bw = b.band_bw[g];
          row[2 * f0 + (bin - f0)] = v[k3].x;
          row[2 * f0 + bw + (bin - f0)] = v[k3].y;
        }
      }
    }
  }
}

extern "C" int ws_stft_bandsplit(const float* wav, int R, int T, const ws_bands* b, float* xbs,
                                 void* stream) {
  WS_REQUIRE(wav && b && xbs && R > 0, "ws_stft_bandsplit: bad args");
  WS_REQUIRE(T > NFFT / 2, "ws_stft_bandsplit: T=%d must exceed the reflect pad %d", T, NFFT / 2);
  WS_REQUIRE(b->nbins == NBIN && b->band_of_bin && b->band_f0 && b->band_bw, "ws_stft_bandsplit: bad bands");
  const int Tf = 1 + T / HOP;
  const int nframes = R * Tf;
  hipLaunchKernelGGL(stft_bandsplit_kernel, dim3((nframes + FR_PER_WG - 1) / FR_PER_WG), dim3(256),
                     0, (hipStream_t)stream, wav, R, T, Tf, *b, xbs);
  return ws_check_launch("ws_stft_bandsplit");
}

// -------------------------------------------------------------------------------------------
// mask apply + inverse real FFT + synthesis window -> frames [R*Tf][512]
// -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mask_istft_frames_kernel(const float* __restrict__ xbs,
                                                                const float* __restrict__ m3,
                                                                int nframes, const ws_bands b,
                                                                float* __restrict__ frames) {
  __shared__ cpx tw[NFFT];
  __shared__ cpx scratch[4 * WAVE_SCRATCH];
  build_twiddles(tw);
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  cpx* sc = scratch + wave * WAVE_SCRATCH;
  for (int it = 0; it < FR_PER_WG / 4; ++it) {
    const int f = blockIdx.x * FR_PER_WG + it * 4 + wave;
    const bool active = f < nframes;
    const long long ff = active ? f : 0;
    const float* xr = xbs + ff * (2 * NBIN);
    const float* mr = m3 + ff * (4 * NBIN);
    cpx v[8];
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1) {
      const int bin = 64 * n1 + lane;
      const int src = bin <= NFFT / 2 ? bin : NFFT - bin;
      const int g = b.band_of_bin[src];
      const int f0 = b.band_f0[g], bw = b.band_bw[g], fl = src - f0;
      const float Xr = xr[2 * f0 + fl], Xi = xr[2 * f0 + bw + fl];
      const float* o = mr + 4 * f0 + fl;
      const float mre = o[0] * ws_sigmoid(o[2 * bw]);
      const float mim = o[bw] * ws_sigmoid(o[3 * bw]);
      const float er = Xr * mre - Xi * mim;
      float ei = Xr * mim + Xi * mre;
      if (src == 0 || src == NFFT / 2) ei = 0.f;  // c2r ignores Im of DC / Nyquist
      // Hermitian extension Y[512-k] = conj(Y[k]); inverse FFT = conj(FFT(conj(Y))) / N
      v[n1] = cpx{er, bin <= NFFT / 2 ? -ei : ei};
    }
    fft512(v, sc, tw, lane);
    if (active) {
#pragma unroll
      for (int k3 = 0; k3 < 8; ++k3) {
        const int n = lane + 64 * k3;
        frames[ff * NFFT + n] = v[k3].x * (1.0f / NFFT) * hann(tw, n);
      }
    }
  }
}

extern "C" int ws_mask_istft_frames(const float* xbs, const float* mask3, int R, int Tf,
                                    const ws_bands* b, float* frames, void* stream) {
  WS_REQUIRE(xbs && mask3 && b && frames && R > 0 && Tf > 0, "ws_mask_istft_frames: bad args");
  WS_REQUIRE(b->nbins == NBIN, "ws_mask_istft_frames: bad bands");
  const int nframes = R * Tf;
  hipLaunchKernelGGL(mask_istft_frames_kernel, dim3((nframes + FR_PER_WG - 1) / FR_PER_WG),
                     dim3(256), 0, (hipStream_t)stream, xbs, mask3, nframes, *b, frames);
  return ws_check_launch("ws_mask_istft_frames");
}

__global__ __launch_bounds__(256) void istft_ola_kernel(const float* __restrict__ frames, int R,
                                                        int Tf, int T, float* __restrict__ wav) {
  __shared__ cpx tw[NFFT];
  build_twiddles(tw);
  __syncthreads();
  const long long total = (long long)R * T;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(i / T), pos = (int)(i - (long long)r * T);
    const int q = pos + NFFT / 2;
    const int t_hi = min(q / HOP, Tf - 1);
    const int t_lo = q >= NFFT ? (q - (NFFT - HOP)) / HOP : 0;
    float y = 0.f, e = 0.f;
    for (int t = t_lo; t <= t_hi; ++t) {
      const int n = q - HOP * t;
      y += frames[((long long)r * Tf + t) * NFFT + n];
      const float w = hann(tw, n);
      e += w * w;
    }
    wav[i] = y / e;
  }
}

extern "C" int ws_istft_ola(const float* frames, int R, int Tf, int T, float* wav, void* stream) {
  WS_REQUIRE(frames && wav && R > 0 && Tf > 0 && T > 0, "ws_istft_ola: bad args");
  WS_REQUIRE(Tf == 1 + T / HOP, "ws_istft_ola: Tf=%d does not match T=%d", Tf, T);
  long long blocks = ((long long)R * T + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(istft_ola_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     frames, R, Tf, T, wav);
  return ws_check_launch("ws_istft_ola");
}

// -------------------------------------------------------------------------------------------
// backward of (mask apply -> irfft -> window -> OLA/envelope): dwav -> dmask3
// -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mask_istft_bwd_kernel(const float* __restrict__ dwav,
                                                             const float* __restrict__ xbs,
                                                             const float* __restrict__ m3, int R,
                                                             int Tf, int T, const ws_bands b,
                                                             float* __restrict__ dm3) {
  __shared__ cpx tw[NFFT];
  __shared__ cpx scratch[4 * WAVE_SCRATCH];
  build_twiddles(tw);
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  cpx* sc = scratch + wave * WAVE_SCRATCH;
  const int nframes = R * Tf;
  for (int it = 0; it < FR_PER_WG / 4; ++it) {
    const int f = blockIdx.x * FR_PER_WG + it * 4 + wave;
    const bool active = f < nframes;
    const int ff = active ? f : 0;
    const int r = ff / Tf, t = ff - r * Tf;
    cpx v[8];
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1) {
      const int n = 64 * n1 + lane;
      const int pos = t * HOP + n - NFFT / 2;
      float dv = 0.f;
      if (pos >= 0 && pos < T)
        dv = dwav[(long long)r * T + pos] * hann(tw, n) / ola_envelope(tw, pos + NFFT / 2, Tf);
      v[n1] = cpx{dv, 0.f};
    }
    fft512(v, sc, tw, lane);
    if (active) {
      const float* xr = xbs + (long long)f * (2 * NBIN);
      const float* mr = m3 + (long long)f * (4 * NBIN);
      float* dr = dm3 + (long long)f * (4 * NBIN);
#pragma unroll
      for (int k3 = 0; k3 < 5; ++k3) {
        const int bin = lane + 64 * k3;
        if (bin < NBIN) {
          const bool edge = bin == 0 || bin == NFFT / 2;
          const float sc_ = (edge ? 1.0f : 2.0f) / NFFT;
          const float gre = v[k3].x * sc_;
          const float gim = edge ? 0.f : v[k3].y * sc_;
          const int g = b.band_of_bin[bin];
          const int f0 = b.band_f0[g], bw = b.band_bw[g], fl = bin - f0;
          const float Xr = xr[2 * f0 + fl], Xi = xr[2 * f0 + bw + fl];
          const float* o = mr + 4 * f0 + fl;
          const float o00 = o[0], o01 = o[bw], s0 = ws_sigmoid(o[2 * bw]), s1 = ws_sigmoid(o[3 * bw]);
          const float dmr = gre * Xr + gim * Xi;
          const float dmi = -gre * Xi + gim * Xr;
          float* d = dr + 4 * f0 + fl;
          d[0] = dmr * s0;
          d[bw] = dmi * s1;
          d[2 * bw] = dmr * o00 * s0 * (1.f - s0);
          d[3 * bw] = dmi * o01 * s1 * (1.f - s1);
        }
      }
    }
  }
}

extern "C" int ws_mask_istft_bwd(const float* dwav, const float* xbs, const float* mask3, int R,
                                 int Tf, int T, const ws_bands* b, float* dmask3, void* stream) {
  WS_REQUIRE(dwav && xbs && mask3 && b && dmask3 && R > 0, "ws_mask_istft_bwd: bad args");
  WS_REQUIRE(Tf == 1 + T / HOP && b->nbins == NBIN, "ws_mask_istft_bwd: bad Tf/bands");
  const int nframes = R * Tf;
  hipLaunchKernelGGL(mask_istft_bwd_kernel, dim3((nframes + FR_PER_WG - 1) / FR_PER_WG), dim3(256),
                     0, (hipStream_t)stream, dwav, xbs, mask3, R, Tf, T, *b, dmask3);
  return ws_check_launch("ws_mask_istft_bwd");
}
