// 2-D convolution pieces of the wespeaker ResNet speaker encoder (the model source is
// a third-party dependency absent from the reference tree, call sites wesep/models/bsrnn.py:9,217,352-356), on
// CHANNELS-LAST activations [R][H][W][C]:
//   conv2d(k x k, stride s, padding p, no bias) = im2col (this file) + one row-major GEMM with K = k*k*Cin
//   (gemm*.hip, split-bf16 MFMA); its input gradient = GEMM + col2im (a GATHER over the <= k*k patches that
//   contain a pixel: deterministic, no atomics); its weight gradient = the TN GEMM on the same patch matrix.
// Plus the TSTP pooling (mean || unbiased std over time) and its backward.
#include "common.h"

struct ConvGeom {
  int R, H, W, C;   // input  [R][H][W][C]
  int Ho, Wo;       // output spatial size
  int k, s, p;      // kernel, stride, padding
};

// patches[m][(ky*k + kx)*C + c] = x[r][ho*s + ky - p][wo*s + kx - p][c] (0 outside), m = (r*Ho + ho)*Wo + wo
__global__ void im2col_kernel(const float* __restrict__ x, ConvGeom g, float* __restrict__ patches) {
  const int c4n = g.C >> 2, kk = g.k * g.k;
  const long long total = (long long)g.R * g.Ho * g.Wo * kk * c4n;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % c4n) * 4;
    long long q = i / c4n;
    const int tap = (int)(q % kk);
    q /= kk;
    const int wo = (int)(q % g.Wo);
    q /= g.Wo;
    const int ho = (int)(q % g.Ho), r = (int)(q / g.Ho);
    const int hi = ho * g.s + tap / g.k - g.p, wi = wo * g.s + tap % g.k - g.p;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (hi >= 0 && hi < g.H && wi >= 0 && wi < g.W)
      v = *reinterpret_cast<const f32x4*>(x + (((long long)r * g.H + hi) * g.W + wi) * g.C + c);
    *reinterpret_cast<f32x4*>(patches + i * 4) = v;
  }
}

// single input channel (the first layer): one thread per patch element
__global__ void im2col_c1_kernel(const float* __restrict__ x, ConvGeom g, int ldp, float* __restrict__ patches) {
  const int kk = g.k * g.k;
  const long long total = (long long)g.R * g.Ho * g.Wo * kk;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int tap = (int)(i % kk);
    long long q = i / kk;
    const int wo = (int)(q % g.Wo);
    const long long m = q;
    q /= g.Wo;
    const int ho = (int)(q % g.Ho), r = (int)(q / g.Ho);
    const int hi = ho * g.s + tap / g.k - g.p, wi = wo * g.s + tap % g.k - g.p;
    float v = 0.f;
    if (hi >= 0 && hi < g.H && wi >= 0 && wi < g.W) v = x[((long long)r * g.H + hi) * g.W + wi];
    patches[m * ldp + tap] = v;
  }
}

// dx[r][hi][wi][c] = sum over taps (ky, kx) with (hi + p - ky) % s == 0, (wi + p - kx) % s == 0 and the output
// position in range of dpatches[m(ho, wo)][(ky*k + kx)*C + c]
__global__ void col2im_kernel(const float* __restrict__ dpatches, ConvGeom g, float* __restrict__ dx) {
  const int c4n = g.C >> 2, kk = g.k * g.k;
  const long long total = (long long)g.R * g.H * g.W * c4n;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % c4n) * 4;
    long long q = i / c4n;
    const int wi = (int)(q % g.W);
    q /= g.W;
    const int hi = (int)(q % g.H), r = (int)(q / g.H);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int ky = 0; ky < g.k; ++ky) {
      const int hn = hi + g.p - ky;
      if (hn < 0 || hn % g.s) continue;
      const int ho = hn / g.s;
      if (ho >= g.Ho) continue;
      for (int kx = 0; kx < g.k; ++kx) {
        const int wn = wi + g.p - kx;
        if (wn < 0 || wn % g.s) continue;
        const int wo = wn / g.s;
        if (wo >= g.Wo) continue;
        const long long m = ((long long)r * g.Ho + ho) * g.Wo + wo;
        acc += *reinterpret_cast<const f32x4*>(dpatches + (m * kk + ky * g.k + kx) * g.C + c);
      }
    }
    *reinterpret_cast<f32x4*>(dx + i * 4) = acc;
  }
}

static inline unsigned cv_blocks(long long n) {
  long long b = (n + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > 65536 ? 65536 : b));
}

static int cv_check(const char* who, int R, int H, int W, int C, int k, int s, int p, int* Ho, int* Wo) {
  WS_REQUIRE(R > 0 && H > 0 && W > 0 && C > 0 && k >= 1 && s >= 1 && p >= 0, "%s: bad geometry", who);
  *Ho = (H + 2 * p - k) / s + 1;
  *Wo = (W + 2 * p - k) / s + 1;
  WS_REQUIRE(*Ho > 0 && *Wo > 0, "%s: empty output", who);
  return WS_OK;
}

extern "C" int ws_im2col(const float* x, int R, int H, int W, int C, int k, int s, int p, long long ldp,
                         float* patches, void* stream) {
  int Ho, Wo;
  int rc = cv_check("ws_im2col", R, H, W, C, k, s, p, &Ho, &Wo);
  if (rc != WS_OK) return rc;
  WS_REQUIRE(x && patches, "ws_im2col: null pointer");
  const ConvGeom g{R, H, W, C, Ho, Wo, k, s, p};
  if (C == 1) {
    WS_REQUIRE(ldp >= k * k, "ws_im2col: ldp < k*k");
    hipLaunchKernelGGL(im2col_c1_kernel, dim3(cv_blocks((long long)R * Ho * Wo * k * k)), dim3(256), 0,
                       (hipStream_t)stream, x, g, (int)ldp, patches);
  } else {
    WS_REQUIRE(C % 4 == 0 && ldp == (long long)k * k * C, "ws_im2col: C %% 4 and ldp == k*k*C for C > 1");
    hipLaunchKernelGGL(im2col_kernel, dim3(cv_blocks((long long)R * Ho * Wo * k * k * (C / 4))), dim3(256), 0,
                       (hipStream_t)stream, x, g, patches);
  }
  return ws_check_launch("ws_im2col");
}

extern "C" int ws_col2im(const float* dpatches, int R, int H, int W, int C, int k, int s, int p, float* dx,
                         void* stream) {
  int Ho, Wo;
  int rc = cv_check("ws_col2im", R, H, W, C, k, s, p, &Ho, &Wo);
  if (rc != WS_OK) return rc;
  WS_REQUIRE(dpatches && dx && C % 4 == 0, "ws_col2im: null pointer / C %% 4");
  const ConvGeom g{R, H, W, C, Ho, Wo, k, s, p};
  hipLaunchKernelGGL(col2im_kernel, dim3(cv_blocks((long long)R * H * W * (C / 4))), dim3(256), 0, (hipStream_t)stream,
                     dpatches, g, dx);
  return ws_check_launch("ws_col2im");
}

// ---------------------------------------------------------------------------------------------
// TSTP (temporal statistics pooling): x [R][F][T][C] -> stats [R][2][C*F]: mean over T and
// sqrt(unbiased var over T + 1e-7), feature index c*F + f (the reference flattens [C][F]).
// One thread per (r, f, c), two passes over T.
// ---------------------------------------------------------------------------------------------
__global__ void tstp_fwd_kernel(const float* __restrict__ x, int R, int F, int T, int C, float eps,
                                float* __restrict__ stats) {
  const long long total = (long long)R * F * C;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long long q = i / C;
    const int f = (int)(q % F), r = (int)(q / F);
    const float* b = x + (((long long)r * F + f) * T) * C + c;
    float s = 0.f;
    for (int t = 0; t < T; ++t) s += b[(long long)t * C];
    const float mean = s / (float)T;
    float m2 = 0.f;
    for (int t = 0; t < T; ++t) {
      const float dv = b[(long long)t * C] - mean;
      m2 += dv * dv;
    }
    const float var = T > 1 ? m2 / (float)(T - 1) : 0.f;
    float* o = stats + (long long)r * 2 * C * F;
    o[c * F + f] = mean;
    o[C * F + c * F + f] = sqrtf(var + eps);
  }
}

// dx[r][f][t][c] = dmean / T + dstd * (x - mean) / ((T - 1) * std)
__global__ void tstp_bwd_kernel(const float* __restrict__ x, const float* __restrict__ stats,
                                const float* __restrict__ dstats, int R, int F, int T, int C,
                                float* __restrict__ dx) {
  const long long total = (long long)R * F * T * C;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    long long q = i / C;
    q /= T;
    const int f = (int)(q % F), r = (int)(q / F);
    const long long o = (long long)r * 2 * C * F + c * F + f;
    const float mean = stats[o], sd = stats[o + (long long)C * F];
    const float gm = dstats[o], gs = dstats[o + (long long)C * F];
    dx[i] = gm / (float)T + (T > 1 ? gs * (x[i] - mean) / ((float)(T - 1) * sd) : 0.f);
  }
}

extern "C" int ws_tstp_fwd(const float* x, int R, int F, int T, int C, float eps, float* stats, void* stream) {
  WS_REQUIRE(x && stats && R > 0 && F > 0 && T > 0 && C > 0, "ws_tstp_fwd: bad args");
  hipLaunchKernelGGL(tstp_fwd_kernel, dim3(cv_blocks((long long)R * F * C)), dim3(256), 0, (hipStream_t)stream, x, R,
                     F, T, C, eps, stats);
  return ws_check_launch("ws_tstp_fwd");
}

extern "C" int ws_tstp_bwd(const float* x, const float* stats, const float* dstats, int R, int F, int T, int C,
                           float* dx, void* stream) {
  WS_REQUIRE(x && stats && dstats && dx && R > 0 && F > 0 && T > 0 && C > 0, "ws_tstp_bwd: bad args");
  hipLaunchKernelGGL(tstp_bwd_kernel, dim3(cv_blocks((long long)R * F * T * C)), dim3(256), 0, (hipStream_t)stream, x,
                     stats, dstats, R, F, T, C, dx);
  return ws_check_launch("ws_tstp_bwd");
}

// ---------------------------------------------------------------------------------------------
// In-model enrollment front-end (wesep/models/bsrnn.py:231-242,343-350):
// PreEmphasis (wesep/modules/common/speaker.py:10-23) + the framing half of
// torchaudio.transforms.MelSpectrogram(n_fft = win_length = 512, hop 128, hamming, center, reflect, power 2).
// The DFT and the mel projection are two exact-fp32 MFMA GEMMs (gemm.hip) on row views of these buffers.
// ---------------------------------------------------------------------------------------------
// out[r][j] = y[reflect(j - pad)],  y[i] = x[i] - coef * x[i-1]  (y[0] = x[0] - coef * x[1]: reflect pad of 1)
__global__ void preemph_pad_kernel(const float* __restrict__ x, int R, int T, int pad, int ldo, float coef,
                                   float* __restrict__ out) {
  const int Tp = T + 2 * pad;
  const long long total = (long long)R * Tp;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(i / Tp), j = (int)(i - (long long)r * Tp);
    int k = j - pad;
    if (k < 0) k = -k;
    if (k >= T) k = 2 * (T - 1) - k;
    const float* xr = x + (long long)r * T;
    out[(long long)r * ldo + j] = xr[k] - coef * (k > 0 ? xr[k - 1] : xr[1]);
  }
}

// p[m][f] = re^2 + im^2 from interleaved spectra [M][2*nf (ld lds_)]; columns nf..ldp-1 of p are zeroed
__global__ void power_spec_kernel(const float* __restrict__ spec, long long M, int nf, int lds_, int ldp,
                                  float* __restrict__ p) {
  const long long total = M * ldp;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long m = i / ldp;
    const int f = (int)(i - m * ldp);
    float v = 0.f;
    if (f < nf) {
      const float re = spec[m * lds_ + 2 * f], im = spec[m * lds_ + 2 * f + 1];
      v = re * re + im * im;
    }
    p[i] = v;
  }
}

// x = log(x + eps), in place
__global__ void log_eps_kernel(float* x, long long n, float eps) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    x[i] = logf(x[i] + eps);
}

extern "C" int ws_preemph_pad(const float* x, int R, int T, int pad, int ldo, float coef, float* out, void* stream) {
  WS_REQUIRE(x && out && R > 0 && T > pad && pad >= 0 && ldo >= T + 2 * pad, "ws_preemph_pad: bad args (T > pad)");
  hipLaunchKernelGGL(preemph_pad_kernel, dim3(cv_blocks((long long)R * (T + 2 * pad))), dim3(256), 0,
                     (hipStream_t)stream, x, R, T, pad, ldo, coef, out);
  return ws_check_launch("ws_preemph_pad");
}

extern "C" int ws_power_spec(const float* spec, long long M, int nf, int lds_, int ldp, float* p, void* stream) {
  WS_REQUIRE(spec && p && M > 0 && nf > 0 && lds_ >= 2 * nf && ldp >= nf, "ws_power_spec: bad args");
  hipLaunchKernelGGL(power_spec_kernel, dim3(cv_blocks(M * ldp)), dim3(256), 0, (hipStream_t)stream, spec, M, nf, lds_,
                     ldp, p);
  return ws_check_launch("ws_power_spec");
}

extern "C" int ws_log_eps(float* x, long long n, float eps, void* stream) {
  WS_REQUIRE(x && n > 0, "ws_log_eps: bad args");
  hipLaunchKernelGGL(log_eps_kernel, dim3(cv_blocks(n)), dim3(256), 0, (hipStream_t)stream, x, n, eps);
  return ws_check_launch("ws_log_eps");
}
