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
  int k, sh, sw, p; // kernel, stride along H / W, padding
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
    const int hi = ho * g.sh + tap / g.k - g.p, wi = wo * g.sw + tap % g.k - g.p;
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
    const int hi = ho * g.sh + tap / g.k - g.p, wi = wo * g.sw + tap % g.k - g.p;
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
      if (hn < 0 || hn % g.sh) continue;
      const int ho = hn / g.sh;
      if (ho >= g.Ho) continue;
      for (int kx = 0; kx < g.k; ++kx) {
        const int wn = wi + g.p - kx;
        if (wn < 0 || wn % g.sw) continue;
        const int wo = wn / g.sw;
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

static int cv_check(const char* who, int R, int H, int W, int C, int k, int sh, int sw, int p, int* Ho, int* Wo) {
  WS_REQUIRE(R > 0 && H > 0 && W > 0 && C > 0 && k >= 1 && sh >= 1 && sw >= 1 && p >= 0, "%s: bad geometry", who);
  *Ho = (H + 2 * p - k) / sh + 1;
  *Wo = (W + 2 * p - k) / sw + 1;
  WS_REQUIRE(*Ho > 0 && *Wo > 0, "%s: empty output", who);
  return WS_OK;
}

extern "C" int ws_im2col_hw(const float* x, int R, int H, int W, int C, int k, int sh, int sw, int p, long long ldp,
                            float* patches, void* stream) {
  int Ho, Wo;
  int rc = cv_check("ws_im2col", R, H, W, C, k, sh, sw, p, &Ho, &Wo);
  if (rc != WS_OK) return rc;
  WS_REQUIRE(x && patches, "ws_im2col: null pointer");
  const ConvGeom g{R, H, W, C, Ho, Wo, k, sh, sw, p};
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

extern "C" int ws_im2col(const float* x, int R, int H, int W, int C, int k, int s, int p, long long ldp,
                         float* patches, void* stream) {
  return ws_im2col_hw(x, R, H, W, C, k, s, s, p, ldp, patches, stream);
}

extern "C" int ws_col2im_hw(const float* dpatches, int R, int H, int W, int C, int k, int sh, int sw, int p, float* dx,
                            void* stream) {
  int Ho, Wo;
  int rc = cv_check("ws_col2im", R, H, W, C, k, sh, sw, p, &Ho, &Wo);
  if (rc != WS_OK) return rc;
  WS_REQUIRE(dpatches && dx && C % 4 == 0, "ws_col2im: null pointer / C %% 4");
  const ConvGeom g{R, H, W, C, Ho, Wo, k, sh, sw, p};
  hipLaunchKernelGGL(col2im_kernel, dim3(cv_blocks((long long)R * H * W * (C / 4))), dim3(256), 0, (hipStream_t)stream,
                     dpatches, g, dx);
  return ws_check_launch("ws_col2im");
}

extern "C" int ws_col2im(const float* dpatches, int R, int H, int W, int C, int k, int s, int p, float* dx,
                         void* stream) {
  return ws_col2im_hw(dpatches, R, H, W, C, k, s, s, p, dx, stream);
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
// ASTP (attentive statistics pooling of the wespeaker ECAPA-TDNN; call sites wesep/models/bsrnn.py:217,352-356, recipe
// examples/librimix/tse/v2/confs/bsrnn.yaml:66-71) on channels-last x, logits [R][T][C]:
//   alpha = softmax over T of logits;  mean = sum_t alpha x;  std = sqrt(max(sum_t alpha x^2 - mean^2, 1e-7))
//   out [R][2C] = mean || std;  aux [R][4][C] = (max logit, sum exp, mean, sum alpha x^2) for the backward.
// One thread per (r, c): lanes = consecutive channels (coalesced), three passes over T.
// ---------------------------------------------------------------------------------------------
__global__ void astp_fwd_kernel(const float* __restrict__ x, const float* __restrict__ lg, int R, int T, int C,
                                float floor_, float* __restrict__ out, float* __restrict__ aux) {
  const long long total = (long long)R * C;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C), r = (int)(i / C);
    const float* xb = x + (long long)r * T * C + c;
    const float* lb = lg + (long long)r * T * C + c;
    float m = -INFINITY;
    for (int t = 0; t < T; ++t) m = fmaxf(m, lb[(long long)t * C]);
    float z = 0.f, s1 = 0.f, s2 = 0.f;
    for (int t = 0; t < T; ++t) {
      const float e = expf(lb[(long long)t * C] - m), v = xb[(long long)t * C];
      z += e;
      s1 += e * v;
      s2 += e * v * v;
    }
    const float mean = s1 / z, ex2 = s2 / z;
    out[(long long)r * 2 * C + c] = mean;
    out[(long long)r * 2 * C + C + c] = sqrtf(fmaxf(ex2 - mean * mean, floor_));
    float* a = aux + (long long)r * 4 * C + c;
    a[0] = m;
    a[C] = z;
    a[2 * C] = mean;
    a[3 * C] = ex2;
  }
}

// dx_t = alpha_t (gm + 2 gv x_t),  dlogit_t = alpha_t (da_t - sum_s alpha_s da_s),  da_t = gm x_t + gv x_t^2,
// gv = dstd / (2 std) where the variance is above the floor (0 below), gm = dmean - 2 mean gv
__global__ void astp_bwd_kernel(const float* __restrict__ x, const float* __restrict__ lg, const float* __restrict__ out,
                                const float* __restrict__ aux, const float* __restrict__ dout, int R, int T, int C,
                                float floor_, float* __restrict__ dx, float* __restrict__ dlg) {
  const long long total = (long long)R * T * C;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int r = (int)(i / ((long long)T * C));
    const float* a = aux + (long long)r * 4 * C + c;
    const float m = a[0], z = a[C], mean = a[2 * C], ex2 = a[3 * C];
    const float sd = out[(long long)r * 2 * C + C + c];
    const float gmean = dout[(long long)r * 2 * C + c], gstd = dout[(long long)r * 2 * C + C + c];
    const float gv = (ex2 - mean * mean > floor_) ? gstd / (2.f * sd) : 0.f;
    const float gm = gmean - 2.f * mean * gv;
    const float v = x[i], al = expf(lg[i] - m) / z;
    const float da = gm * v + gv * v * v, dbar = gm * mean + gv * ex2;
    dx[i] = al * (gm + 2.f * gv * v);
    dlg[i] = al * (da - dbar);
  }
}

extern "C" int ws_astp_fwd(const float* x, const float* logits, int R, int T, int C, float floor_, float* out, float* aux,
                           void* stream) {
  WS_REQUIRE(x && logits && out && aux && R > 0 && T > 0 && C > 0, "ws_astp_fwd: bad args");
  hipLaunchKernelGGL(astp_fwd_kernel, dim3(cv_blocks((long long)R * C)), dim3(256), 0, (hipStream_t)stream, x, logits, R,
                     T, C, floor_, out, aux);
  return ws_check_launch("ws_astp_fwd");
}

extern "C" int ws_astp_bwd(const float* x, const float* logits, const float* out, const float* aux, const float* dout,
                           int R, int T, int C, float floor_, float* dx, float* dlogits, void* stream) {
  WS_REQUIRE(x && logits && out && aux && dout && dx && dlogits && R > 0 && T > 0 && C > 0, "ws_astp_bwd: bad args");
  hipLaunchKernelGGL(astp_bwd_kernel, dim3(cv_blocks((long long)R * T * C)), dim3(256), 0, (hipStream_t)stream, x, logits,
                     out, aux, dout, R, T, C, floor_, dx, dlogits);
  return ws_check_launch("ws_astp_bwd");
}

// y = act(x + rb[row / rows_per_r])  (act 1 tanh, 3 sigmoid) on [rows][C], and its backward from the saved output:
// dx = dy * (1 - y^2) / dy * y (1 - y); the small bottleneck activations of ECAPA's attention and SE blocks
__global__ void rowbias_act_fwd_kernel(const float* __restrict__ x, const float* __restrict__ rb, long long rows, int C,
                                       int rows_per_r, int act, float* __restrict__ y) {
  const long long total = rows * C;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long row = i / C;
    const int c = (int)(i - row * C);
    const float v = x[i] + (rb ? rb[(row / rows_per_r) * C + c] : 0.f);
    y[i] = act == 1 ? tanhf(v) : 1.f / (1.f + expf(-v));
  }
}
__global__ void act_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy, long long n, int act,
                               float* __restrict__ dx) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float t = y[i];
    dx[i] = dy[i] * (act == 1 ? 1.f - t * t : t * (1.f - t));
  }
}
extern "C" int ws_rowbias_act_fwd(const float* x, const float* rb, long long rows, int C, int rows_per_r, int act, float* y,
                                  void* stream) {
  WS_REQUIRE(x && y && rows > 0 && C > 0 && rows_per_r > 0 && (act == 1 || act == 3), "ws_rowbias_act_fwd: bad args");
  hipLaunchKernelGGL(rowbias_act_fwd_kernel, dim3(cv_blocks(rows * C)), dim3(256), 0, (hipStream_t)stream, x, rb, rows, C,
                     rows_per_r, act, y);
  return ws_check_launch("ws_rowbias_act_fwd");
}
extern "C" int ws_act_bwd(const float* y, const float* dy, long long n, int act, float* dx, void* stream) {
  WS_REQUIRE(y && dy && dx && n > 0 && (act == 1 || act == 3), "ws_act_bwd: bad args");
  hipLaunchKernelGGL(act_bwd_kernel, dim3(cv_blocks(n)), dim3(256), 0, (hipStream_t)stream, y, dy, n, act, dx);
  return ws_check_launch("ws_act_bwd");
}

// Segment pooling of CAM++'s context-aware mask (wespeaker CAMLayer.seg_pooling: F.avg_pool1d(seg_len, ceil_mode) expanded
// back over its frames; wesep/models/bsrnn.py:217 via the recipe's `CAMPPlus` alternative, bsrnn.yaml:66-74) on
// channels-last [R][T][C], nseg = ceil(T / seg_len), the last segment of each utterance may be shorter:
//   out[r][s][c] = sum_{t in segment s} a[r][t][c] (* b[r][t][c])                       (ws_seg_sums)
//   out[r][t][c] = (x ? x[r][t][c] : 1) * m[r][t / seg_len][c]                          (ws_seg_scale)
// The pair is the pooling and its adjoint, and the mask product with both of its gradients.
__global__ void seg_sums_kernel(const float* __restrict__ a, const float* __restrict__ b, int R, int T, int C, int seg_len,
                                int nseg, float* __restrict__ out) {
  const int c4 = C / 4;
  const long long total = (long long)R * nseg * c4;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int q = (int)(i % c4);
    const long long rs = i / c4;
    const int sgm = (int)(rs % nseg);
    const long long r = rs / nseg;
    const int t0 = sgm * seg_len, t1 = min(T, t0 + seg_len);
    const float* pa = a + (r * T + t0) * C + q * 4;
    const float* pb = b ? b + (r * T + t0) * C + q * 4 : nullptr;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    int t = t0;
    for (; t + 1 < t1; t += 2) {
      f32x4 v0 = *reinterpret_cast<const f32x4*>(pa), v1 = *reinterpret_cast<const f32x4*>(pa + C);
      if (pb) {
        v0 *= *reinterpret_cast<const f32x4*>(pb);
        v1 *= *reinterpret_cast<const f32x4*>(pb + C);
        pb += 2 * C;
      }
      acc0 += v0;
      acc1 += v1;
      pa += 2 * C;
    }
    if (t < t1) {
      f32x4 v0 = *reinterpret_cast<const f32x4*>(pa);
      if (pb) v0 *= *reinterpret_cast<const f32x4*>(pb);
      acc0 += v0;
    }
    *reinterpret_cast<f32x4*>(out + rs * C + q * 4) = acc0 + acc1;
  }
}
__global__ void seg_scale_kernel(const float* __restrict__ x, const float* __restrict__ m, int R, int T, int C, int seg_len,
                                 int nseg, float* __restrict__ out) {
  const int c4 = C / 4;
  const long long total = (long long)R * T * c4;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int q = (int)(i % c4);
    const long long row = i / c4;
    const int t = (int)(row % T);
    const long long r = row / T;
    f32x4 v = *reinterpret_cast<const f32x4*>(m + (r * nseg + t / seg_len) * C + q * 4);
    if (x) v *= *reinterpret_cast<const f32x4*>(x + row * C + q * 4);
    *reinterpret_cast<f32x4*>(out + row * C + q * 4) = v;
  }
}
extern "C" int ws_seg_sums(const float* a, const float* b, int R, int T, int C, int seg_len, float* out, void* stream) {
  WS_REQUIRE(a && out && R > 0 && T > 0 && C > 0 && C % 4 == 0 && seg_len > 0, "ws_seg_sums: bad args (C %% 4)");
  const int nseg = (T + seg_len - 1) / seg_len;
  hipLaunchKernelGGL(seg_sums_kernel, dim3(cv_blocks((long long)R * nseg * (C / 4))), dim3(256), 0, (hipStream_t)stream, a,
                     b, R, T, C, seg_len, nseg, out);
  return ws_check_launch("ws_seg_sums");
}
extern "C" int ws_seg_scale(const float* x, const float* m, int R, int T, int C, int seg_len, float* out, void* stream) {
  WS_REQUIRE(m && out && R > 0 && T > 0 && C > 0 && C % 4 == 0 && seg_len > 0, "ws_seg_scale: bad args (C %% 4)");
  const int nseg = (T + seg_len - 1) / seg_len;
  hipLaunchKernelGGL(seg_scale_kernel, dim3(cv_blocks((long long)R * T * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, m,
                     R, T, C, seg_len, nseg, out);
  return ws_check_launch("ws_seg_scale");
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


// ---------------------------------------------------------------------------------------------
// DPCCN pieces (wesep/modules/dpccn/convs.py, wesep/models/dpccn.py), channels-last
// [B][H][W][C] (H = frame, W = frequency bin):  ELU, InstanceNorm (no affine) over the positions of one batch row,
// AvgPool2d(sz), bilinear upsampling (align_corners = False), and the speaker fusion's per-(row, bin) scale / shift.
// 2-D convolutions / transposed convolutions are ws_im2col_hw / ws_col2im_hw + GEMM.
// ---------------------------------------------------------------------------------------------
__global__ void elu_fwd_kernel(const float* __restrict__ x, long long n4, float* __restrict__ y) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + i * 4);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = v[j] > 0.f ? v[j] : expm1f(v[j]);
    *reinterpret_cast<f32x4*>(y + i * 4) = o;
  }
}

// dx = dy * (x > 0 ? 1 : exp(x))   (dx may alias dy)
__global__ void elu_bwd_kernel(const float* __restrict__ x, const float* dy, long long n4, float* dx) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + i * 4), g = *reinterpret_cast<const f32x4*>(dy + i * 4);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = v[j] > 0.f ? g[j] : g[j] * expf(v[j]);
    *reinterpret_cast<f32x4*>(dx + i * 4) = o;
  }
}

extern "C" int ws_elu_fwd(const float* x, long long n, float* y, void* stream) {
  WS_REQUIRE(x && y && n > 0 && n % 4 == 0, "ws_elu_fwd: bad args");
  hipLaunchKernelGGL(elu_fwd_kernel, dim3(cv_blocks(n / 4)), dim3(256), 0, (hipStream_t)stream, x, n / 4, y);
  return ws_check_launch("ws_elu_fwd");
}

extern "C" int ws_elu_bwd(const float* x, const float* dy, long long n, float* dx, void* stream) {
  WS_REQUIRE(x && dy && dx && n > 0 && n % 4 == 0, "ws_elu_bwd: bad args");
  hipLaunchKernelGGL(elu_bwd_kernel, dim3(cv_blocks(n / 4)), dim3(256), 0, (hipStream_t)stream, x, dy, n / 4, dx);
  return ws_check_launch("ws_elu_bwd");
}

// sums [G][2][C] = (sum x, sum x^2) over the P positions of group g (ws_chan_sums with g = x = the input)
// -> stats [G][2][C] = (mean, 1/sqrt(max(E[x^2] - mean^2, 0) + eps))
__global__ void inorm_finalize_kernel(const float* __restrict__ sums, long long n, int C, float inv_p, float eps,
                                      float* __restrict__ stats) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const long long g = i / C;
    const int c = (int)(i - g * C);
    const float mean = sums[g * 2 * C + c] * inv_p;
    const float var = fmaxf(sums[g * 2 * C + C + c] * inv_p - mean * mean, 0.f);
    stats[g * 2 * C + c] = mean;
    stats[g * 2 * C + C + c] = 1.f / sqrtf(var + eps);
  }
}

// y = (x - mean[g][c]) * rstd[g][c],  g = row / P
__global__ void inorm_apply_kernel(const float* __restrict__ x, const float* __restrict__ stats, long long rows, int P,
                                   int C, float* __restrict__ y) {
  const int c4n = C >> 2;
  const long long total = rows * c4n;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long row = i / c4n;
    const int c = (int)(i - row * c4n) * 4;
    const float* st = stats + (row / P) * 2 * C;
    *reinterpret_cast<f32x4*>(y + i * 4) = (*reinterpret_cast<const f32x4*>(x + i * 4) -
                                             *reinterpret_cast<const f32x4*>(st + c)) *
                                            *reinterpret_cast<const f32x4*>(st + C + c);
  }
}

// dx = rstd * (dy - S0/P - y * S1/P),  sums [G][2][C] = (sum dy, sum dy * y) (ws_chan_sums(g = dy, x = y))
__global__ void inorm_bwd_apply_kernel(const float* __restrict__ y, const float* dy, const float* __restrict__ stats,
                                       const float* __restrict__ sums, long long rows, int P, int C, float* dx) {
  const int c4n = C >> 2;
  const long long total = rows * c4n;
  const float inv = 1.f / (float)P;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long row = i / c4n;
    const int c = (int)(i - row * c4n) * 4;
    const long long g = row / P;
    const f32x4 rstd = *reinterpret_cast<const f32x4*>(stats + g * 2 * C + C + c);
    const f32x4 s0 = *reinterpret_cast<const f32x4*>(sums + g * 2 * C + c) * inv;
    const f32x4 s1 = *reinterpret_cast<const f32x4*>(sums + g * 2 * C + C + c) * inv;
    *reinterpret_cast<f32x4*>(dx + i * 4) =
        rstd * (*reinterpret_cast<const f32x4*>(dy + i * 4) - s0 - *reinterpret_cast<const f32x4*>(y + i * 4) * s1);
  }
}

extern "C" int ws_inorm_finalize(const float* sums, int G, int C, long long P, float eps, float* stats, void* stream) {
  WS_REQUIRE(sums && stats && G > 0 && C > 0 && P > 0, "ws_inorm_finalize: bad args");
  hipLaunchKernelGGL(inorm_finalize_kernel, dim3(cv_blocks((long long)G * C)), dim3(256), 0, (hipStream_t)stream, sums,
                     (long long)G * C, C, 1.f / (float)P, eps, stats);
  return ws_check_launch("ws_inorm_finalize");
}

extern "C" int ws_inorm_apply(const float* x, const float* stats, long long rows, int P, int C, float* y, void* stream) {
  WS_REQUIRE(x && stats && y && rows > 0 && P > 0 && C > 0 && C % 4 == 0 && rows % P == 0, "ws_inorm_apply: bad args");
  hipLaunchKernelGGL(inorm_apply_kernel, dim3(cv_blocks(rows * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, stats,
                     rows, P, C, y);
  return ws_check_launch("ws_inorm_apply");
}

extern "C" int ws_inorm_bwd_apply(const float* y, const float* dy, const float* stats, const float* sums, long long rows,
                                  int P, int C, float* dx, void* stream) {
  WS_REQUIRE(y && dy && stats && sums && dx && rows > 0 && P > 0 && C > 0 && C % 4 == 0 && rows % P == 0,
             "ws_inorm_bwd_apply: bad args");
  hipLaunchKernelGGL(inorm_bwd_apply_kernel, dim3(cv_blocks(rows * (C / 4))), dim3(256), 0, (hipStream_t)stream, y, dy,
                     stats, sums, rows, P, C, dx);
  return ws_check_launch("ws_inorm_bwd_apply");
}

// ---------------------------------------------------------------------------------------------
// InstanceNorm fused with its neighbouring ELU (round 3).  DPCCN wraps every convolution as conv - ELU - InstanceNorm
// (convs.py:28-77) and every TCN block as InstanceNorm - ELU - conv (convs.py:115-152); as separate kernels that is five
// passes over the activation forward (ELU read + write, statistics read, normalise read + write) and eight backward.
// With the pointwise function applied on load / on store the forward is three passes (statistics; normalise) and the
// backward five (sums; apply), and only the PRE-activation is kept for the backward.
//   flags bit 0: ELU before the normalisation   y = IN(ELU(x))
//   flags bit 1: ELU after it                   y = ELU(IN(x))
// Statistics [G][2][C] = (mean, rstd) per (row group of P positions, channel), as ws_inorm_finalize writes them.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ f32x4 elu4(const f32x4& v) {
  f32x4 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = v[j] > 0.f ? v[j] : expm1f(v[j]);
  return o;
}
__device__ __forceinline__ f32x4 elud4(const f32x4& v) {   // d ELU(v) / dv
  f32x4 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = v[j] > 0.f ? 1.f : expf(v[j]);
  return o;
}

// slab[split][g][2][C]: forward (dy == NULL): (sum u, sum u^2), u = pre(x);
// backward: (sum d, sum d * n), n = (u - mean) * rstd, d = dy * (flags & 2 ? ELU'(n) : 1)
__global__ __launch_bounds__(256) void in_act_sums_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                          const float* __restrict__ stats, int P, int G, int nsplit, int C,
                                                          int flags, long long ldd, float* __restrict__ slab) {
  __shared__ f32x4 red[2][256];
  const int c4n = C >> 2;
  const int cz = blockIdx.z * 256;
  const int nq = min(256, c4n - cz);
  const int nrl = 256 / nq;
  const int tid = threadIdx.x, rl = tid / nq, q = tid - rl * nq;
  const bool on = rl < nrl;
  const int c = (cz + q) * 4;
  const int split = blockIdx.x, grp = blockIdx.y;
  const int per = (P + nsplit - 1) / nsplit;
  const int lo = split * per, hi = min(P, lo + per);
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
  if (on) {
    f32x4 mean = s0, rstd = s0;
    if (dy) {
      mean = *reinterpret_cast<const f32x4*>(stats + (long long)grp * 2 * C + c);
      rstd = *reinterpret_cast<const f32x4*>(stats + (long long)grp * 2 * C + C + c);
    }
    for (int j = lo + rl; j < hi; j += nrl) {
      const long long row = (long long)grp * P + j;
      f32x4 u = *reinterpret_cast<const f32x4*>(x + row * C + c);
      if (flags & 1) u = elu4(u);
      if (!dy) {
        s0 += u;
        s1 += u * u;
      } else {
        const f32x4 n = (u - mean) * rstd;
        f32x4 d = *reinterpret_cast<const f32x4*>(dy + row * ldd + c);
        if (flags & 2) d *= elud4(n);
        s0 += d;
        s1 += d * n;
      }
    }
  }
  red[0][tid] = s0;
  red[1][tid] = s1;
  __syncthreads();
  if (tid < nq) {
    f32x4 t0 = red[0][tid], t1 = red[1][tid];
    for (int r = 1; r < nrl; ++r) {
      t0 += red[0][r * nq + tid];
      t1 += red[1][r * nq + tid];
    }
    float* o = slab + ((long long)split * G + grp) * 2 * C;
    *reinterpret_cast<f32x4*>(o + (cz + tid) * 4) = t0;
    *reinterpret_cast<f32x4*>(o + C + (cz + tid) * 4) = t1;
  }
}

__global__ void in_act_apply_kernel(const float* __restrict__ x, const float* __restrict__ stats, long long rows, int P,
                                    int C, int flags, long long ldy, float* __restrict__ y) {
  const int c4n = C >> 2;
  const long long total = rows * c4n;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long row = i / c4n;
    const int c = (int)(i - row * c4n) * 4;
    const float* st = stats + (row / P) * 2 * C;
    f32x4 u = *reinterpret_cast<const f32x4*>(x + i * 4);
    if (flags & 1) u = elu4(u);
    f32x4 n = (u - *reinterpret_cast<const f32x4*>(st + c)) * *reinterpret_cast<const f32x4*>(st + C + c);
    if (flags & 2) n = elu4(n);
    *reinterpret_cast<f32x4*>(y + row * ldy + c) = n;
  }
}

// dx = pre'(x) * rstd * (d - S0/P - n * S1/P)      (dx may alias dy)
__global__ void in_act_bwd_apply_kernel(const float* __restrict__ x, const float* dy, const float* __restrict__ stats,
                                        const float* __restrict__ sums, long long rows, int P, int C, int flags,
                                        long long ldd, long long lddx, float* dx) {
  const int c4n = C >> 2;
  const long long total = rows * c4n;
  const float inv = 1.f / (float)P;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long row = i / c4n;
    const int c = (int)(i - row * c4n) * 4;
    const long long g = row / P;
    const f32x4 mean = *reinterpret_cast<const f32x4*>(stats + g * 2 * C + c);
    const f32x4 rstd = *reinterpret_cast<const f32x4*>(stats + g * 2 * C + C + c);
    const f32x4 s0 = *reinterpret_cast<const f32x4*>(sums + g * 2 * C + c) * inv;
    const f32x4 s1 = *reinterpret_cast<const f32x4*>(sums + g * 2 * C + C + c) * inv;
    const f32x4 xv = *reinterpret_cast<const f32x4*>(x + i * 4);
    const f32x4 u = (flags & 1) ? elu4(xv) : xv;
    const f32x4 n = (u - mean) * rstd;
    f32x4 d = *reinterpret_cast<const f32x4*>(dy + row * ldd + c);
    if (flags & 2) d *= elud4(n);
    f32x4 r = rstd * (d - s0 - n * s1);
    if (flags & 1) r *= elud4(xv);
    *reinterpret_cast<f32x4*>(dx + row * lddx + c) = r;
  }
}

extern "C" int ws_in_act_sums(const float* x, const float* dy, long long ldd, const float* stats, int P, int G, int nsplit, int C,
                              int flags, float* slab, void* stream) {
  WS_REQUIRE(x && slab && P > 0 && G > 0 && nsplit > 0 && C > 0 && C % 4 == 0 && (flags & ~3) == 0 && (!dy || stats),
             "ws_in_act_sums: bad args (the backward sums need the statistics)");
  WS_REQUIRE(ldd == 0 || (ldd >= C && ldd % 4 == 0), "ws_in_act_sums: dy row stride %lld (0 = C, else >= C and %% 4)", ldd);
  hipLaunchKernelGGL(in_act_sums_kernel, dim3(nsplit, G, (C / 4 + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, dy, stats,
                     P, G, nsplit, C, flags, ldd ? ldd : (long long)C, slab);
  return ws_check_launch("ws_in_act_sums");
}

extern "C" int ws_in_act_apply(const float* x, const float* stats, long long rows, int P, int C, int flags, float* y,
                               long long ldy, void* stream) {
  WS_REQUIRE(x && stats && y && rows > 0 && P > 0 && C > 0 && C % 4 == 0 && rows % P == 0 && (flags & ~3) == 0,
             "ws_in_act_apply: bad args");
  WS_REQUIRE(ldy == 0 || (ldy >= C && ldy % 4 == 0), "ws_in_act_apply: y row stride %lld (0 = C, else >= C and %% 4)", ldy);
  hipLaunchKernelGGL(in_act_apply_kernel, dim3(cv_blocks(rows * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, stats, rows, P,
                     C, flags, ldy ? ldy : (long long)C, y);
  return ws_check_launch("ws_in_act_apply");
}

extern "C" int ws_in_act_bwd_apply(const float* x, const float* dy, long long ldd, const float* stats, const float* sums,
                                   long long rows, int P, int C, int flags, float* dx, long long lddx, void* stream) {
  WS_REQUIRE(x && dy && stats && sums && dx && rows > 0 && P > 0 && C > 0 && C % 4 == 0 && rows % P == 0 && (flags & ~3) == 0,
             "ws_in_act_bwd_apply: bad args");
  WS_REQUIRE((ldd == 0 || (ldd >= C && ldd % 4 == 0)) && (lddx == 0 || (lddx >= C && lddx % 4 == 0)),
             "ws_in_act_bwd_apply: dy / dx row strides %lld / %lld (0 = C, else >= C and %% 4)", ldd, lddx);
  hipLaunchKernelGGL(in_act_bwd_apply_kernel, dim3(cv_blocks(rows * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, dy, stats,
                     sums, rows, P, C, flags, ldd ? ldd : (long long)C, lddx ? lddx : (long long)C, dx);
  return ws_check_launch("ws_in_act_bwd_apply");
}

// AvgPool2d(sz) (stride sz, floor): [B][H][W][C] -> [B][H/sz][W/sz][C]; backward spreads dy / sz^2 (0 on the dropped tail)
__global__ void avgpool_fwd_kernel(const float* __restrict__ x, int B, int H, int W, int C, int sz,
                                   float* __restrict__ y) {
  const int Ho = H / sz, Wo = W / sz, c4n = C >> 2;
  const long long total = (long long)B * Ho * Wo * c4n;
  const float inv = 1.f / (float)(sz * sz);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % c4n) * 4;
    long long q = i / c4n;
    const int wo = (int)(q % Wo);
    q /= Wo;
    const int ho = (int)(q % Ho), b = (int)(q / Ho);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int dy_ = 0; dy_ < sz; ++dy_)
      for (int dx_ = 0; dx_ < sz; ++dx_)
        acc += *reinterpret_cast<const f32x4*>(x + (((long long)b * H + ho * sz + dy_) * W + wo * sz + dx_) * C + c);
    *reinterpret_cast<f32x4*>(y + i * 4) = acc * inv;
  }
}

__global__ void avgpool_bwd_kernel(const float* __restrict__ dy, int B, int H, int W, int C, int sz,
                                   float* __restrict__ dx) {
  const int Ho = H / sz, Wo = W / sz, c4n = C >> 2;
  const long long total = (long long)B * H * W * c4n;
  const float inv = 1.f / (float)(sz * sz);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % c4n) * 4;
    long long q = i / c4n;
    const int w = (int)(q % W);
    q /= W;
    const int h = (int)(q % H), b = (int)(q / H);
    const int ho = h / sz, wo = w / sz;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (ho < Ho && wo < Wo) v = *reinterpret_cast<const f32x4*>(dy + (((long long)b * Ho + ho) * Wo + wo) * C + c) * inv;
    *reinterpret_cast<f32x4*>(dx + i * 4) = v;
  }
}

extern "C" int ws_avgpool_fwd(const float* x, int B, int H, int W, int C, int sz, float* y, void* stream) {
  WS_REQUIRE(x && y && B > 0 && sz > 0 && H >= sz && W >= sz && C > 0 && C % 4 == 0, "ws_avgpool_fwd: bad args");
  hipLaunchKernelGGL(avgpool_fwd_kernel, dim3(cv_blocks((long long)B * (H / sz) * (W / sz) * (C / 4))), dim3(256), 0,
                     (hipStream_t)stream, x, B, H, W, C, sz, y);
  return ws_check_launch("ws_avgpool_fwd");
}

extern "C" int ws_avgpool_bwd(const float* dy, int B, int H, int W, int C, int sz, float* dx, void* stream) {
  WS_REQUIRE(dy && dx && B > 0 && sz > 0 && H >= sz && W >= sz && C > 0 && C % 4 == 0, "ws_avgpool_bwd: bad args");
  hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(cv_blocks((long long)B * H * W * (C / 4))), dim3(256), 0,
                     (hipStream_t)stream, dy, B, H, W, C, sz, dx);
  return ws_check_launch("ws_avgpool_bwd");
}

// bilinear, align_corners = False (nn.Upsample(size = (H, W), mode = "bilinear"), dpccn.py:263):
//   src = max((dst + 0.5) * (h / H) - 0.5, 0); i0 = floor(src), i1 = min(i0 + 1, h - 1), l = src - i0
__device__ __forceinline__ void bl_src(int dst, float scale, int n, int& i0, int& i1, float& l) {
  const float s = fmaxf(((float)dst + 0.5f) * scale - 0.5f, 0.f);
  i0 = min((int)s, n - 1);
  i1 = min(i0 + 1, n - 1);
  l = s - (float)i0;
}

__global__ void bilinear_fwd_kernel(const float* __restrict__ x, int B, int h, int w, int H, int W, int C,
                                    float* __restrict__ y) {
  const int c4n = C >> 2;
  const long long total = (long long)B * H * W * c4n;
  const float sh = (float)h / (float)H, sw = (float)w / (float)W;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % c4n) * 4;
    long long q = i / c4n;
    const int X = (int)(q % W);
    q /= W;
    const int Y = (int)(q % H), b = (int)(q / H);
    int y0, y1, x0, x1;
    float ly, lx;
    bl_src(Y, sh, h, y0, y1, ly);
    bl_src(X, sw, w, x0, x1, lx);
    const float* base = x + (long long)b * h * w * C + c;
    const f32x4 v00 = *reinterpret_cast<const f32x4*>(base + ((long long)y0 * w + x0) * C);
    const f32x4 v01 = *reinterpret_cast<const f32x4*>(base + ((long long)y0 * w + x1) * C);
    const f32x4 v10 = *reinterpret_cast<const f32x4*>(base + ((long long)y1 * w + x0) * C);
    const f32x4 v11 = *reinterpret_cast<const f32x4*>(base + ((long long)y1 * w + x1) * C);
    *reinterpret_cast<f32x4*>(y + i * 4) =
        (v00 * (1.f - lx) + v01 * lx) * (1.f - ly) + (v10 * (1.f - lx) + v11 * lx) * ly;
  }
}

// adjoint as two separable gathers (deterministic): pass 1 contracts the destination columns of every destination row,
//   tmp[b][Y][xs][c] = sum_X wx(X, xs) * dy[b][Y][X][c],
// pass 2 the destination rows, dx[b][ys][xs][c] = sum_Y wy(Y, ys) * tmp[b][Y][xs][c].  Round 2 had one thread per SOURCE
// pixel walking its whole (2 * scale + 3)^2 support: for the 32x pooling branch of DPCCN that is 3 584 threads x 4 489
// pixels (6 ms per call); here pass 1 has B * H * w * C / 4 threads and either pass walks 2 * scale + 3 taps.
__device__ __forceinline__ void bl_window(int s, float scale, int N, int& lo, int& hi) {
  // destination indices whose two source taps can include s (widened by one; the exact weight test decides)
  lo = max(0, (int)floorf(((float)s - 0.5f) / scale - 0.5f) - 1);
  hi = min(N - 1, (int)ceilf(((float)s + 1.5f) / scale - 0.5f) + 1);
}

__global__ void bilinear_bwd_rows_kernel(const float* __restrict__ dy, int B, int w, int H, int W, int C,
                                         float* __restrict__ tmp) {
  const int c4n = C >> 2;
  const long long total = (long long)B * H * w * c4n;
  const float sw = (float)w / (float)W;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % c4n) * 4;
    long long q = i / c4n;
    const int xs = (int)(q % w);
    const long long bY = q / w;                       // b * H + Y
    int Xlo, Xhi;
    bl_window(xs, sw, W, Xlo, Xhi);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int X = Xlo; X <= Xhi; ++X) {
      int x0, x1;
      float lx;
      bl_src(X, sw, w, x0, x1, lx);
      const float wx = (x0 == xs ? 1.f - lx : 0.f) + (x1 == xs ? lx : 0.f);
      if (wx == 0.f) continue;
      acc += *reinterpret_cast<const f32x4*>(dy + (bY * W + X) * C + c) * wx;
    }
    *reinterpret_cast<f32x4*>(tmp + i * 4) = acc;
  }
}

__global__ void bilinear_bwd_cols_kernel(const float* __restrict__ tmp, int B, int h, int w, int H, int C,
                                         float* __restrict__ dx) {
  const int c4n = C >> 2;
  const long long total = (long long)B * h * w * c4n;
  const float sh = (float)h / (float)H;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % c4n) * 4;
    long long q = i / c4n;
    const int xs = (int)(q % w);
    q /= w;
    const int ys = (int)(q % h), b = (int)(q / h);
    int Ylo, Yhi;
    bl_window(ys, sh, H, Ylo, Yhi);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int Y = Ylo; Y <= Yhi; ++Y) {
      int y0, y1;
      float ly;
      bl_src(Y, sh, h, y0, y1, ly);
      const float wy = (y0 == ys ? 1.f - ly : 0.f) + (y1 == ys ? ly : 0.f);
      if (wy == 0.f) continue;
      acc += *reinterpret_cast<const f32x4*>(tmp + (((long long)b * H + Y) * w + xs) * C + c) * wy;
    }
    *reinterpret_cast<f32x4*>(dx + i * 4) = acc;
  }
}

extern "C" int ws_bilinear_fwd(const float* x, int B, int h, int w, int H, int W, int C, float* y, void* stream) {
  WS_REQUIRE(x && y && B > 0 && h > 0 && w > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "ws_bilinear_fwd: bad args");
  hipLaunchKernelGGL(bilinear_fwd_kernel, dim3(cv_blocks((long long)B * H * W * (C / 4))), dim3(256), 0,
                     (hipStream_t)stream, x, B, h, w, H, W, C, y);
  return ws_check_launch("ws_bilinear_fwd");
}

extern "C" int ws_bilinear_bwd(const float* dy, int B, int h, int w, int H, int W, int C, float* tmp, float* dx,
                               void* stream) {
  WS_REQUIRE(dy && dx && tmp && B > 0 && h > 0 && w > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0,
             "ws_bilinear_bwd: bad args (tmp: B * H * w * C floats of scratch)");
  hipLaunchKernelGGL(bilinear_bwd_rows_kernel, dim3(cv_blocks((long long)B * H * w * (C / 4))), dim3(256), 0,
                     (hipStream_t)stream, dy, B, w, H, W, C, tmp);
  hipLaunchKernelGGL(bilinear_bwd_cols_kernel, dim3(cv_blocks((long long)B * h * w * (C / 4))), dim3(256), 0,
                     (hipStream_t)stream, tmp, B, h, w, H, C, dx);
  return ws_check_launch("ws_bilinear_bwd");
}

// speaker fusion on [B][T][F][C] (speaker.py:102-121 on the [B, C, F, T] view): y = x * s[b][f] (mode 0) or x + s[b][f]
// (mode 1); backward: dx = dy * s (or dy); ds[b][f] = sum over (t, c) of dy * x (or dy): one workgroup per (b, f)
__global__ void scale_bf_fwd_kernel(const float* __restrict__ x, const float* __restrict__ s, int B, int T, int Fq,
                                    int C, int mode, float* __restrict__ y) {
  const int c4n = C >> 2;
  const long long total = (long long)B * T * Fq * c4n;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    long long q = i / c4n;
    const int f = (int)(q % Fq);
    const int b = (int)(q / ((long long)Fq * T));
    const float sv = s[(long long)b * Fq + f];
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + i * 4);
    *reinterpret_cast<f32x4*>(y + i * 4) = mode == 0 ? v * sv : v + sv;
  }
}

__global__ __launch_bounds__(256) void scale_bf_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                           const float* __restrict__ s, int B, int T, int Fq, int C,
                                                           int mode, float* __restrict__ dx, float* __restrict__ ds) {
  __shared__ float red[16];
  const int b = blockIdx.x / Fq, f = blockIdx.x % Fq;
  const float sv = s[(long long)b * Fq + f];
  float acc = 0.f;
  for (int i = threadIdx.x; i < T * C; i += 256) {
    const int t = i / C, c = i - t * C;
    const long long o = (((long long)b * T + t) * Fq + f) * C + c;
    const float g = dy[o];
    acc += mode == 0 ? g * x[o] : g;
    dx[o] = mode == 0 ? g * sv : g;
  }
  acc = ws_block_sum(acc, red);
  if (threadIdx.x == 0) ds[blockIdx.x] = acc;
}

// C % 4 == 0: the same sums with 16-byte accesses -- C / 4 threads per row, 256 / (C / 4) rows per pass, two passes in flight (the
// scalar kernel above moved 0.6 GB in 0.7 ms on TF-GridNet's map, seven launches per step)
__global__ __launch_bounds__(256) void scale_bf_bwd4_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            const float* __restrict__ s, int B, int T, int Fq, int C,
                                                            int mode, float* __restrict__ dx, float* __restrict__ ds) {
  __shared__ float red[16];
  const int b = blockIdx.x / Fq, f = blockIdx.x % Fq;
  const float sv = s[(long long)b * Fq + f];
  const int c4n = C >> 2, nrl = 256 / c4n;          // row lanes (C <= 1024)
  const int c4 = threadIdx.x % c4n, rl = threadIdx.x / c4n;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (rl < nrl) {
    const long long base = ((long long)b * T * Fq + f) * C + 4 * c4, rs = (long long)Fq * C;
    for (int t = rl; t < T; t += 2 * nrl) {
      const bool two = t + nrl < T;
      const long long o0 = base + t * rs, o1 = base + (two ? t + nrl : t) * rs;
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(dy + o0), g1 = *reinterpret_cast<const f32x4*>(dy + o1);
      if (mode == 0) {
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(x + o0), x1 = *reinterpret_cast<const f32x4*>(x + o1);
        acc += g0 * x0;
        if (two) acc += g1 * x1;
        *reinterpret_cast<f32x4*>(dx + o0) = g0 * sv;
        if (two) *reinterpret_cast<f32x4*>(dx + o1) = g1 * sv;
      } else {
        acc += g0;
        if (two) acc += g1;
        *reinterpret_cast<f32x4*>(dx + o0) = g0;
        if (two) *reinterpret_cast<f32x4*>(dx + o1) = g1;
      }
    }
  }
  const float tot = ws_block_sum(acc[0] + acc[1] + acc[2] + acc[3], red);
  if (threadIdx.x == 0) ds[blockIdx.x] = tot;
}

extern "C" int ws_scale_bf_fwd(const float* x, const float* s, int B, int T, int Fq, int C, int mode, float* y,
                               void* stream) {
  WS_REQUIRE(x && s && y && B > 0 && T > 0 && Fq > 0 && C > 0 && C % 4 == 0 && (mode == 0 || mode == 1),
             "ws_scale_bf_fwd: bad args");
  hipLaunchKernelGGL(scale_bf_fwd_kernel, dim3(cv_blocks((long long)B * T * Fq * (C / 4))), dim3(256), 0,
                     (hipStream_t)stream, x, s, B, T, Fq, C, mode, y);
  return ws_check_launch("ws_scale_bf_fwd");
}

extern "C" int ws_scale_bf_bwd(const float* x, const float* dy, const float* s, int B, int T, int Fq, int C, int mode,
                               float* dx, float* ds, void* stream) {
  WS_REQUIRE(x && dy && s && dx && ds && B > 0 && T > 0 && Fq > 0 && C > 0 && (mode == 0 || mode == 1),
             "ws_scale_bf_bwd: bad args");
  if (C % 4 == 0 && C <= 1024 && 256 % (C / 4) == 0)
    hipLaunchKernelGGL(scale_bf_bwd4_kernel, dim3(B * Fq), dim3(256), 0, (hipStream_t)stream, x, dy, s, B, T, Fq, C, mode, dx, ds);
  else
    hipLaunchKernelGGL(scale_bf_bwd_kernel, dim3(B * Fq), dim3(256), 0, (hipStream_t)stream, x, dy, s, B, T, Fq, C, mode,
                       dx, ds);
  return ws_check_launch("ws_scale_bf_bwd");
}

// SpeakerFuseLayer 'concat' on [B][T][F][C] (speaker.py:95-101 on the [B, C, F, T] view): a Linear over the FREQUENCY axis of
// cat[x, e]:  y[b][t][f'][c] = sum_f W[f'][f] x[b][t][f][c] + rb[b][f'],  rb = We e + bias (a plain GEMM of the caller).
// One workgroup per (b, t) keeps the [F][C] tile in LDS; a thread owns (f', four channels).  Exact fp32 FMAs: this is the
// native runtime's (inference) form of the fusion -- the training path composes it from GEMMs on a transposed view
// (models/dpccn.py fuse_bins).
__global__ __launch_bounds__(256) void freq_linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                              long long ldw, const float* __restrict__ rb, int T, int Fq,
                                                              int C, float* __restrict__ y) {
  extern __shared__ __attribute__((aligned(16))) float fl_xt[];
  const long long tile = blockIdx.x;
  const int b = (int)(tile / T);
  const float* xs = x + tile * Fq * C;
  for (int i = threadIdx.x * 4; i < Fq * C; i += 1024) *reinterpret_cast<f32x4*>(fl_xt + i) = *reinterpret_cast<const f32x4*>(xs + i);
  __syncthreads();
  const int c4n = C >> 2;
  for (int o = threadIdx.x; o < Fq * c4n; o += 256) {
    const int fo = o / c4n, c = (o - fo * c4n) * 4;
    const float* wr = W + fo * ldw;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int f = 0; f < Fq; ++f) acc += wr[f] * *reinterpret_cast<const f32x4*>(fl_xt + f * C + c);
    *reinterpret_cast<f32x4*>(y + (tile * Fq + fo) * C + c) = acc + rb[(long long)b * Fq + fo];
  }
}

extern "C" int ws_freq_linear_fwd(const float* x, const float* W, long long ldw, const float* rb, int B, int T, int Fq, int C,
                                  float* y, void* stream) {
  WS_REQUIRE(x && W && rb && y && B > 0 && T > 0 && Fq > 0 && C > 0 && C % 4 == 0 && ldw >= Fq,
             "ws_freq_linear_fwd: bad args");
  WS_REQUIRE((long long)Fq * C <= 16384, "ws_freq_linear_fwd: F * C = %lld floats exceed the 64 KB tile", (long long)Fq * C);
  hipLaunchKernelGGL(freq_linear_fwd_kernel, dim3(B * T), dim3(256), (size_t)Fq * C * 4, (hipStream_t)stream, x, W, ldw, rb, T,
                     Fq, C, y);
  return ws_check_launch("ws_freq_linear_fwd");
}

// ---------------------------------------------------------------------------------------------
// Row softmax for the full-band self-attention of TF-GridNet (gridnet_block.py:212-213):
//   y[r][:] = softmax(scale * x[r][:]);   dx = scale * y * (dy - sum(dy * y))      one workgroup per row
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_rows_fwd_kernel(const float* __restrict__ x, int n, float scale,
                                                               float* __restrict__ y) {
  __shared__ float red[16];
  const float* xr = x + (long long)blockIdx.x * n;
  float* yr = y + (long long)blockIdx.x * n;
  float mx = -3.4e38f;
  for (int j = threadIdx.x; j < n; j += 256) mx = fmaxf(mx, scale * xr[j]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float se = 0.f;
  for (int j = threadIdx.x; j < n; j += 256) {
    const float e = expf(scale * xr[j] - mx);
    yr[j] = e;
    se += e;
  }
  se = ws_block_sum(se, red);
  const float inv = 1.f / se;
  for (int j = threadIdx.x; j < n; j += 256) yr[j] *= inv;
}

__global__ __launch_bounds__(256) void softmax_rows_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                                               int n, float scale, float* __restrict__ dx) {
  __shared__ float red[16];
  const long long o = (long long)blockIdx.x * n;
  float s = 0.f;
  for (int j = threadIdx.x; j < n; j += 256) s += dy[o + j] * y[o + j];
  s = ws_block_sum(s, red);
  for (int j = threadIdx.x; j < n; j += 256) dx[o + j] = scale * y[o + j] * (dy[o + j] - s);
}

// n <= 1024, n % 4 == 0 (TF-GridNet's padded key axis): the row lives in one float4 per thread -- one read and one write of the
// row instead of three passes of 4-byte accesses (0.29 ms for 2 x 72 MB per launch before)
__global__ __launch_bounds__(256) void softmax_rows_fwd4_kernel(const float* __restrict__ x, int n, float scale,
                                                                float* __restrict__ y) {
  __shared__ float red[16];
  const long long o = (long long)blockIdx.x * n + 4 * threadIdx.x;
  const bool on = 4 * (int)threadIdx.x < n;
  f32x4 v = {-3.4e38f, -3.4e38f, -3.4e38f, -3.4e38f};
  if (on) v = *reinterpret_cast<const f32x4*>(x + o) * scale;
  float mx = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
#pragma unroll
  for (int k = 32; k > 0; k >>= 1) mx = fmaxf(mx, __shfl_xor(mx, k, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  f32x4 e = {0.f, 0.f, 0.f, 0.f};
  if (on) {
#pragma unroll
    for (int j = 0; j < 4; ++j) e[j] = expf(v[j] - mx);
  }
  const float se = ws_block_sum(e[0] + e[1] + e[2] + e[3], red);
  if (on) *reinterpret_cast<f32x4*>(y + o) = e * (1.f / se);
}

__global__ __launch_bounds__(256) void softmax_rows_bwd4_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                                                int n, float scale, float* __restrict__ dx) {
  __shared__ float red[16];
  const long long o = (long long)blockIdx.x * n + 4 * threadIdx.x;
  const bool on = 4 * (int)threadIdx.x < n;
  f32x4 yv = {0.f, 0.f, 0.f, 0.f}, dv = yv;
  if (on) {
    yv = *reinterpret_cast<const f32x4*>(y + o);
    dv = *reinterpret_cast<const f32x4*>(dy + o);
  }
  const f32x4 p = yv * dv;
  const float sm = ws_block_sum(p[0] + p[1] + p[2] + p[3], red);
  if (on) *reinterpret_cast<f32x4*>(dx + o) = yv * (dv - sm) * scale;
}

extern "C" int ws_softmax_rows_fwd(const float* x, long long rows, int n, float scale, float* y, void* stream) {
  WS_REQUIRE(x && y && rows > 0 && rows < (1LL << 31) && n > 0, "ws_softmax_rows_fwd: bad args");
  if (n <= 1024 && n % 4 == 0 && ((size_t)x & 15) == 0 && ((size_t)y & 15) == 0) {
    hipLaunchKernelGGL(softmax_rows_fwd4_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, x, n, scale, y);
    return ws_check_launch("ws_softmax_rows_fwd");
  }
  hipLaunchKernelGGL(softmax_rows_fwd_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, x, n, scale, y);
  return ws_check_launch("ws_softmax_rows_fwd");
}

extern "C" int ws_softmax_rows_bwd(const float* y, const float* dy, long long rows, int n, float scale, float* dx,
                                   void* stream) {
  WS_REQUIRE(y && dy && dx && rows > 0 && rows < (1LL << 31) && n > 0, "ws_softmax_rows_bwd: bad args");
  if (n <= 1024 && n % 4 == 0 && ((size_t)y & 15) == 0 && ((size_t)dy & 15) == 0 && ((size_t)dx & 15) == 0) {
    hipLaunchKernelGGL(softmax_rows_bwd4_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, y, dy, n, scale, dx);
    return ws_check_launch("ws_softmax_rows_bwd");
  }
  hipLaunchKernelGGL(softmax_rows_bwd_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, y, dy, n, scale,
                     dx);
  return ws_check_launch("ws_softmax_rows_bwd");
}
