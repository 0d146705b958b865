// TF-GridNet attention heads: PReLU + all-head LayerNorm + the head-major layout, one pass (round 3).
//
// gridnet_block.py:176-199 runs, per head, Conv2d(1x1) -> PReLU -> LayerNormalization4DCF over (E, F) and then
// concatenates heads along the batch: `cat(all_Q, 0)`, `.transpose(1, 2).flatten(2)` -> [nh*B, T, E*F].  Composed from
// the library's generic entry points that was, per projection, a permute copy, a PReLU pass, a statistics pass, a
// normalise pass, a stack copy and (keys / values: the time axis padded to 16-byte rows) a pad copy forward, and about
// twice that backward -- every one a round trip of the 100-150 MB tensor through HBM.  Here one workgroup owns one
// (batch, frame) row of the projection output
//   x[(b*T + t)*Q + q][off + h*ch + e]              (the columns [off, off + nh*ch) of rows of stride ldx: the three
//                                                    projections are ONE GEMM with concatenated weights)
// keeps it in registers, and writes
//   y[((h*B + b)*Tp + t)][q*ch + e] = gamma[h][q*ch + e] * (u - mean_h) * rstd_h + beta[h][q*ch + e],  u = PReLU_h(x)
// with mean / variance over the Q*ch elements of (row, head) (two passes over the registers), zero rows for t in [T, Tp),
// and the (mean, rstd) pairs stats[h][b*T + t][2] for the backward.
// Backward: the same ownership, a workgroup walks rows r = blockIdx.x, blockIdx.x + grid, ...:
//   d = dy * gamma, n = (u - mean) * rstd,  dl = rstd * (d - mean_row(d) - n * mean_row(d * n)),  dx = dl * PReLU'(x)
// and accumulates in registers, for the elements it owns, d(gamma) = sum dy * n and d(beta) = sum dy, and per head
// d(slope) = sum over x <= 0 of dl * x; it writes one slab [2 W | 8] (W = Q*nh*ch, element order (q, h, e)) that the
// caller reduces (ws_reduce_slabs: deterministic, the grid is an argument); slab rows are 2 W + 8 floats.
#include "common.h"

#define HD_MAXH 8        // heads
// template <NW waves, NI float4 per thread>: <4, 4> takes W = Q * nh * ch <= 4096, <12, 3> W <= 9216 (the recipe's values:
// 65 bins x 128 channels)

__device__ __forceinline__ bool hd_live(int i, int nh, bool two) { return i < HD_MAXH ? i < nh : (two && i - HD_MAXH < nh); }

// sums of v[h] (and, if two, v[HD_MAXH + h]) for h < nh over the NW waves, result in every thread
template <int NW>
__device__ __forceinline__ void hd_block_sum(float (&v)[2 * HD_MAXH], int nh, bool two, float (*red)[2 * HD_MAXH], int tid) {
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int i = 0; i < 2 * HD_MAXH; ++i) {
    if (hd_live(i, nh, two)) {
      float t = v[i];
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) t += __shfl_xor(t, o, 64);
      v[i] = t;
    }
  }
  __syncthreads();                     // the previous use of red is over
  if (lane == 0)
#pragma unroll
    for (int i = 0; i < 2 * HD_MAXH; ++i)
      if (hd_live(i, nh, two)) red[wave][i] = v[i];
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2 * HD_MAXH; ++i)
    if (hd_live(i, nh, two)) {
      float t = red[0][i];
#pragma unroll
      for (int w = 1; w < NW; ++w) t += red[w][i];
      v[i] = t;
    }
}

struct HdItem {
  int q, c, h;          // position in the row, column inside the projection (h*ch + e), head
  bool on;
};

__device__ __forceinline__ HdItem hd_item(int i4, int n4, int HC, int ch) {
  HdItem it;
  it.on = i4 < n4;
  const int p = it.on ? 4 * i4 : 0;
  it.q = p / HC;
  it.c = p - it.q * HC;
  it.h = it.c / ch;
  return it;
}

template <int NW, int HD_NI>
__global__ __launch_bounds__(NW * 64) void heads_fwd_kernel(const ws_heads_args p) {
  constexpr int NTH = NW * 64;
  __shared__ float red[NW][2 * HD_MAXH];
  const int tid = threadIdx.x;
  const int b = blockIdx.x / p.Tp, t = blockIdx.x - b * p.Tp;
  const int nh = p.nh, ch = p.ch, Q = p.Q, HC = nh * ch, W = Q * HC, n4 = W >> 2, D = Q * ch;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  HdItem it[HD_NI];
#pragma unroll
  for (int k = 0; k < HD_NI; ++k) it[k] = hd_item(tid + NTH * k, n4, HC, ch);
  if (t >= p.T) {                      // the padded frames of keys / values: zero rows
#pragma unroll
    for (int k = 0; k < HD_NI; ++k)
      if (it[k].on)
        *reinterpret_cast<f32x4*>(p.y + (((long long)it[k].h * p.B + b) * p.Tp + t) * D + it[k].q * ch + (it[k].c - it[k].h * ch)) = zero4;
    return;
  }
  const long long r = (long long)b * p.T + t;
  f32x4 u[HD_NI];
  float s[2 * HD_MAXH];
#pragma unroll
  for (int i = 0; i < 2 * HD_MAXH; ++i) s[i] = 0.f;
#pragma unroll
  for (int k = 0; k < HD_NI; ++k) {
    u[k] = zero4;
    if (it[k].on) {
      const f32x4 x = *reinterpret_cast<const f32x4*>(p.x + (r * Q + it[k].q) * p.ldx + it[k].c);
      const float a = p.slope[it[k].h];
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        u[k][j] = x[j] > 0.f ? x[j] : a * x[j];
        sum += u[k][j];
      }
#pragma unroll
      for (int h = 0; h < HD_MAXH; ++h) s[h] += it[k].h == h ? sum : 0.f;
    }
  }
  hd_block_sum<NW>(s, nh, false, red, tid);
  const float invD = 1.f / (float)D;
  float mean[HD_MAXH], v2[2 * HD_MAXH];
#pragma unroll
  for (int h = 0; h < HD_MAXH; ++h) {
    mean[h] = s[h] * invD;
    v2[h] = 0.f;
    v2[HD_MAXH + h] = 0.f;
  }
#pragma unroll
  for (int k = 0; k < HD_NI; ++k)
    if (it[k].on) {
      float m = 0.f;
#pragma unroll
      for (int h = 0; h < HD_MAXH; ++h) m = it[k].h == h ? mean[h] : m;
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) sum += (u[k][j] - m) * (u[k][j] - m);
#pragma unroll
      for (int h = 0; h < HD_MAXH; ++h) v2[h] += it[k].h == h ? sum : 0.f;
    }
  hd_block_sum<NW>(v2, nh, false, red, tid);
  float rstd[HD_MAXH];
#pragma unroll
  for (int h = 0; h < HD_MAXH; ++h) rstd[h] = rsqrtf(v2[h] * invD + p.eps);
  if (tid < nh) {
    float m = 0.f, rs = 0.f;
#pragma unroll
    for (int h = 0; h < HD_MAXH; ++h) {
      m = tid == h ? mean[h] : m;
      rs = tid == h ? rstd[h] : rs;
    }
    float* st = p.stats + ((long long)tid * p.B * p.T + r) * 2;
    st[0] = m;
    st[1] = rs;
  }
#pragma unroll
  for (int k = 0; k < HD_NI; ++k)
    if (it[k].on) {
      float m = 0.f, rs = 0.f;
#pragma unroll
      for (int h = 0; h < HD_MAXH; ++h) {
        m = it[k].h == h ? mean[h] : m;
        rs = it[k].h == h ? rstd[h] : rs;
      }
      const int e = it[k].c - it[k].h * ch, gi = it[k].h * D + it[k].q * ch + e;
      const f32x4 g = *reinterpret_cast<const f32x4*>(p.gamma + gi), bt = *reinterpret_cast<const f32x4*>(p.beta + gi);
      f32x4 y;
#pragma unroll
      for (int j = 0; j < 4; ++j) y[j] = g[j] * ((u[k][j] - m) * rs) + bt[j];
      *reinterpret_cast<f32x4*>(p.y + (((long long)it[k].h * p.B + b) * p.Tp + t) * D + it[k].q * ch + e) = y;
    }
}

template <int NW, int HD_NI>
__global__ __launch_bounds__(NW * 64) void heads_bwd_kernel(const ws_heads_args p) {
  constexpr int NTH = NW * 64;
  __shared__ float red[NW][2 * HD_MAXH];
  const int tid = threadIdx.x;
  const int nh = p.nh, ch = p.ch, Q = p.Q, HC = nh * ch, W = Q * HC, n4 = W >> 2, D = Q * ch;
  const long long R = (long long)p.B * p.T;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  HdItem it[HD_NI];
  f32x4 g[HD_NI], dgam[HD_NI], dbet[HD_NI];
  float a[HD_NI];
#pragma unroll
  for (int k = 0; k < HD_NI; ++k) {
    it[k] = hd_item(tid + NTH * k, n4, HC, ch);
    dgam[k] = dbet[k] = g[k] = zero4;
    a[k] = 0.f;
    if (it[k].on) {
      g[k] = *reinterpret_cast<const f32x4*>(p.gamma + it[k].h * D + it[k].q * ch + (it[k].c - it[k].h * ch));
      a[k] = p.slope[it[k].h];
    }
  }
  float dsl[2 * HD_MAXH];
#pragma unroll
  for (int i = 0; i < 2 * HD_MAXH; ++i) dsl[i] = 0.f;
  const float invD = 1.f / (float)D;
  for (long long r = blockIdx.x; r < R; r += gridDim.x) {
    const int b = (int)(r / p.T), t = (int)(r - (long long)b * p.T);
    f32x4 x[HD_NI], n[HD_NI], d[HD_NI];
    float rs[HD_NI];
    float s[2 * HD_MAXH];              // per head: sum d | sum d * n
#pragma unroll
    for (int i = 0; i < 2 * HD_MAXH; ++i) s[i] = 0.f;
#pragma unroll
    for (int k = 0; k < HD_NI; ++k) {
      x[k] = n[k] = d[k] = zero4;
      rs[k] = 0.f;
      if (it[k].on) {
        const int e = it[k].c - it[k].h * ch;
        x[k] = *reinterpret_cast<const f32x4*>(p.x + (r * Q + it[k].q) * p.ldx + it[k].c);
        const f32x4 dy = *reinterpret_cast<const f32x4*>(p.dy + (((long long)it[k].h * p.B + b) * p.Tp + t) * D + it[k].q * ch + e);
        const float* st = p.stats + ((long long)it[k].h * R + r) * 2;
        const float m = st[0];
        rs[k] = st[1];
        float sd = 0.f, sdn = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float uu = x[k][j] > 0.f ? x[k][j] : a[k] * x[k][j];
          n[k][j] = (uu - m) * rs[k];
          d[k][j] = dy[j] * g[k][j];
          dgam[k][j] += dy[j] * n[k][j];
          dbet[k][j] += dy[j];
          sd += d[k][j];
          sdn += d[k][j] * n[k][j];
        }
#pragma unroll
        for (int h = 0; h < HD_MAXH; ++h) {
          s[h] += it[k].h == h ? sd : 0.f;
          s[HD_MAXH + h] += it[k].h == h ? sdn : 0.f;
        }
      }
    }
    hd_block_sum<NW>(s, nh, true, red, tid);
#pragma unroll
    for (int k = 0; k < HD_NI; ++k)
      if (it[k].on) {
        float md = 0.f, mdn = 0.f;
#pragma unroll
        for (int h = 0; h < HD_MAXH; ++h) {
          md = it[k].h == h ? s[h] : md;
          mdn = it[k].h == h ? s[HD_MAXH + h] : mdn;
        }
        md *= invD;
        mdn *= invD;
        f32x4 dx;
        float sl = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float dl = rs[k] * (d[k][j] - md - n[k][j] * mdn);
          const bool pos = x[k][j] > 0.f;
          dx[j] = pos ? dl : a[k] * dl;
          sl += pos ? 0.f : dl * x[k][j];
        }
#pragma unroll
        for (int h = 0; h < HD_MAXH; ++h) dsl[h] += it[k].h == h ? sl : 0.f;
        *reinterpret_cast<f32x4*>(p.dx + (r * Q + it[k].q) * p.lddx + it[k].c) = dx;
      }
  }
  float* out = p.slab + (long long)blockIdx.x * (2 * W + HD_MAXH);      // rows of 2 W + 8 floats: 16-byte aligned
#pragma unroll
  for (int k = 0; k < HD_NI; ++k)
    if (it[k].on) {
      const int i = 4 * (tid + NTH * k);
      *reinterpret_cast<f32x4*>(out + i) = dgam[k];
      *reinterpret_cast<f32x4*>(out + W + i) = dbet[k];
    }
  hd_block_sum<NW>(dsl, nh, false, red, tid);
  if (tid < HD_MAXH) {
    float v = 0.f;
#pragma unroll
    for (int h = 0; h < HD_MAXH; ++h) v = tid == h ? dsl[h] : v;
    out[2 * W + tid] = tid < nh ? v : 0.f;
  }
}

static int hd_check(const ws_heads_args* a, const char* who) {
  WS_REQUIRE(a && a->x && a->slope && a->gamma && a->stats, "%s: null pointer", who);
  WS_REQUIRE(a->B > 0 && a->T > 0 && a->Tp >= a->T && a->Q > 0 && a->nh > 0 && a->nh <= HD_MAXH && a->ch > 0 && a->ch % 4 == 0,
             "%s: B, T <= Tp, Q, nh <= %d, ch %% 4 (got nh=%d ch=%d)", who, HD_MAXH, a->nh, a->ch);
  WS_REQUIRE((long long)a->Q * a->nh * a->ch <= 9216, "%s: Q * nh * ch = %lld above 9216", who,
             (long long)a->Q * a->nh * a->ch);
  WS_REQUIRE(a->ldx >= a->nh * a->ch && a->ldx % 4 == 0, "%s: ldx >= nh * ch, %% 4", who);
  return 0;
}

extern "C" int ws_heads_fwd(const ws_heads_args* a, void* stream) {
  if (int rc = hd_check(a, "ws_heads_fwd")) return rc;
  WS_REQUIRE(a->beta && a->y, "ws_heads_fwd: null pointer");
  WS_REQUIRE((long long)a->B * a->Tp < (1LL << 31), "ws_heads_fwd: B * Tp indexes the launch grid");
  if (a->Q * a->nh * a->ch <= 4096)
    hipLaunchKernelGGL((heads_fwd_kernel<4, 4>), dim3((unsigned)(a->B * a->Tp)), dim3(256), 0, (hipStream_t)stream, *a);
  else
    hipLaunchKernelGGL((heads_fwd_kernel<12, 3>), dim3((unsigned)(a->B * a->Tp)), dim3(768), 0, (hipStream_t)stream, *a);
  return ws_check_launch("ws_heads_fwd");
}

extern "C" int ws_heads_bwd(const ws_heads_args* a, void* stream) {
  if (int rc = hd_check(a, "ws_heads_bwd")) return rc;
  WS_REQUIRE(a->dy && a->dx && a->slab && a->nwg > 0, "ws_heads_bwd: null pointer / no workgroups");
  WS_REQUIRE(a->lddx >= a->nh * a->ch && a->lddx % 4 == 0, "ws_heads_bwd: lddx >= nh * ch, %% 4");
  if (a->Q * a->nh * a->ch <= 4096)
    hipLaunchKernelGGL((heads_bwd_kernel<4, 4>), dim3((unsigned)a->nwg), dim3(256), 0, (hipStream_t)stream, *a);
  else
    hipLaunchKernelGGL((heads_bwd_kernel<12, 3>), dim3((unsigned)a->nwg), dim3(768), 0, (hipStream_t)stream, *a);
  return ws_check_launch("ws_heads_bwd");
}
