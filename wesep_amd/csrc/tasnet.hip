// Conv-TasNet / SpEx+ pieces that are not GEMMs (wesep/modules/tasnet/convs.py, encoder.py,
// decoder.py; wesep/modules/common/norm.py), on CHANNELS-LAST activations [R*T'][C]: row m = r*T' + t.
// All of them are HBM-bound elementwise / stencil / reduction passes: 16-byte accesses along C,
// grid-stride loops, wave-shuffle + LDS block reductions, per-split slabs summed by ws_reduce_slabs
// (deterministic, no atomics).  The 1x1 convolutions run on the generic MFMA GEMMs (gemm*.hip) with
// the normalisation applied on operand load.
#include "common.h"

#define TN_MAXP 7

static inline unsigned ew_blocks(long long n, int per_block) {
  long long b = (n + per_block - 1) / per_block;
  return (unsigned)(b < 1 ? 1 : (b > 32768 ? 32768 : b));
}

// ---------------------------------------------------------------------------------------------
// Chunked mean / variance of contiguous groups (gLN: one group = one row's T'*C floats,
// norm.py:29-48): each workgroup reduces one chunk to (count, mean, M2) with a two-pass sum over
// its own chunk (second pass from L2), the finalize kernel merges the chunks in fixed order
// (Chan et al.) -- as accurate as the single-workgroup two-pass kernel, but it fills the chip.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void flat_stats_chunk_kernel(const float* __restrict__ x, long long n_per_group,
                                                               int nchunk, float* __restrict__ scratch) {
  __shared__ float red[16];
  const int g = blockIdx.y, ch = blockIdx.x;
  const long long n4 = n_per_group >> 2;
  const long long per = (n4 + nchunk - 1) / nchunk;
  const long long lo = ch * per, hi = min(n4, lo + per);
  const f32x4* xb = reinterpret_cast<const f32x4*>(x + (long long)g * n_per_group);
  float s = 0.f;
  for (long long i = lo + threadIdx.x; i < hi; i += 256) {
    const f32x4 t = xb[i];
    s += (t[0] + t[1]) + (t[2] + t[3]);
  }
  const float cnt = (float)(max(hi - lo, 0LL) * 4);
  const float mean = cnt > 0.f ? ws_block_sum(s, red) / cnt : 0.f;
  float q = 0.f;
  for (long long i = lo + threadIdx.x; i < hi; i += 256) {
    const f32x4 t = xb[i] - mean;
    q += (t[0] * t[0] + t[1] * t[1]) + (t[2] * t[2] + t[3] * t[3]);
  }
  q = ws_block_sum(q, red);
  if (threadIdx.x == 0) {
    float* o = scratch + ((long long)g * nchunk + ch) * 4;
    o[0] = cnt;
    o[1] = mean;
    o[2] = q;
  }
}

// one wave per group: lane l merges chunks l, l+64, ... serially, then the 64 partial triples are merged
// by a shuffle butterfly (the pairwise merge is associative; the order is fixed, so results are reproducible)
__device__ __forceinline__ void chan_merge(double& n, double& mean, double& m2, double nb, double mb, double qb) {
  if (nb <= 0.0) return;
  const double nt = n + nb, d = mb - mean;
  mean += d * nb / nt;
  m2 += qb + d * d * n * nb / nt;
  n = nt;
}

__global__ __launch_bounds__(64) void flat_stats_final_kernel(const float* __restrict__ scratch, int ngroups,
                                                              int nchunk, float eps, float* __restrict__ stats) {
  const int g = blockIdx.x, lane = threadIdx.x;
  double n = 0.0, mean = 0.0, m2 = 0.0;
  for (int c = lane; c < nchunk; c += 64) {
    const float* s = scratch + ((long long)g * nchunk + c) * 4;
    chan_merge(n, mean, m2, s[0], s[1], s[2]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const double nb = __shfl_xor(n, o, 64), mb = __shfl_xor(mean, o, 64), qb = __shfl_xor(m2, o, 64);
    // both partners must compute the same merged triple: merge (lower lane's, upper lane's) in that order
    double an = n, am = mean, aq = m2, bn = nb, bm = mb, bq = qb;
    if (lane & o) {
      an = nb; am = mb; aq = qb; bn = n; bm = mean; bq = m2;
    }
    chan_merge(an, am, aq, bn, bm, bq);
    n = an; mean = am; m2 = aq;
  }
  if (lane == 0) {
    stats[2 * g] = (float)mean;
    stats[2 * g + 1] = 1.f / sqrtf((float)(m2 / n) + eps);
  }
}

extern "C" int ws_flat_stats(const float* x, int ngroups, long long n_per_group, float eps, int nchunk,
                             float* scratch, float* stats, void* stream) {
  WS_REQUIRE(x && scratch && stats && ngroups > 0 && n_per_group > 0 && n_per_group % 4 == 0 && nchunk > 0,
             "ws_flat_stats: bad args (n_per_group %% 4)");
  hipLaunchKernelGGL(flat_stats_chunk_kernel, dim3(nchunk, ngroups), dim3(256), 0, (hipStream_t)stream, x,
                     n_per_group, nchunk, scratch);
  hipLaunchKernelGGL(flat_stats_final_kernel, dim3(ngroups), dim3(64), 0, (hipStream_t)stream, scratch, ngroups,
                     nchunk, eps, stats);
  return ws_check_launch("ws_flat_stats");
}

// ---------------------------------------------------------------------------------------------
// PReLU with one slope (nn.PReLU(), convs.py:60,73,123,137), optionally after adding a per-(row
// group, channel) vector -- the speaker half of the concatConv 1x1 convolution (convs.py:143-148:
// conv1x1(cat[x, aux]) = conv(x) + W_e e, the second term constant over time).
//   pre = x + rb[m / rows_per_r]   (written back over x: saved for the backward)
//   y   = pre > 0 ? pre : a * pre
// ---------------------------------------------------------------------------------------------
__global__ void prelu_fwd_kernel(float* x, const float* __restrict__ rb, const float* __restrict__ a,
                                 long long rows, int C, int rows_per_r, float* __restrict__ y) {
  const float slope = a[0];
  const int c4n = C >> 2;
  const long long total = rows * c4n;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    f32x4 v = *reinterpret_cast<const f32x4*>(x + i * 4);
    if (rb) {
      const long long row = i / c4n;
      const int c = (int)(i - row * c4n) * 4;
      v += *reinterpret_cast<const f32x4*>(rb + (row / rows_per_r) * C + c);
      *reinterpret_cast<f32x4*>(x + i * 4) = v;
    }
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = v[j] > 0.f ? v[j] : slope * v[j];
    *reinterpret_cast<f32x4*>(y + i * 4) = o;
  }
}

extern "C" int ws_prelu_fwd(float* x, const float* rb, const float* a, long long rows, int C, int rows_per_r,
                            float* y, void* stream) {
  WS_REQUIRE(x && a && y && rows > 0 && C > 0 && C % 4 == 0 && (!rb || rows_per_r > 0), "ws_prelu_fwd: bad args");
  hipLaunchKernelGGL(prelu_fwd_kernel, dim3(ew_blocks(rows * (C / 4), 256)), dim3(256), 0, (hipStream_t)stream, x,
                     rb, a, rows, C, rows_per_r, y);
  return ws_check_launch("ws_prelu_fwd");
}

// dx = dy * (pre > 0 ? 1 : a) (dx may alias dy);  slab[block] = sum dy * min(pre, 0)  (-> d slope)
__global__ __launch_bounds__(256) void prelu_bwd_kernel(const float* __restrict__ pre, const float* dy,
                                                        const float* __restrict__ a, long long n4, float* dx,
                                                        float* __restrict__ slab) {
  __shared__ float red[16];
  const float slope = a[0];
  float s = 0.f;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const f32x4 p = *reinterpret_cast<const f32x4*>(pre + i * 4);
    const f32x4 g = *reinterpret_cast<const f32x4*>(dy + i * 4);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      o[j] = p[j] > 0.f ? g[j] : slope * g[j];
      s += p[j] > 0.f ? 0.f : g[j] * p[j];
    }
    *reinterpret_cast<f32x4*>(dx + i * 4) = o;
  }
  s = ws_block_sum(s, red);
  if (threadIdx.x == 0) slab[blockIdx.x] = s;
}

extern "C" int ws_prelu_bwd(const float* pre, const float* dy, const float* a, long long n, float* dx, float* slab,
                            int nslab, void* stream) {
  WS_REQUIRE(pre && dy && a && dx && slab && n > 0 && n % 4 == 0 && nslab > 0, "ws_prelu_bwd: bad args");
  hipLaunchKernelGGL(prelu_bwd_kernel, dim3(nslab), dim3(256), 0, (hipStream_t)stream, pre, dy, a, n / 4, dx, slab);
  return ws_check_launch("ws_prelu_bwd");
}

// ---------------------------------------------------------------------------------------------
// Depthwise dilated convolution (convs.py:63-70,125-133; groups = channels, "same" padding) on the NORMALISED input,
// normalisation applied on load:
//   xn[r][t][c] = (x - mean_s) * rstd_s * gamma[c] + beta[c],  s = m / st_div  (gLN: T', cLN: 1)
//   y[r][t][c]  = b[c] + sum_p w[c][p] * xn[r][t + (p - ctr) * dil][c]          (zero outside [0, T'))
// ctr = (P-1)/2 (non-causal), or P-1 for the causal blocks (convs.py:61-62,91-92: padding dil*(P-1) on both sides, the
// last dil*(P-1) outputs cut -- every tap at or before t)
// ---------------------------------------------------------------------------------------------
struct DwGeom {
  int R, Tp, C, P, dil, st_div, ctr;
};

__device__ __forceinline__ f32x4 dw_xn(const float* __restrict__ x, const float* __restrict__ stats,
                                       const f32x4& gm, const f32x4& bt, const DwGeom& g, long long row, int c) {
  const long long s = row / g.st_div;
  const float mean = stats[2 * s], rstd = stats[2 * s + 1];
  return (*reinterpret_cast<const f32x4*>(x + row * g.C + c) - mean) * rstd * gm + bt;
}

__global__ void dwconv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ stats,
                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                  const float* __restrict__ w, const float* __restrict__ b, DwGeom g,
                                  float* __restrict__ y) {
  const int c4n = g.C >> 2;
  const long long total = (long long)g.R * g.Tp * c4n;
  const int ctr = g.ctr;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long row = i / c4n;
    const int c = (int)(i - row * c4n) * 4;
    const int t = (int)(row % g.Tp);
    const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + c), bt = *reinterpret_cast<const f32x4*>(beta + c);
    f32x4 acc = *reinterpret_cast<const f32x4*>(b + c);
    for (int p = 0; p < g.P; ++p) {
      const int tt = t + (p - ctr) * g.dil;
      if (tt < 0 || tt >= g.Tp) continue;
      const f32x4 v = dw_xn(x, stats, gm, bt, g, row + (tt - t), c);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] += w[(c + j) * g.P + p] * v[j];
    }
    *reinterpret_cast<f32x4*>(y + i * 4) = acc;
  }
}

// d(xn)[r][t][c] = sum_p w[c][p] * dy[r][t - (p - ctr) * dil][c]
__global__ void dwconv_bwd_dx_kernel(const float* __restrict__ dy, const float* __restrict__ w, DwGeom g,
                                     float* __restrict__ dxn) {
  const int c4n = g.C >> 2;
  const long long total = (long long)g.R * g.Tp * c4n;
  const int ctr = g.ctr;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long row = i / c4n;
    const int c = (int)(i - row * c4n) * 4;
    const int t = (int)(row % g.Tp);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int p = 0; p < g.P; ++p) {
      const int tt = t - (p - ctr) * g.dil;
      if (tt < 0 || tt >= g.Tp) continue;
      const f32x4 v = *reinterpret_cast<const f32x4*>(dy + (row + (tt - t)) * g.C + c);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] += w[(c + j) * g.P + p] * v[j];
    }
    *reinterpret_cast<f32x4*>(dxn + i * 4) = acc;
  }
}

// slab[split][p][c] = sum_{rows of split} dy[m][c] * xn[m + (p - ctr) * dil][c]  (p < P);  slab[split][P][c] = sum dy
// threadIdx.x = channel quad (coalesced along C), threadIdx.y = row lane: the split's rows are dealt round-robin
// to the row lanes, whose partial sums meet in LDS in fixed order.
#define DW_RY 4
__global__ __launch_bounds__(1024) void dwconv_bwd_w_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ stats,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, DwGeom g,
                                                            int rows_per_split, float* __restrict__ slab) {
  extern __shared__ f32x4 dw_part[];  // [DW_RY - 1][P + 1][blockDim.x]
  const int c = (blockIdx.y * blockDim.x + threadIdx.x) * 4;
  const bool live = c < g.C;
  const int cc = live ? c : 0;
  const int split = blockIdx.x, ry = threadIdx.y;
  const long long M = (long long)g.R * g.Tp;
  const long long lo = (long long)split * rows_per_split, hi = min(M, lo + rows_per_split);
  const int ctr = g.ctr;
  const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + cc), bt = *reinterpret_cast<const f32x4*>(beta + cc);
  f32x4 acc[TN_MAXP + 1];
#pragma unroll
  for (int p = 0; p <= TN_MAXP; ++p) acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (long long row = lo + ry; row < hi; row += DW_RY) {
    const int t = (int)(row % g.Tp);
    const f32x4 d = *reinterpret_cast<const f32x4*>(dy + row * g.C + cc);
    acc[TN_MAXP] += d;
#pragma unroll
    for (int p = 0; p < TN_MAXP; ++p) {
      if (p >= g.P) break;
      const int tt = t + (p - ctr) * g.dil;
      if (tt < 0 || tt >= g.Tp) continue;
      acc[p] += d * dw_xn(x, stats, gm, bt, g, row + (tt - t), cc);
    }
  }
  const int np = g.P + 1, bx = blockDim.x;
  if (ry > 0) {
    f32x4* o = dw_part + ((long long)(ry - 1) * np) * bx + threadIdx.x;
#pragma unroll
    for (int p = 0; p < TN_MAXP; ++p)
      if (p < g.P) o[p * bx] = acc[p];
    o[g.P * bx] = acc[TN_MAXP];
  }
  __syncthreads();
  if (ry == 0 && live) {
    float* o = slab + (long long)split * np * g.C;
#pragma unroll
    for (int p = 0; p <= TN_MAXP; ++p) {
      if (p > g.P) break;
      f32x4 v = p < g.P ? acc[p] : acc[TN_MAXP];  // row P of the slab = bias gradient
      for (int k = 0; k < DW_RY - 1; ++k) v += dw_part[((long long)k * np + p) * bx + threadIdx.x];
      *reinterpret_cast<f32x4*>(o + p * g.C + c) = v;
    }
  }
}

static int dw_check(const char* who, int R, int Tp, int C, int P, int dil, int st_div) {
  WS_REQUIRE(R > 0 && Tp > 0 && C > 0 && C % 4 == 0 && P >= 1 && P <= TN_MAXP && (P & 1) && dil >= 1 && st_div > 0,
             "%s: bad geometry (C %% 4, odd P <= %d)", who, TN_MAXP);
  return WS_OK;
}

extern "C" int ws_dwconv_ex_fwd(const float* x, const float* stats, const float* gamma, const float* beta,
                                const float* w, const float* b, int R, int Tp, int C, int P, int dil, int st_div,
                                int causal, float* y, void* stream) {
  int rc = dw_check("ws_dwconv_fwd", R, Tp, C, P, dil, st_div);
  if (rc != WS_OK) return rc;
  WS_REQUIRE(x && stats && gamma && beta && w && b && y, "ws_dwconv_fwd: null pointer");
  const DwGeom g{R, Tp, C, P, dil, st_div, causal ? P - 1 : (P - 1) / 2};
  hipLaunchKernelGGL(dwconv_fwd_kernel, dim3(ew_blocks((long long)R * Tp * (C / 4), 256)), dim3(256), 0,
                     (hipStream_t)stream, x, stats, gamma, beta, w, b, g, y);
  return ws_check_launch("ws_dwconv_fwd");
}
extern "C" int ws_dwconv_fwd(const float* x, const float* stats, const float* gamma, const float* beta,
                             const float* w, const float* b, int R, int Tp, int C, int P, int dil, int st_div,
                             float* y, void* stream) {
  return ws_dwconv_ex_fwd(x, stats, gamma, beta, w, b, R, Tp, C, P, dil, st_div, 0, y, stream);
}

extern "C" int ws_dwconv_ex_bwd(const float* dy, const float* x, const float* stats, const float* gamma,
                                const float* beta, const float* w, int R, int Tp, int C, int P, int dil, int st_div,
                                int causal, float* dxn, int nsplit, int rows_per_split, float* slab, void* stream) {
  int rc = dw_check("ws_dwconv_bwd", R, Tp, C, P, dil, st_div);
  if (rc != WS_OK) return rc;
  WS_REQUIRE(dy && x && stats && gamma && beta && w && dxn && slab && nsplit > 0 && rows_per_split > 0 &&
                 (long long)nsplit * rows_per_split >= (long long)R * Tp,
             "ws_dwconv_bwd: bad args");
  const DwGeom g{R, Tp, C, P, dil, st_div, causal ? P - 1 : (P - 1) / 2};
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(dwconv_bwd_dx_kernel, dim3(ew_blocks((long long)R * Tp * (C / 4), 256)), dim3(256), 0, s, dy, w,
                     g, dxn);
  const int threads = C / 4 >= 256 ? 256 : ((C / 4 + 63) / 64) * 64;
  const size_t lds = (size_t)(DW_RY - 1) * (P + 1) * threads * sizeof(f32x4);
  hipLaunchKernelGGL(dwconv_bwd_w_kernel, dim3(nsplit, (C / 4 + threads - 1) / threads), dim3(threads, DW_RY), lds, s,
                     dy, x, stats, gamma, beta, g, rows_per_split, slab);
  return ws_check_launch("ws_dwconv_bwd");
}
extern "C" int ws_dwconv_bwd(const float* dy, const float* x, const float* stats, const float* gamma,
                             const float* beta, const float* w, int R, int Tp, int C, int P, int dil, int st_div,
                             float* dxn, int nsplit, int rows_per_split, float* slab, void* stream) {
  return ws_dwconv_ex_bwd(dy, x, stats, gamma, beta, w, R, Tp, C, P, dil, st_div, 0, dxn, nsplit, rows_per_split, slab,
                          stream);
}

// ---------------------------------------------------------------------------------------------
// Per-channel sums over row ranges of a channels-last tensor -- one pass serves the backward of the
// channel-affine norms (norm.py:29-59) and the row-bias gradient of the concatConv fusion:
//   slab[(split * ngroups + grp)][0][c] = sum_m g[m][c]
//   slab[(split * ngroups + grp)][1][c] = sum_m g[m][c] * xhat[m][c]   (x given; xhat = (x - mean_s) * rstd_s,
//                                                                      s = m / st_div; stats NULL: xhat = x)
// over the rows m of group grp ([grp * rows_per_group, +rows_per_group)) that fall into the split.
// gLN:  dbeta = sum_grp S0, dgamma = sum_grp S1, and the two group means of the norm backward are
//       sum_c gamma[c] * S{0,1}[grp][c] / n  (ws_norm_ab below).
// ---------------------------------------------------------------------------------------------
// One workgroup per (split, group): thread = (row lane, channel quad), so a 16-channel tensor (DPCCN's dense blocks:
// 4 quads) still uses all 256 threads -- round 1 ran one thread per quad and 4 of 64 lanes; the row lanes are summed
// in fixed order through LDS (deterministic).
__global__ __launch_bounds__(256) void chan_sums_kernel(const float* __restrict__ gsrc, const float* __restrict__ x,
                                                        const float* __restrict__ stats, int st_div, int rows_per_group,
                                                        int ngroups, int nsplit, int C, float* __restrict__ slab) {
  __shared__ f32x4 red[2][256];
  const int c4n = C >> 2;                      // <= 256 per z-slice
  const int cz = blockIdx.z * 256;             // first quad of this slice
  const int nq = min(256, c4n - cz);           // quads in this slice
  const int nrl = 256 / nq;                    // row lanes
  const int tid = threadIdx.x, rl = tid / nq, q = tid - rl * nq;
  const bool on = rl < nrl;
  const int c = (cz + q) * 4;
  const int split = blockIdx.x, grp = blockIdx.y;
  const int per = (rows_per_group + nsplit - 1) / nsplit;
  const int lo = split * per, hi = min(rows_per_group, lo + per);
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
  if (on) {
    for (int j = lo + rl; j < hi; j += nrl) {
      const long long row = (long long)grp * rows_per_group + j;
      const f32x4 d = *reinterpret_cast<const f32x4*>(gsrc + row * C + c);
      s0 += d;
      if (x) {
        f32x4 v = *reinterpret_cast<const f32x4*>(x + row * C + c);
        if (stats) {
          const long long s = row / st_div;
          v = (v - stats[2 * s]) * stats[2 * s + 1];
        }
        s1 += d * v;
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
    float* o = slab + ((long long)split * ngroups + grp) * 2 * C;
    *reinterpret_cast<f32x4*>(o + (cz + tid) * 4) = t0;
    *reinterpret_cast<f32x4*>(o + C + (cz + tid) * 4) = t1;
  }
}

extern "C" int ws_chan_sums(const float* g, const float* x, const float* stats, int st_div, int rows_per_group,
                            int ngroups, int nsplit, int C, float* slab, void* stream) {
  WS_REQUIRE(g && slab && rows_per_group > 0 && ngroups > 0 && nsplit > 0 && C > 0 && C % 4 == 0 &&
                 (!stats || (x && st_div > 0)),
             "ws_chan_sums: bad args");
  hipLaunchKernelGGL(chan_sums_kernel, dim3(nsplit, ngroups, (C / 4 + 255) / 256), dim3(256), 0,
                     (hipStream_t)stream, g, x, stats, st_div, rows_per_group, ngroups, nsplit, C, slab);
  return ws_check_launch("ws_chan_sums");
}

// ab[grp] = (sum_c gamma[c] * S0[grp][c], sum_c gamma[c] * S1[grp][c]) / n   from sums [ngroups][2][C]
__global__ __launch_bounds__(256) void norm_ab_kernel(const float* __restrict__ sums, const float* __restrict__ gamma,
                                                      int C, float inv_n, float* __restrict__ ab) {
  __shared__ float red[16];
  const int grp = blockIdx.x;
  const float* s = sums + (long long)grp * 2 * C;
  float a = 0.f, b = 0.f;
  for (int c = threadIdx.x; c < C; c += 256) {
    a += gamma[c] * s[c];
    b += gamma[c] * s[C + c];
  }
  a = ws_block_sum(a, red);
  b = ws_block_sum(b, red);
  if (threadIdx.x == 0) {
    ab[2 * grp] = a * inv_n;
    ab[2 * grp + 1] = b * inv_n;
  }
}

extern "C" int ws_norm_ab(const float* sums, const float* gamma, int ngroups, int C, long long n_per_group,
                          float* ab, void* stream) {
  WS_REQUIRE(sums && gamma && ab && ngroups > 0 && C > 0 && n_per_group > 0, "ws_norm_ab: bad args");
  hipLaunchKernelGGL(norm_ab_kernel, dim3(ngroups), dim3(256), 0, (hipStream_t)stream, sums, gamma, C,
                     1.f / (float)n_per_group, ab);
  return ws_check_launch("ws_norm_ab");
}

// dx = rstd_s * (dxn * gamma - ab0_s - xhat * ab1_s) (+ res),  s = m / st_div   (dx may alias dxn)
__global__ void norm_bwd_apply_cl_kernel(const float* __restrict__ x, const float* dxn,
                                         const float* __restrict__ stats, const float* __restrict__ ab,
                                         const float* __restrict__ gamma, const float* __restrict__ res,
                                         long long rows, int C, int st_div, float* dx) {
  const int c4n = C >> 2;
  const long long total = rows * c4n;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long row = i / c4n;
    const int c = (int)(i - row * c4n) * 4;
    const long long s = row / st_div;
    const float mean = stats[2 * s], rstd = stats[2 * s + 1], a0 = ab[2 * s], a1 = ab[2 * s + 1];
    const f32x4 xh = (*reinterpret_cast<const f32x4*>(x + i * 4) - mean) * rstd;
    f32x4 r = (*reinterpret_cast<const f32x4*>(dxn + i * 4) * *reinterpret_cast<const f32x4*>(gamma + c) - a0 - xh * a1) * rstd;
    if (res) r += *reinterpret_cast<const f32x4*>(res + i * 4);
    *reinterpret_cast<f32x4*>(dx + i * 4) = r;
  }
}

extern "C" int ws_norm_bwd_apply_cl(const float* x, const float* dxn, const float* stats, const float* ab,
                                    const float* gamma, const float* res, long long rows, int C, int st_div,
                                    float* dx, void* stream) {
  WS_REQUIRE(x && dxn && stats && ab && gamma && dx && rows > 0 && C > 0 && C % 4 == 0 && st_div > 0,
             "ws_norm_bwd_apply_cl: bad args");
  hipLaunchKernelGGL(norm_bwd_apply_cl_kernel, dim3(ew_blocks(rows * (C / 4), 256)), dim3(256), 0,
                     (hipStream_t)stream, x, dxn, stats, ab, gamma, res, rows, C, st_div, dx);
  return ws_check_launch("ws_norm_bwd_apply_cl");
}

// ---------------------------------------------------------------------------------------------
// Decoder mask product (decoder.py:96-101, actLayer = ReLU applied by the mask GEMM's epilogue):
//   s = w * m;   backward:  dw = ds * m (written with leading dimension ld_dw),  dm_pre = ds * w * (m > 0)
// w lives inside the encoder's concatenated [M][3N] buffer (leading dimension ldw).
// ---------------------------------------------------------------------------------------------
__global__ void maskmul_fwd_kernel(const float* __restrict__ w, long long ldw, const float* __restrict__ m,
                                   long long rows, int N, float* __restrict__ s) {
  const int n4 = N >> 2;
  const long long total = rows * n4;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long row = i / n4;
    const int c = (int)(i - row * n4) * 4;
    *reinterpret_cast<f32x4*>(s + i * 4) =
        *reinterpret_cast<const f32x4*>(w + row * ldw + c) * *reinterpret_cast<const f32x4*>(m + i * 4);
  }
}

__global__ void maskmul_bwd_kernel(const float* __restrict__ ds, const float* __restrict__ w, long long ldw,
                                   const float* __restrict__ m, long long rows, int N, float* __restrict__ dw,
                                   long long ld_dw, float* __restrict__ dm) {
  const int n4 = N >> 2;
  const long long total = rows * n4;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long row = i / n4;
    const int c = (int)(i - row * n4) * 4;
    const f32x4 g = *reinterpret_cast<const f32x4*>(ds + i * 4);
    const f32x4 mv = *reinterpret_cast<const f32x4*>(m + i * 4);
    const f32x4 wv = *reinterpret_cast<const f32x4*>(w + row * ldw + c);
    *reinterpret_cast<f32x4*>(dw + row * ld_dw + c) = g * mv;
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = mv[j] > 0.f ? g[j] * wv[j] : 0.f;
    *reinterpret_cast<f32x4*>(dm + i * 4) = o;
  }
}

extern "C" int ws_maskmul_fwd(const float* w, long long ldw, const float* m, long long rows, int N, float* s,
                              void* stream) {
  WS_REQUIRE(w && m && s && rows > 0 && N > 0 && N % 4 == 0 && ldw % 4 == 0, "ws_maskmul_fwd: bad args");
  hipLaunchKernelGGL(maskmul_fwd_kernel, dim3(ew_blocks(rows * (N / 4), 256)), dim3(256), 0, (hipStream_t)stream, w,
                     ldw, m, rows, N, s);
  return ws_check_launch("ws_maskmul_fwd");
}

extern "C" int ws_maskmul_bwd(const float* ds, const float* w, long long ldw, const float* m, long long rows, int N,
                              float* dw, long long ld_dw, float* dm, void* stream) {
  WS_REQUIRE(ds && w && m && dw && dm && rows > 0 && N > 0 && N % 4 == 0 && ldw % 4 == 0 && ld_dw % 4 == 0,
             "ws_maskmul_bwd: bad args");
  hipLaunchKernelGGL(maskmul_bwd_kernel, dim3(ew_blocks(rows * (N / 4), 256)), dim3(256), 0, (hipStream_t)stream, ds,
                     w, ldw, m, rows, N, dw, ld_dw, dm);
  return ws_check_launch("ws_maskmul_bwd");
}

// d[m][c] *= (y[m][c] > 0)   -- ReLU backward from the saved output (encoder.py:104-111), in place
__global__ void relu_mask_kernel(float* d, const float* __restrict__ y, long long n4) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4;
       i += (long long)gridDim.x * blockDim.x) {
    f32x4 g = *reinterpret_cast<const f32x4*>(d + i * 4);
    const f32x4 v = *reinterpret_cast<const f32x4*>(y + i * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) g[j] = v[j] > 0.f ? g[j] : 0.f;
    *reinterpret_cast<f32x4*>(d + i * 4) = g;
  }
}

extern "C" int ws_relu_mask(float* d, const float* y, long long n, void* stream) {
  WS_REQUIRE(d && y && n > 0 && n % 4 == 0, "ws_relu_mask: bad args");
  hipLaunchKernelGGL(relu_mask_kernel, dim3(ew_blocks(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, d, y, n / 4);
  return ws_check_launch("ws_relu_mask");
}

// ---------------------------------------------------------------------------------------------
// ConvTranspose1d(N -> 1, kernel L, stride hop) (decoder.py:79-91) = per-frame GEMM [M][N] x [N][L]
// (generic GEMM) followed by this overlap-add:
//   est[r][j] = bias + sum_{t : 0 <= j - hop*t < L} frames[r*T' + t][j - hop*t],   j < Tout (cropped)
// and its adjoint, the frame gather  dframes[m][k] = (hop*t + k < Tout) ? dest[r][hop*t + k] : 0.
// ---------------------------------------------------------------------------------------------
__global__ void ola_fwd_kernel(const float* __restrict__ frames, const float* __restrict__ bias, int R, int Tp,
                               int L, int hop, int Tout, float* __restrict__ est) {
  const long long total = (long long)R * Tout;
  const float bv = bias ? bias[0] : 0.f;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(i / Tout), j = (int)(i - (long long)r * Tout);
    int t_hi = j / hop;
    if (t_hi > Tp - 1) t_hi = Tp - 1;
    int t_lo = (j - L + hop) / hop;  // ceil((j - L + 1) / hop)
    if (j - L + 1 <= 0) t_lo = 0;
    float acc = bv;
    for (int t = t_lo; t <= t_hi; ++t) acc += frames[((long long)r * Tp + t) * L + (j - hop * t)];
    est[i] = acc;
  }
}

__global__ void ola_bwd_kernel(const float* __restrict__ dest, int R, int Tp, int L, int hop, int Tout,
                               float* __restrict__ dframes) {
  const long long total = (long long)R * Tp * L;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long m = i / L;
    const int k = (int)(i - m * L);
    const int r = (int)(m / Tp), t = (int)(m - (long long)r * Tp);
    const int j = hop * t + k;
    dframes[i] = j < Tout ? dest[(long long)r * Tout + j] : 0.f;
  }
}

extern "C" int ws_ola_fwd(const float* frames, const float* bias, int R, int Tp, int L, int hop, int Tout,
                          float* est, void* stream) {
  WS_REQUIRE(frames && est && R > 0 && Tp > 0 && L > 0 && hop > 0 && Tout > 0 && Tout <= (Tp - 1) * hop + L,
             "ws_ola_fwd: bad args");
  hipLaunchKernelGGL(ola_fwd_kernel, dim3(ew_blocks((long long)R * Tout, 256)), dim3(256), 0, (hipStream_t)stream,
                     frames, bias, R, Tp, L, hop, Tout, est);
  return ws_check_launch("ws_ola_fwd");
}

extern "C" int ws_ola_bwd(const float* dest, int R, int Tp, int L, int hop, int Tout, float* dframes, void* stream) {
  WS_REQUIRE(dest && dframes && R > 0 && Tp > 0 && L > 0 && hop > 0 && Tout > 0, "ws_ola_bwd: bad args");
  hipLaunchKernelGGL(ola_bwd_kernel, dim3(ew_blocks((long long)R * Tp * L, 256)), dim3(256), 0, (hipStream_t)stream,
                     dest, R, Tp, L, hop, Tout, dframes);
  return ws_check_launch("ws_ola_bwd");
}

// slab[block] = sum of x over the block's grid-stride share (-> ws_reduce_slabs(count = 1))
__global__ __launch_bounds__(256) void sum_partial_kernel(const float* __restrict__ x, long long n,
                                                          float* __restrict__ slab) {
  __shared__ float red[16];
  float s = 0.f;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256) s += x[i];
  s = ws_block_sum(s, red);
  if (threadIdx.x == 0) slab[blockIdx.x] = s;
}

extern "C" int ws_sum_partial(const float* x, long long n, float* slab, int nslab, void* stream) {
  WS_REQUIRE(x && slab && n > 0 && nslab > 0, "ws_sum_partial: bad args");
  hipLaunchKernelGGL(sum_partial_kernel, dim3(nslab), dim3(256), 0, (hipStream_t)stream, x, n, slab);
  return ws_check_launch("ws_sum_partial");
}

// =============================================================================================
// SpEx+ speaker encoder pieces (wesep/modules/tasnet/speaker.py:7-64), channels-last [M][C]
// =============================================================================================

// Column reductions over M rows of a channels-last [M][C] tensor.  A 256-thread workgroup = LX channel quads
// (LX = C/4 rounded up to a power of two, <= 256) x RY = 256/LX row lanes: with C = 32 (the first ResNet stage,
// 1 M rows) a one-thread-per-quad layout would leave 7/8 of every wave idle.  A wave reads 64/LX consecutive
// rows = one contiguous KB; the row lanes meet in LDS in fixed order.
__device__ __forceinline__ int bn_lx(int C) {
  int lx = 1;
  while (lx < (C >> 2)) lx <<= 1;
  return lx > 256 ? 256 : lx;
}

// slab[split][c] = sum over the split's rows of (shift ? (x - shift[c])^2 : x)
__global__ __launch_bounds__(256) void bn_partial_kernel(const float* __restrict__ x, const float* __restrict__ shift,
                                                         long long M, int C, int rows_per_split,
                                                         float* __restrict__ slab) {
  __shared__ f32x4 part[256];
  const int LX = bn_lx(C), RY = 256 / LX;
  const int lx = threadIdx.x % LX, ly = threadIdx.x / LX;
  const int c = (blockIdx.y * LX + lx) * 4;
  const bool live = c < C;
  const int cc = live ? c : 0;
  const long long lo = (long long)blockIdx.x * rows_per_split, hi = min(M, lo + rows_per_split);
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (shift) {
    const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + cc);
    for (long long r = lo + ly; r < hi; r += RY) {
      const f32x4 d = *reinterpret_cast<const f32x4*>(x + r * C + cc) - sh;
      s += d * d;
    }
  } else {
    for (long long r = lo + ly; r < hi; r += RY) s += *reinterpret_cast<const f32x4*>(x + r * C + cc);
  }
  part[threadIdx.x] = s;
  __syncthreads();
  if (ly == 0 && live) {
    for (int k = 1; k < RY; ++k) s += part[k * LX + lx];
    *reinterpret_cast<f32x4*>(slab + (long long)blockIdx.x * C + c) = s;
  }
}

// phase 0: stats[0][c] = mean.  phase 1: stats[1][c] = 1/sqrt(var + eps) and the running statistics
// (nn.BatchNorm training mode: biased variance normalises, unbiased variance is tracked).
// 64 channels x 4 split lanes per workgroup.
__global__ __launch_bounds__(256) void bn_final_kernel(const float* __restrict__ slab, int nsplit, int C, long long M,
                                                       int phase, float eps, float momentum,
                                                       float* __restrict__ stats, float* running_mean,
                                                       float* running_var) {
  __shared__ float part[4][64];
  const int x = threadIdx.x & 63, y = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + x;
  float s = 0.f;
  if (c < C)
    for (int k = y; k < nsplit; k += 4) s += slab[(long long)k * C + c];
  part[y][x] = s;
  __syncthreads();
  if (y != 0 || c >= C) return;
  s = (part[0][x] + part[1][x]) + (part[2][x] + part[3][x]);
  if (phase == 0) {
    stats[c] = s / (float)M;
  } else {
    const float var = s / (float)M;
    stats[C + c] = 1.f / sqrtf(var + eps);
    if (running_mean) {
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * stats[c];
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (M > 1 ? s / (float)(M - 1) : var);
    }
  }
}

extern "C" int ws_bn_stats(const float* x, long long M, int C, float eps, float momentum, float* running_mean,
                           float* running_var, int nsplit, float* scratch, float* stats, void* stream) {
  WS_REQUIRE(x && scratch && stats && M > 0 && C > 0 && C % 4 == 0 && nsplit > 0 && (!running_mean == !running_var),
             "ws_bn_stats: bad args");
  hipStream_t s = (hipStream_t)stream;
  const int rps = (int)((M + nsplit - 1) / nsplit);
  int lx = 1;
  while (lx < C / 4) lx <<= 1;
  if (lx > 256) lx = 256;
  const dim3 grid(nsplit, (C / 4 + lx - 1) / lx);
  hipLaunchKernelGGL(bn_partial_kernel, grid, dim3(256), 0, s, x, (const float*)nullptr, M, C, rps, scratch);
  hipLaunchKernelGGL(bn_final_kernel, dim3((C + 63) / 64), dim3(256), 0, s, scratch, nsplit, C, M, 0, eps, momentum,
                     stats, running_mean, running_var);
  hipLaunchKernelGGL(bn_partial_kernel, grid, dim3(256), 0, s, x, (const float*)stats, M, C, rps, scratch);
  hipLaunchKernelGGL(bn_final_kernel, dim3((C + 63) / 64), dim3(256), 0, s, scratch, nsplit, C, M, 1, eps, momentum,
                     stats, running_mean, running_var);
  return ws_check_launch("ws_bn_stats");
}

// u = gamma * (x - mean_c) * rstd_c + beta (+ res);  y = u > 0 ? u : a * u      (stats = [mean | rstd], [2][C])
__global__ void bn_prelu_fwd_kernel(const float* __restrict__ x, const float* __restrict__ stats,
                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                    const float* __restrict__ res, const float* __restrict__ a, long long M, int C,
                                    float* __restrict__ u, float* __restrict__ y) {
  const float slope = a[0];
  const int c4n = C >> 2;
  const long long total = M * c4n;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % c4n) * 4;
    f32x4 v = (*reinterpret_cast<const f32x4*>(x + i * 4) - *reinterpret_cast<const f32x4*>(stats + c)) *
                  *reinterpret_cast<const f32x4*>(stats + C + c) * *reinterpret_cast<const f32x4*>(gamma + c) +
              *reinterpret_cast<const f32x4*>(beta + c);
    if (res) v += *reinterpret_cast<const f32x4*>(res + i * 4);
    *reinterpret_cast<f32x4*>(u + i * 4) = v;
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = v[j] > 0.f ? v[j] : slope * v[j];
    *reinterpret_cast<f32x4*>(y + i * 4) = o;
  }
}

extern "C" int ws_bn_prelu_fwd(const float* x, const float* stats, const float* gamma, const float* beta,
                               const float* res, const float* a, long long M, int C, float* u, float* y,
                               void* stream) {
  WS_REQUIRE(x && stats && gamma && beta && a && u && y && M > 0 && C > 0 && C % 4 == 0, "ws_bn_prelu_fwd: bad args");
  hipLaunchKernelGGL(bn_prelu_fwd_kernel, dim3(ew_blocks(M * (C / 4), 256)), dim3(256), 0, (hipStream_t)stream, x,
                     stats, gamma, beta, res, a, M, C, u, y);
  return ws_check_launch("ws_bn_prelu_fwd");
}

// slab[split][0][c] = sum du, slab[split][1][c] = sum du * xhat,  xhat = (x - mean_c) * rstd_c
// (same LX x RY workgroup shape as bn_partial_kernel)
__global__ __launch_bounds__(256) void bn_bwd_sums_kernel(const float* __restrict__ x, const float* __restrict__ du,
                                                          const float* __restrict__ stats, long long M, int C,
                                                          int rows_per_split, float* __restrict__ slab) {
  __shared__ f32x4 part[2][256];
  const int LX = bn_lx(C), RY = 256 / LX;
  const int lx = threadIdx.x % LX, ly = threadIdx.x / LX;
  const int c = (blockIdx.y * LX + lx) * 4;
  const bool live = c < C;
  const int cc = live ? c : 0;
  const long long lo = (long long)blockIdx.x * rows_per_split, hi = min(M, lo + rows_per_split);
  const f32x4 mean = *reinterpret_cast<const f32x4*>(stats + cc), rstd = *reinterpret_cast<const f32x4*>(stats + C + cc);
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
  for (long long r = lo + ly; r < hi; r += RY) {
    const f32x4 d = *reinterpret_cast<const f32x4*>(du + r * C + cc);
    s0 += d;
    s1 += d * ((*reinterpret_cast<const f32x4*>(x + r * C + cc) - mean) * rstd);
  }
  part[0][threadIdx.x] = s0;
  part[1][threadIdx.x] = s1;
  __syncthreads();
  if (ly == 0 && live) {
    for (int k = 1; k < RY; ++k) {
      s0 += part[0][k * LX + lx];
      s1 += part[1][k * LX + lx];
    }
    float* o = slab + (long long)blockIdx.x * 2 * C;
    *reinterpret_cast<f32x4*>(o + c) = s0;
    *reinterpret_cast<f32x4*>(o + C + c) = s1;
  }
}

// dx = gamma * rstd * (du - sums[0]/M - xhat * sums[1]/M)     (dx may alias du)
__global__ void bn_bwd_apply_kernel(const float* __restrict__ x, const float* du, const float* __restrict__ stats,
                                    const float* __restrict__ gamma, const float* __restrict__ sums, long long M,
                                    int C, float* dx) {
  const int c4n = C >> 2;
  const long long total = M * c4n;
  const float inv = 1.f / (float)M;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % c4n) * 4;
    const f32x4 rstd = *reinterpret_cast<const f32x4*>(stats + C + c);
    const f32x4 xh = (*reinterpret_cast<const f32x4*>(x + i * 4) - *reinterpret_cast<const f32x4*>(stats + c)) * rstd;
    const f32x4 r = *reinterpret_cast<const f32x4*>(gamma + c) * rstd *
                    (*reinterpret_cast<const f32x4*>(du + i * 4) - *reinterpret_cast<const f32x4*>(sums + c) * inv -
                     xh * (*reinterpret_cast<const f32x4*>(sums + C + c) * inv));
    *reinterpret_cast<f32x4*>(dx + i * 4) = r;
  }
}

extern "C" int ws_bn_bwd(const float* x, const float* du, const float* stats, const float* gamma, long long M, int C,
                         int nsplit, float* slab, float* sums, float* dx, void* stream) {
  WS_REQUIRE(x && du && stats && gamma && slab && sums && dx && M > 0 && C > 0 && C % 4 == 0 && nsplit > 0,
             "ws_bn_bwd: bad args");
  hipStream_t s = (hipStream_t)stream;
  const int rps = (int)((M + nsplit - 1) / nsplit);
  int lx = 1;
  while (lx < C / 4) lx <<= 1;
  if (lx > 256) lx = 256;
  hipLaunchKernelGGL(bn_bwd_sums_kernel, dim3(nsplit, (C / 4 + lx - 1) / lx), dim3(256), 0, s, x, du, stats, M, C, rps,
                     slab);
  int rc = ws_reduce_slabs(slab, nsplit, 2LL * C, 2LL * C, sums, 0, 0, stream);
  if (rc != WS_OK) return rc;
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_blocks(M * (C / 4), 256)), dim3(256), 0, s, x, du, stats, gamma,
                     sums, M, C, dx);
  return ws_check_launch("ws_bn_bwd");
}

// MaxPool1d(3) over time (stride 3, floor): y[r][t'][c] = max_j x[r][3t'+j][c];  backward routes the gradient
// to the first maximal position (torch's tie rule), zero elsewhere and on the dropped tail rows.
__global__ void maxpool3_fwd_kernel(const float* __restrict__ x, int R, int T, int C, float* __restrict__ y) {
  const int To = T / 3, c4n = C >> 2;
  const long long total = (long long)R * To * c4n;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long row = i / c4n;
    const int c = (int)(i - row * c4n) * 4;
    const int r = (int)(row / To), t = (int)(row - (long long)r * To);
    const float* b = x + ((long long)r * T + 3 * t) * C + c;
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(b), v1 = *reinterpret_cast<const f32x4*>(b + C),
                v2 = *reinterpret_cast<const f32x4*>(b + 2 * C);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = fmaxf(fmaxf(v0[j], v1[j]), v2[j]);
    *reinterpret_cast<f32x4*>(y + i * 4) = o;
  }
}

__global__ void maxpool3_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, int R, int T, int C,
                                    float* __restrict__ dx) {
  const int To = T / 3, c4n = C >> 2;
  const long long total = (long long)R * T * c4n;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long row = i / c4n;
    const int c = (int)(i - row * c4n) * 4;
    const int r = (int)(row / T), t = (int)(row - (long long)r * T);
    const int to = t / 3, j = t - 3 * to;
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    if (to < To) {
      const float* b = x + ((long long)r * T + 3 * to) * C + c;
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(b), v1 = *reinterpret_cast<const f32x4*>(b + C),
                  v2 = *reinterpret_cast<const f32x4*>(b + 2 * C);
      const f32x4 g = *reinterpret_cast<const f32x4*>(dy + ((long long)r * To + to) * C + c);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int arg = (v0[k] >= v1[k] && v0[k] >= v2[k]) ? 0 : (v1[k] >= v2[k] ? 1 : 2);
        o[k] = arg == j ? g[k] : 0.f;
      }
    }
    *reinterpret_cast<f32x4*>(dx + i * 4) = o;
  }
}

extern "C" int ws_maxpool3_fwd(const float* x, int R, int T, int C, float* y, void* stream) {
  WS_REQUIRE(x && y && R > 0 && T >= 3 && C > 0 && C % 4 == 0, "ws_maxpool3_fwd: bad args (T >= 3, C %% 4)");
  hipLaunchKernelGGL(maxpool3_fwd_kernel, dim3(ew_blocks((long long)R * (T / 3) * (C / 4), 256)), dim3(256), 0,
                     (hipStream_t)stream, x, R, T, C, y);
  return ws_check_launch("ws_maxpool3_fwd");
}

extern "C" int ws_maxpool3_bwd(const float* x, const float* dy, int R, int T, int C, float* dx, void* stream) {
  WS_REQUIRE(x && dy && dx && R > 0 && T >= 3 && C > 0 && C % 4 == 0, "ws_maxpool3_bwd: bad args");
  hipLaunchKernelGGL(maxpool3_bwd_kernel, dim3(ew_blocks((long long)R * T * (C / 4), 256)), dim3(256), 0,
                     (hipStream_t)stream, x, dy, R, T, C, dx);
  return ws_check_launch("ws_maxpool3_bwd");
}

// out[m][c] = scale * src[m / rows_per_r][c]   (adjoint of the mean over time, speaker.py:62-63)
__global__ void bcast_rows_kernel(const float* __restrict__ src, float scale, int rows_per_r, long long M, int C,
                                  float* __restrict__ out) {
  const int c4n = C >> 2;
  const long long total = M * c4n;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long row = i / c4n;
    const int c = (int)(i - row * c4n) * 4;
    *reinterpret_cast<f32x4*>(out + i * 4) = *reinterpret_cast<const f32x4*>(src + (row / rows_per_r) * C + c) * scale;
  }
}

extern "C" int ws_bcast_rows(const float* src, float scale, int rows_per_r, long long M, int C, float* out,
                             void* stream) {
  WS_REQUIRE(src && out && rows_per_r > 0 && M > 0 && C > 0 && C % 4 == 0, "ws_bcast_rows: bad args");
  hipLaunchKernelGGL(bcast_rows_kernel, dim3(ew_blocks(M * (C / 4), 256)), dim3(256), 0, (hipStream_t)stream, src,
                     scale, rows_per_r, M, C, out);
  return ws_check_launch("ws_bcast_rows");
}

// nn.CrossEntropyLoss (mean reduction) on [R][S] logits with int64 labels (losses.py:11, executor.py:112-118):
// loss = mean_r (logsumexp_r - logit_r[label_r]);  dlogits = (softmax - onehot) / R.  One workgroup.
__global__ __launch_bounds__(256) void ce_kernel(const float* __restrict__ logits, const long long* __restrict__ label,
                                                 int R, int S, float* __restrict__ loss, float* __restrict__ dlogits) {
  __shared__ float red[16];
  float total = 0.f;
  for (int r = 0; r < R; ++r) {
    const float* z = logits + (long long)r * S;
    float mx = -3.4e38f;
    for (int j = threadIdx.x; j < S; j += 256) mx = fmaxf(mx, z[j]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float se = 0.f;
    for (int j = threadIdx.x; j < S; j += 256) se += expf(z[j] - mx);
    se = ws_block_sum(se, red);
    const int lb = (int)label[r];
    const float lse = mx + logf(se);
    total += lse - z[lb];
    for (int j = threadIdx.x; j < S; j += 256)
      dlogits[(long long)r * S + j] = (expf(z[j] - lse) - (j == lb ? 1.f : 0.f)) / (float)R;
  }
  if (threadIdx.x == 0) loss[0] = total / (float)R;
}

extern "C" int ws_cross_entropy(const float* logits, const long long* label, int R, int S, float* loss,
                                float* dlogits, void* stream) {
  WS_REQUIRE(logits && label && loss && dlogits && R > 0 && S > 0, "ws_cross_entropy: bad args");
  hipLaunchKernelGGL(ce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, label, R, S, loss, dlogits);
  return ws_check_launch("ws_cross_entropy");
}
