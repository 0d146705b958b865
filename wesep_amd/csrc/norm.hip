// GroupNorm(1, C, eps) statistics and backward for the strided "group" geometries of pBSRNN
// (wesep/models/bsrnn.py:26 ResRNN.norm, :256 BN[i][0], :275 mask[i][0]).  The forward
// normalisation itself is fused into the consuming GEMM's operand load (gemm.hip); here are
// the reductions: two-pass mean/variance (eps = FLT_EPSILON makes a one-pass E[x^2]-E[x]^2
// unsafe), the two backward group means, and the dgamma/dbeta column sums.  All HBM-bound:
// one workgroup per group, coalesced row reads, wave-shuffle + LDS reductions.
#include "common.h"

struct GroupView {
  long long base;
  int W;
  int band;
};

__device__ __forceinline__ GroupView group_view(const ws_groups_geom& geo, int g) {
  GroupView v;
  v.band = g % geo.nbands;
  v.W = geo.band_w ? geo.band_w[v.band] : geo.W;
  v.base = (long long)(g / geo.gdiv) * geo.gs1 + (long long)(g % geo.gdiv) * geo.gs2 +
           (geo.band_off ? geo.band_off[v.band] : 0);
  return v;
}

// V4: uniform width, W and every stride multiples of 4 floats -> 16-byte accesses (the ResRNN / mask
// geometries); the scalar path serves the ragged per-band widths of the band-split norm.
static bool geom_vec4(const ws_groups_geom* g) {
  return !g->band_w && !g->band_off && g->W % 4 == 0 && g->rs % 4 == 0 && g->gs1 % 4 == 0 && g->gs2 % 4 == 0;
}

template <bool V4>
__global__ __launch_bounds__(256) void group_stats_kernel(const float* __restrict__ x,
                                                          const ws_groups_geom geo, float eps,
                                                          float* __restrict__ stats) {
  __shared__ float red[16];
  const int g = blockIdx.x;
  const GroupView v = group_view(geo, g);
  const int n = geo.L * v.W;
  const float* xb = x + v.base;
  float s = 0.f;
  if (V4) {
    const int w4 = v.W >> 2;
    for (int i = threadIdx.x; i < (n >> 2); i += 256) {
      const int row = i / w4, c4 = i - row * w4;
      const f32x4 t = *reinterpret_cast<const f32x4*>(xb + (long long)row * geo.rs + 4 * c4);
      s += (t[0] + t[1]) + (t[2] + t[3]);
    }
  } else {
    for (int i = threadIdx.x; i < n; i += 256) {
      const int row = i / v.W, col = i - row * v.W;
      s += xb[(long long)row * geo.rs + col];
    }
  }
  const float mean = ws_block_sum(s, red) / (float)n;
  float q = 0.f;
  if (V4) {
    const int w4 = v.W >> 2;
    for (int i = threadIdx.x; i < (n >> 2); i += 256) {
      const int row = i / w4, c4 = i - row * w4;
      const f32x4 t = *reinterpret_cast<const f32x4*>(xb + (long long)row * geo.rs + 4 * c4) - mean;
      q += (t[0] * t[0] + t[1] * t[1]) + (t[2] * t[2] + t[3] * t[3]);
    }
  } else
  for (int i = threadIdx.x; i < n; i += 256) {
    const int row = i / v.W, col = i - row * v.W;
    const float dv = xb[(long long)row * geo.rs + col] - mean;
    q += dv * dv;
  }
  const float var = ws_block_sum(q, red) / (float)n;
  if (threadIdx.x == 0) {
    stats[2 * (long long)g] = mean;
    stats[2 * (long long)g + 1] = 1.f / sqrtf(var + eps);
  }
}

static int geom_check(const ws_groups_geom* geo, const char* who) {
  WS_REQUIRE(geo && geo->ngroups > 0 && geo->gdiv > 0 && geo->L > 0 && geo->W > 0 && geo->nbands > 0,
             "%s: bad geometry", who);
  return WS_OK;
}

extern "C" int ws_group_stats(const float* x, const ws_groups_geom* geo, float eps, float* stats,
                              void* stream) {
  int rc = geom_check(geo, "ws_group_stats");
  if (rc != WS_OK) return rc;
  WS_REQUIRE(x && stats, "ws_group_stats: null pointer");
  if (geom_vec4(geo))
    hipLaunchKernelGGL((group_stats_kernel<true>), dim3(geo->ngroups), dim3(256), 0, (hipStream_t)stream, x,
                       *geo, eps, stats);
  else
    hipLaunchKernelGGL((group_stats_kernel<false>), dim3(geo->ngroups), dim3(256), 0, (hipStream_t)stream, x,
                       *geo, eps, stats);
  return ws_check_launch("ws_group_stats");
}

template <bool V4>
__global__ __launch_bounds__(256) void gn_bwd_reduce_kernel(
    const float* __restrict__ x, const float* __restrict__ dxn, const float* __restrict__ stats,
    const float* __restrict__ gamma, const float* const* __restrict__ gamma_tab,
    const ws_groups_geom geo, float* __restrict__ ab) {
  __shared__ float red[16];
  const int g = blockIdx.x;
  const GroupView v = group_view(geo, g);
  const float* gm = gamma_tab ? gamma_tab[v.band] : gamma;
  const int n = geo.L * v.W;
  const float mean = stats[2 * (long long)g], rstd = stats[2 * (long long)g + 1];
  float s1 = 0.f, s2 = 0.f;
  if (V4) {
    const int w4 = v.W >> 2;
    // (a 4-way unrolled version with eight loads in flight per thread measured SLOWER in the step: 920 vs 500 us)
    for (int i = threadIdx.x; i < (n >> 2); i += 256) {
      const int row = i / w4, c4 = i - row * w4;
      const long long o = v.base + (long long)row * geo.rs + 4 * c4;
      const f32x4 dg = *reinterpret_cast<const f32x4*>(dxn + o) * *reinterpret_cast<const f32x4*>(gm + 4 * c4);
      const f32x4 xh = (*reinterpret_cast<const f32x4*>(x + o) - mean) * rstd;
      s1 += (dg[0] + dg[1]) + (dg[2] + dg[3]);
      s2 += (dg[0] * xh[0] + dg[1] * xh[1]) + (dg[2] * xh[2] + dg[3] * xh[3]);
    }
  } else
  for (int i = threadIdx.x; i < n; i += 256) {
    const int row = i / v.W, col = i - row * v.W;
    const long long o = v.base + (long long)row * geo.rs + col;
    const float dg = dxn[o] * gm[col];
    s1 += dg;
    s2 += dg * (x[o] - mean) * rstd;
  }
  s1 = ws_block_sum(s1, red);
  s2 = ws_block_sum(s2, red);
  if (threadIdx.x == 0) {
    ab[2 * (long long)g] = s1 / (float)n;
    ab[2 * (long long)g + 1] = s2 / (float)n;
  }
}

extern "C" int ws_gn_bwd_reduce(const float* x, const float* dxn, const float* stats,
                                const float* gamma, const float* const* gamma_tab,
                                const ws_groups_geom* geo, float* ab, void* stream) {
  int rc = geom_check(geo, "ws_gn_bwd_reduce");
  if (rc != WS_OK) return rc;
  WS_REQUIRE(x && dxn && stats && ab && (gamma || gamma_tab), "ws_gn_bwd_reduce: null pointer");
  if (geom_vec4(geo))
    hipLaunchKernelGGL((gn_bwd_reduce_kernel<true>), dim3(geo->ngroups), dim3(256), 0, (hipStream_t)stream,
                       x, dxn, stats, gamma, gamma_tab, *geo, ab);
  else
    hipLaunchKernelGGL((gn_bwd_reduce_kernel<false>), dim3(geo->ngroups), dim3(256), 0, (hipStream_t)stream,
                       x, dxn, stats, gamma, gamma_tab, *geo, ab);
  return ws_check_launch("ws_gn_bwd_reduce");
}

// Apply + PARAMETER SUMS in one pass (round 4; single-band groups of 128-float rows -- the time view of ResRNN.norm): the
// apply pass reads x and dxn anyway, so dgamma[c] = sum dxn * xhat and dbeta[c] = sum dxn accumulate on the way (a thread
// keeps one column quad: 256 threads = 8 row lanes x 32 quads), one partial [2][128] per group goes to `pslab`, and the last
// workgroup of the launch adds the partials up in group order into `pout` [2][128] (ws_last_block): ws_gn_param_grad's pass
// over x and dxn and the ws_reduce_slabs launch behind it are gone.
__global__ __launch_bounds__(256) void gn_bwd_apply_pg_kernel(
    const float* __restrict__ x, const float* dxn, const float* __restrict__ stats, const float* __restrict__ ab,
    const float* __restrict__ gamma, const float* __restrict__ res, const ws_groups_geom geo, float* dx,
    float* __restrict__ pslab, float* __restrict__ pout, unsigned* counter) {
  __shared__ f32x4 sh[2][8][32];
  const int g = blockIdx.x;
  const GroupView v = group_view(geo, g);
  const int n = geo.L * 128;
  const float mean = stats[2 * (long long)g], rstd = stats[2 * (long long)g + 1];
  const float a0 = ab[2 * (long long)g], a1 = ab[2 * (long long)g + 1];
  const int c4 = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + 4 * c4);
  f32x4 sg = {0.f, 0.f, 0.f, 0.f}, sb = {0.f, 0.f, 0.f, 0.f};
  for (int row = rl; row < geo.L; row += 8) {
    const long long o = v.base + (long long)row * geo.rs + 4 * c4;
    const f32x4 d = *reinterpret_cast<const f32x4*>(dxn + o);
    const f32x4 xh = (*reinterpret_cast<const f32x4*>(x + o) - mean) * rstd;
    f32x4 r = (d * gm - a0 - xh * a1) * rstd;
    if (res) r += *reinterpret_cast<const f32x4*>(res + o);
    *reinterpret_cast<f32x4*>(dx + o) = r;
    sg += d * xh;
    sb += d;
  }
  (void)n;
  sh[0][rl][c4] = sg;
  sh[1][rl][c4] = sb;
  __syncthreads();
  if (rl < 2) {  // rl 0 -> dgamma, rl 1 -> dbeta: the eight row lanes in fixed order
    f32x4 t = sh[rl][0][c4];
#pragma unroll
    for (int r = 1; r < 8; ++r) t += sh[rl][r][c4];
    float* o = pslab + ((long long)g * 2 + rl) * 128 + 4 * c4;  // agent-scope stores: read by another workgroup (common.h)
#pragma unroll
    for (int q = 0; q < 4; ++q) ws_st_agent(o + q, t[q]);
  }
  ws_tree_sum256(pslab, gridDim.x, pout, counter);  // 256 threads = the 256 outputs (dgamma | dbeta)
}

extern "C" int ws_gn_bwd_apply_pg(const float* x, const float* dxn, const float* stats, const float* ab,
                                  const float* gamma, const float* res, const ws_groups_geom* geo, float* dx,
                                  float* pslab, float* pout, unsigned* counter, void* stream) {
  int rc = geom_check(geo, "ws_gn_bwd_apply_pg");
  if (rc != WS_OK) return rc;
  WS_REQUIRE(x && dxn && stats && ab && dx && gamma && pslab && pout && counter, "ws_gn_bwd_apply_pg: null pointer");
  WS_REQUIRE(geom_vec4(geo) && geo->nbands == 1 && geo->W == 128, "ws_gn_bwd_apply_pg: single-band groups of 128-float rows");
  hipLaunchKernelGGL(gn_bwd_apply_pg_kernel, dim3(geo->ngroups), dim3(256), 0, (hipStream_t)stream, x, dxn, stats, ab,
                     gamma, res, *geo, dx, pslab, pout, counter);
  return ws_check_launch("ws_gn_bwd_apply_pg");
}

template <bool V4>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(
    const float* __restrict__ x, const float* dxn, const float* __restrict__ stats,
    const float* __restrict__ ab, const float* __restrict__ gamma,
    const float* const* __restrict__ gamma_tab, const float* __restrict__ res,
    const ws_groups_geom geo, float* dx) {
  const int g = blockIdx.x;
  const GroupView v = group_view(geo, g);
  const float* gm = gamma_tab ? gamma_tab[v.band] : gamma;
  const int n = geo.L * v.W;
  const float mean = stats[2 * (long long)g], rstd = stats[2 * (long long)g + 1];
  const float a0 = ab[2 * (long long)g], a1 = ab[2 * (long long)g + 1];
  if (V4) {
    const int w4 = v.W >> 2;
    for (int i = threadIdx.x; i < (n >> 2); i += 256) {
      const int row = i / w4, c4 = i - row * w4;
      const long long o = v.base + (long long)row * geo.rs + 4 * c4;
      const f32x4 xh = (*reinterpret_cast<const f32x4*>(x + o) - mean) * rstd;
      f32x4 r = (*reinterpret_cast<const f32x4*>(dxn + o) * *reinterpret_cast<const f32x4*>(gm + 4 * c4) - a0 - xh * a1) * rstd;
      if (res) r += *reinterpret_cast<const f32x4*>(res + o);
      *reinterpret_cast<f32x4*>(dx + o) = r;
    }
    return;
  }
  for (int i = threadIdx.x; i < n; i += 256) {
    const int row = i / v.W, col = i - row * v.W;
    const long long o = v.base + (long long)row * geo.rs + col;
    const float xh = (x[o] - mean) * rstd;
    float r = rstd * (dxn[o] * gm[col] - a0 - xh * a1);
    if (res) r += res[o];
    dx[o] = r;
  }
}

extern "C" int ws_gn_bwd_apply(const float* x, const float* dxn, const float* stats,
                               const float* ab, const float* gamma, const float* const* gamma_tab,
                               const float* res, const ws_groups_geom* geo, float* dx,
                               void* stream) {
  int rc = geom_check(geo, "ws_gn_bwd_apply");
  if (rc != WS_OK) return rc;
  WS_REQUIRE(x && dxn && stats && ab && dx && (gamma || gamma_tab), "ws_gn_bwd_apply: null pointer");
  if (geom_vec4(geo))
    hipLaunchKernelGGL((gn_bwd_apply_kernel<true>), dim3(geo->ngroups), dim3(256), 0, (hipStream_t)stream, x,
                       dxn, stats, ab, gamma, gamma_tab, res, *geo, dx);
  else
    hipLaunchKernelGGL((gn_bwd_apply_kernel<false>), dim3(geo->ngroups), dim3(256), 0, (hipStream_t)stream, x,
                       dxn, stats, ab, gamma, gamma_tab, res, *geo, dx);
  return ws_check_launch("ws_gn_bwd_apply");
}

// dgamma[col] = sum dxn*xhat, dbeta[col] = sum dxn over every row of every group of a band.
// grid (nbands, nsplit); 256 threads = 128 columns x 2 row lanes.
__global__ __launch_bounds__(256) void gn_param_grad_kernel(const float* __restrict__ x,
                                                            const float* __restrict__ dxn,
                                                            const float* __restrict__ stats,
                                                            const ws_groups_geom geo, int nsplit,
                                                            float* __restrict__ slab) {
  __shared__ float sh[2][128];
  const int band = blockIdx.x, split = blockIdx.y;
  const int col = threadIdx.x & 127, rl = threadIdx.x >> 7;
  const int per_band = geo.ngroups / geo.nbands;
  float sg = 0.f, sb = 0.f;
  for (int j = split; j < per_band; j += nsplit) {
    const int g = j * geo.nbands + band;
    const GroupView v = group_view(geo, g);
    if (col >= v.W) continue;
    const float mean = stats[2 * (long long)g], rstd = stats[2 * (long long)g + 1];
    for (int row = rl; row < geo.L; row += 2) {
      const long long o = v.base + (long long)row * geo.rs + col;
      const float dv = dxn[o];
      sg += dv * (x[o] - mean) * rstd;
      sb += dv;
    }
  }
  if (rl == 1) {
    sh[0][col] = sg;
    sh[1][col] = sb;
  }
  __syncthreads();
  if (rl == 0 && col < geo.W) {
    float* out = slab + ((long long)(split * geo.nbands + band) * 2) * geo.W;
    out[col] = sg + sh[0][col];
    out[geo.W + col] = sb + sh[1][col];
  }
}

// Same sums for the single-band, 128-wide geometries (ResRNN norms): 32 threads x float4 cover a row,
// 8 row lanes per workgroup, groups strided over the splits.
__global__ __launch_bounds__(256) void gn_param_grad128_kernel(const float* __restrict__ x,
                                                               const float* __restrict__ dxn,
                                                               const float* __restrict__ stats,
                                                               const ws_groups_geom geo, int nsplit,
                                                               float* __restrict__ slab) {
  __shared__ f32x4 sh[2][8][32];
  const int split = blockIdx.x;
  const int c4 = threadIdx.x & 31, rl = threadIdx.x >> 5;
  f32x4 sg = {0.f, 0.f, 0.f, 0.f}, sb = {0.f, 0.f, 0.f, 0.f};
  for (int g = split; g < geo.ngroups; g += nsplit) {
    const GroupView v = group_view(geo, g);
    const float mean = stats[2 * (long long)g], rstd = stats[2 * (long long)g + 1];
    for (int row = rl; row < geo.L; row += 8) {
      const long long o = v.base + (long long)row * geo.rs + 4 * c4;
      const f32x4 dv = *reinterpret_cast<const f32x4*>(dxn + o);
      sg += dv * ((*reinterpret_cast<const f32x4*>(x + o) - mean) * rstd);
      sb += dv;
    }
  }
  sh[0][rl][c4] = sg;
  sh[1][rl][c4] = sb;
  __syncthreads();
  if (rl < 2) {  // rl 0 -> dgamma, rl 1 -> dbeta
    f32x4 t = sh[rl][0][c4];
#pragma unroll
    for (int r = 1; r < 8; ++r) t += sh[rl][r][c4];
    *reinterpret_cast<f32x4*>(slab + ((long long)split * 2 + rl) * 128 + 4 * c4) = t;
  }
}

extern "C" int ws_gn_param_grad(const float* x, const float* dxn, const float* stats,
                                const ws_groups_geom* geo, int nsplit, float* slab, void* stream) {
  int rc = geom_check(geo, "ws_gn_param_grad");
  if (rc != WS_OK) return rc;
  WS_REQUIRE(x && dxn && stats && slab && nsplit > 0, "ws_gn_param_grad: bad args");
  WS_REQUIRE(geo->ngroups % geo->nbands == 0, "ws_gn_param_grad: ngroups %% nbands != 0");
  WS_REQUIRE(geo->W <= 128, "ws_gn_param_grad: W > 128");
  if (geo->nbands == 1 && geo->W == 128 && geom_vec4(geo))
    hipLaunchKernelGGL(gn_param_grad128_kernel, dim3(nsplit), dim3(256), 0, (hipStream_t)stream, x, dxn, stats,
                       *geo, nsplit, slab);
  else
    hipLaunchKernelGGL(gn_param_grad_kernel, dim3(geo->nbands, nsplit), dim3(256), 0,
                       (hipStream_t)stream, x, dxn, stats, *geo, nsplit, slab);
  return ws_check_launch("ws_gn_param_grad");
}

// ---------------------------------------------------------------------------------------------
// Fused GroupNorm backward for SMALL groups (the band view of ResRNN: a group = the K = 32 rows x 128 floats of one
// (row, frame), 16 KB, rows Tf*N floats apart).  The three-kernel form above launches one 256-thread workgroup per
// group, i.e. 16 032 workgroups that each touch 2 x 16 KB -- latency-bound (1.0 ms for the reduction alone at
// R = 32) -- and reads x / dxn three times.  Here ONE WAVE owns a group: its x and dxn live in registers (16 float4
// each per lane), the two group means are wave reductions, dx is written from the registers, and the per-column
// dgamma / dbeta sums accumulate in the lane that owns the column across all groups of the wave: x, dxn and the
// residual cross HBM exactly once.  Deterministic (fixed group -> wave assignment, fixed reduction order).
//   lane = (row lane rl = lane >> 5, column quad c4 = lane & 31); rows rl, rl + 2, ...; L <= 32 and even, W = 128.
// ---------------------------------------------------------------------------------------------
#define GNF_ROWS 16  // rows per lane (L / 2 <= 16)
__global__ __launch_bounds__(256) void gn_bwd_fused_kernel(const float* __restrict__ x, const float* __restrict__ dxn,
                                                           const float* __restrict__ dxn2,   // optional second addend
                                                           const float* __restrict__ stats,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ res, const ws_groups_geom geo,
                                                           float* __restrict__ dx, float* __restrict__ pslab,
                                                           float* __restrict__ pout, unsigned* counter) {
  __shared__ f32x4 sh[2][4][32];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c4 = lane & 31, rl = lane >> 5;
  const int nrow = geo.L >> 1;  // rows per lane
  const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + 4 * c4);
  const float inv_n = 1.f / (float)(geo.L * 128);
  f32x4 sg = {0.f, 0.f, 0.f, 0.f}, sb = {0.f, 0.f, 0.f, 0.f};
  const int nwave = gridDim.x * 4;
  for (int g = blockIdx.x * 4 + w; g < geo.ngroups; g += nwave) {
    const long long base = (long long)(g / geo.gdiv) * geo.gs1 + (long long)(g % geo.gdiv) * geo.gs2 + 4 * c4;
    const float mean = stats[2 * (long long)g], rstd = stats[2 * (long long)g + 1];
    f32x4 xh[GNF_ROWS], dg[GNF_ROWS];
#pragma unroll
    for (int j = 0; j < GNF_ROWS; ++j)
      if (j < nrow) {
        const long long o = base + (long long)(rl + 2 * j) * geo.rs;
        xh[j] = *reinterpret_cast<const f32x4*>(x + o);
        dg[j] = *reinterpret_cast<const f32x4*>(dxn + o);
      }
    if (dxn2) {   // (uniform) d(xn) of the two LSTM directions, written apart by the BPTT (ws_lstm_args.dxn)
#pragma unroll
      for (int j = 0; j < GNF_ROWS; ++j)
        if (j < nrow) dg[j] += *reinterpret_cast<const f32x4*>(dxn2 + base + (long long)(rl + 2 * j) * geo.rs);
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < GNF_ROWS; ++j)
      if (j < nrow) {
        xh[j] = (xh[j] - mean) * rstd;
        sg += dg[j] * xh[j];
        sb += dg[j];
        dg[j] = dg[j] * gm;
        s1 += (dg[j][0] + dg[j][1]) + (dg[j][2] + dg[j][3]);
        s2 += (dg[j][0] * xh[j][0] + dg[j][1] * xh[j][1]) + (dg[j][2] * xh[j][2] + dg[j][3] * xh[j][3]);
      }
    const float a0 = ws_wave_sum(s1) * inv_n, a1 = ws_wave_sum(s2) * inv_n;
#pragma unroll
    for (int j = 0; j < GNF_ROWS; ++j)
      if (j < nrow) {
        const long long o = base + (long long)(rl + 2 * j) * geo.rs;
        f32x4 r = (dg[j] - a0 - xh[j] * a1) * rstd;
        if (res) r += *reinterpret_cast<const f32x4*>(res + o);
        *reinterpret_cast<f32x4*>(dx + o) = r;
      }
  }
  // column sums: the two row lanes of a wave, then the four waves, in fixed order
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    sg[q] += __shfl_xor(sg[q], 32, 64);
    sb[q] += __shfl_xor(sb[q], 32, 64);
  }
  if (rl == 0) {
    sh[0][w][c4] = sg;
    sh[1][w][c4] = sb;
  }
  __syncthreads();
  if (threadIdx.x < 64) {  // lanes 0..31 -> dgamma, 32..63 -> dbeta
    f32x4 t = sh[rl][0][c4];
#pragma unroll
    for (int r = 1; r < 4; ++r) t += sh[rl][r][c4];
    float* o = pslab + ((long long)blockIdx.x * 2 + rl) * 128 + 4 * c4;
    if (pout) {  // agent-scope stores: read by another workgroup (common.h ws_last_block)
#pragma unroll
      for (int q = 0; q < 4; ++q) ws_st_agent(o + q, t[q]);
    } else {
      *reinterpret_cast<f32x4*>(o) = t;
    }
  }
  // (optional) the workgroups of the launch add their shares up themselves, two levels, fixed order: no reduction launch
  if (pout) ws_tree_sum256(pslab, gridDim.x, pout, counter);   // (uniform: every workgroup takes the same branch)
}

extern "C" int ws_gn_bwd_fused(const float* x, const float* dxn, const float* stats, const float* gamma,
                               const float* res, const ws_groups_geom* geo, int nwg, float* dx, float* pslab,
                               float* pout, unsigned* counter, void* stream) {
  return ws_gn_bwd_fused2(x, dxn, nullptr, stats, gamma, res, geo, nwg, dx, pslab, pout, counter, stream);
}

extern "C" int ws_gn_bwd_fused2(const float* x, const float* dxn, const float* dxn2, const float* stats, const float* gamma,
                                const float* res, const ws_groups_geom* geo, int nwg, float* dx, float* pslab,
                                float* pout, unsigned* counter, void* stream) {
  int rc = geom_check(geo, "ws_gn_bwd_fused");
  if (rc != WS_OK) return rc;
  WS_REQUIRE(x && dxn && stats && gamma && dx && pslab && nwg > 0, "ws_gn_bwd_fused: null pointer / nwg");
  WS_REQUIRE(geom_vec4(geo) && geo->nbands == 1 && geo->W == 128 && geo->L >= 2 && geo->L <= 2 * GNF_ROWS &&
                 geo->L % 2 == 0,
             "ws_gn_bwd_fused: built for single-band groups of an even number (<= %d) of 128-float rows (L=%d, W=%d)",
             2 * GNF_ROWS, geo->L, geo->W);
  WS_REQUIRE(!pout || counter, "ws_gn_bwd_fused: pout needs a counter word");
  hipLaunchKernelGGL(gn_bwd_fused_kernel, dim3(nwg), dim3(256), 0, (hipStream_t)stream, x, dxn, dxn2, stats, gamma, res,
                     *geo, dx, pslab, pout, counter);
  return ws_check_launch("ws_gn_bwd_fused");
}

// ---------------------------------------------------------------------------------------------
// Row LayerNorm for SHORT rows (width <= 256 floats, % 4 == 0): y = gamma * (x - mean_row) * rstd_row + beta.
// TF-GridNet normalises every (batch, time, frequency) position over its 128 channels before each BLSTM
// (gridnet_block.py:139-160 `intra_norm` / `inter_norm` = nn.LayerNorm(emb_dim)): 780 k rows of 512 bytes at the recipe's
// shape.  The generic group kernels above give every group a 256-thread workgroup and make three to five passes
// (statistics, apply; backward: row sums, channel sums, apply); here LPR = 8 .. 64 lanes own a row (one float4 each),
// the row statistics are lane-group shuffles, and forward and backward are ONE pass each.  The backward leaves the
// per-workgroup partial sums of d(gamma) / d(beta) in `slab` ([grid][2][W]: d(beta), d(gamma)) for ws_reduce_slabs --
// a fixed grid, so the result is reproducible.  Same two-pass mean / variance as group_stats_kernel.
// ---------------------------------------------------------------------------------------------
#define ROWLN_GRID_MAX 2048

template <int LPR>
__device__ __forceinline__ float rowln_sum(float v) {
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <int LPR>
__global__ __launch_bounds__(256) void rowln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, long long M, int W, float eps,
                                                        float* __restrict__ y, float* __restrict__ stats) {
  constexpr int RPB = 256 / LPR;  // rows per workgroup and iteration
  const int lr = threadIdx.x % LPR, slot = threadIdx.x / LPR, c = 4 * lr;
  const bool act = c < W;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  const f32x4 g = act ? *reinterpret_cast<const f32x4*>(gamma + c) : zero;
  const f32x4 b = act ? *reinterpret_cast<const f32x4*>(beta + c) : zero;
  const float inv = 1.f / (float)W;
  for (long long row = (long long)blockIdx.x * RPB + slot; row < M; row += (long long)gridDim.x * RPB) {
    const f32x4 v = act ? *reinterpret_cast<const f32x4*>(x + row * W + c) : zero;
    const float mean = rowln_sum<LPR>(v[0] + v[1] + v[2] + v[3]) * inv;
    const f32x4 dv = act ? v - mean : zero;
    const float var = rowln_sum<LPR>(dv[0] * dv[0] + dv[1] * dv[1] + dv[2] * dv[2] + dv[3] * dv[3]) * inv;
    const float rstd = 1.f / sqrtf(var + eps);
    if (act) *reinterpret_cast<f32x4*>(y + row * W + c) = dv * rstd * g + b;
    if (lr == 0) {
      stats[2 * row] = mean;
      stats[2 * row + 1] = rstd;
    }
  }
}

template <int LPR>
__global__ __launch_bounds__(256) void rowln_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                        const float* __restrict__ stats, const float* __restrict__ gamma,
                                                        const float* __restrict__ res, long long M, int W,
                                                        float* __restrict__ dx, float* __restrict__ slab) {
  constexpr int RPB = 256 / LPR;
  __shared__ f32x4 red[2][256];
  const int lr = threadIdx.x % LPR, slot = threadIdx.x / LPR, c = 4 * lr;
  const bool act = c < W;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  const f32x4 g = act ? *reinterpret_cast<const f32x4*>(gamma + c) : zero;
  const float inv = 1.f / (float)W;
  f32x4 adb = zero, adg = zero;
  for (long long row = (long long)blockIdx.x * RPB + slot; row < M; row += (long long)gridDim.x * RPB) {
    const f32x4 v = act ? *reinterpret_cast<const f32x4*>(x + row * W + c) : zero;
    const f32x4 d = act ? *reinterpret_cast<const f32x4*>(dy + row * W + c) : zero;
    const float mean = stats[2 * row], rstd = stats[2 * row + 1];
    const f32x4 xh = act ? (v - mean) * rstd : zero;
    const f32x4 gd = g * d;
    const float s1 = rowln_sum<LPR>(gd[0] + gd[1] + gd[2] + gd[3]) * inv;
    const float s2 = rowln_sum<LPR>(gd[0] * xh[0] + gd[1] * xh[1] + gd[2] * xh[2] + gd[3] * xh[3]) * inv;
    if (act) {
      f32x4 o = (gd - s1 - xh * s2) * rstd;
      if (res) o += *reinterpret_cast<const f32x4*>(res + row * W + c);
      *reinterpret_cast<f32x4*>(dx + row * W + c) = o;
    }
    adb += d;
    adg += d * xh;
  }
  red[0][threadIdx.x] = adb;
  red[1][threadIdx.x] = adg;
  __syncthreads();
  if (slot == 0 && act) {
#pragma unroll 4
    for (int s = 1; s < RPB; ++s) {
      adb += red[0][s * LPR + lr];
      adg += red[1][s * LPR + lr];
    }
    float* o = slab + (long long)blockIdx.x * 2 * W;
    *reinterpret_cast<f32x4*>(o + c) = adb;
    *reinterpret_cast<f32x4*>(o + W + c) = adg;
  }
}

static int rowln_lpr(int W) { return W > 128 ? 64 : W > 64 ? 32 : W > 32 ? 16 : 8; }
static int rowln_grid(long long M, int lpr) {
  const long long need = (M + 256 / lpr - 1) / (256 / lpr);
  return (int)(need < ROWLN_GRID_MAX ? need : ROWLN_GRID_MAX);
}

extern "C" int ws_rowln_grid(long long M, int W) { return M > 0 && W > 0 ? rowln_grid(M, rowln_lpr(W)) : 0; }

extern "C" int ws_rowln_fwd(const float* x, const float* gamma, const float* beta, long long M, int W, float eps,
                            float* y, float* stats, void* stream) {
  WS_REQUIRE(x && gamma && beta && y && stats, "ws_rowln_fwd: null pointer");
  WS_REQUIRE(M > 0 && W > 0 && W <= 256 && W % 4 == 0, "ws_rowln_fwd: rows of 4 .. 256 floats, width %% 4 == 0 (got %d)", W);
  const int lpr = rowln_lpr(W), grid = rowln_grid(M, lpr);
  hipStream_t s = (hipStream_t)stream;
  switch (lpr) {
    case 64: hipLaunchKernelGGL(rowln_fwd_kernel<64>, dim3(grid), dim3(256), 0, s, x, gamma, beta, M, W, eps, y, stats); break;
    case 32: hipLaunchKernelGGL(rowln_fwd_kernel<32>, dim3(grid), dim3(256), 0, s, x, gamma, beta, M, W, eps, y, stats); break;
    case 16: hipLaunchKernelGGL(rowln_fwd_kernel<16>, dim3(grid), dim3(256), 0, s, x, gamma, beta, M, W, eps, y, stats); break;
    default: hipLaunchKernelGGL(rowln_fwd_kernel<8>, dim3(grid), dim3(256), 0, s, x, gamma, beta, M, W, eps, y, stats);
  }
  return ws_check_launch("ws_rowln_fwd");
}

extern "C" int ws_rowln_bwd(const float* x, const float* dy, const float* stats, const float* gamma, const float* res,
                            long long M, int W, float* dx, float* slab, void* stream) {
  WS_REQUIRE(x && dy && stats && gamma && dx && slab, "ws_rowln_bwd: null pointer");
  WS_REQUIRE(M > 0 && W > 0 && W <= 256 && W % 4 == 0, "ws_rowln_bwd: rows of 4 .. 256 floats, width %% 4 == 0 (got %d)", W);
  const int lpr = rowln_lpr(W), grid = rowln_grid(M, lpr);
  hipStream_t s = (hipStream_t)stream;
  switch (lpr) {
    case 64: hipLaunchKernelGGL(rowln_bwd_kernel<64>, dim3(grid), dim3(256), 0, s, x, dy, stats, gamma, res, M, W, dx, slab); break;
    case 32: hipLaunchKernelGGL(rowln_bwd_kernel<32>, dim3(grid), dim3(256), 0, s, x, dy, stats, gamma, res, M, W, dx, slab); break;
    case 16: hipLaunchKernelGGL(rowln_bwd_kernel<16>, dim3(grid), dim3(256), 0, s, x, dy, stats, gamma, res, M, W, dx, slab); break;
    default: hipLaunchKernelGGL(rowln_bwd_kernel<8>, dim3(grid), dim3(256), 0, s, x, dy, stats, gamma, res, M, W, dx, slab);
  }
  return ws_check_launch("ws_rowln_bwd");
}
