// fp32 MFMA GEMM family for gfx950 (v_mfma_f32_32x32x2_f32: exact f32, 64 FLOP/clk/SIMD).
//
//   ws_gemm_nt : C[M][N] = epi(pro(A)[M][K] * W[N][K]^T)      data path (fwd + data-gradients)
//   ws_gemm_tn : dW[Nn][Kk] = G[M][Nn]^T * pro(A)[M][Kk]      weight gradients (split over M)
//
// Block tile 128x128, 4 waves (2x2), each wave 64x64 = 2x2 MFMA tiles of 32x32 -> 64 acc VGPRs.
// Operands go global -> registers -> LDS (k-major, +1 pad: conflict-free ds_read_b32 by the
// MFMA's lane map i = lane&31, k = lane>>5), double-buffered in LDS, one barrier per k-tile,
// next tile's global loads in flight under the current tile's 64 MFMAs (4096 cycles).
// GroupNorm is applied on load, bias/tanh/tanh'/residual in the accumulator epilogue.
#include "common.h"

#define NT_BK 32
#define NT_LD 129  // 128 + 1

// D-fragment row of register `reg` for the 32x32 MFMA (lane>>5 = half)
__device__ __forceinline__ int frag_row32(int reg, int half) {
  return (reg & 3) + 8 * (reg >> 2) + 4 * half;
}

template <bool VEC_A, bool VEC_W>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const ws_gemm_nt_args p) {
  __shared__ float lds[2 * 2 * NT_BK * NT_LD];  // [buf][A|W][k][m]  = 66 KB
  const float* A = p.A;
  const float* W = p.W;
  const float* bias = p.bias;
  const float* gamma = p.gamma;
  const float* beta = p.beta;
  float* C = p.C;
  const float* R = p.R;
  const float* T = p.T;
  int K = p.K, N = p.N, ldw = p.ldw;
  long long st_base = p.st_base;
  if (p.groups) {
    const ws_group_nt g = p.groups[blockIdx.z];
    A += g.a_off;
    W = g.W;
    bias = g.bias;
    gamma = g.gamma;
    beta = g.beta;
    C += g.c_off;
    if (R) R += g.c_off;
    if (T) T += g.c_off;
    st_base = g.st_base;
    K = g.K;
    N = g.N;
    ldw = g.ldw;
  }
  const int M = p.M;
  const int m_blk = blockIdx.x * 128, n_blk = blockIdx.y * 128;
  if (n_blk >= N) return;  // ragged N across groups
  const int tid = threadIdx.x;
  const int lrow = tid >> 3, lk = (tid & 7) * 4;
  const bool has_norm = p.stats != nullptr;

  long long aoff[4];
  long long woff[4];
  bool vm[4], vn[4];
  float mean[4], rstd[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m_blk + lrow + 32 * i;
    vm[i] = m < M;
    const int mm = vm[i] ? m : 0;
    aoff[i] = ws_row_off(mm, p.a_div, p.a_s1, p.a_s2);
    mean[i] = 0.f;
    rstd[i] = 1.f;
    if (has_norm) {
      const long long s = (long long)(mm / p.st_div1) * p.st_m1 +
                          (long long)(mm % p.st_div2) * p.st_m2 + st_base;
      mean[i] = p.stats[2 * s];
      rstd[i] = p.stats[2 * s + 1];
    }
    const int n = n_blk + lrow + 32 * i;
    vn[i] = n < N;
    woff[i] = (long long)(vn[i] ? n : 0) * ldw;
  }

  f32x4 ra[4], rw[4];
  auto load_tile = [&](int kt) {
    const int k = kt * NT_BK + lk;
    float gm[4] = {1.f, 1.f, 1.f, 1.f}, bt[4] = {0.f, 0.f, 0.f, 0.f};
    if (has_norm) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (k + j < K) {
          gm[j] = gamma[k + j];
          bt[j] = beta[k + j];
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (vm[i]) {
        if (VEC_A) {
          if (k < K) v = *reinterpret_cast<const f32x4*>(A + aoff[i] + k);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (k + j < K) v[j] = A[aoff[i] + k + j];
        }
        if (has_norm) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            v[j] = (k + j < K) ? (v[j] - mean[i]) * rstd[i] * gm[j] + bt[j] : 0.f;
        }
      }
      ra[i] = v;
      f32x4 w = {0.f, 0.f, 0.f, 0.f};
      if (vn[i]) {
        if (VEC_W) {
          if (k < K) w = *reinterpret_cast<const f32x4*>(W + woff[i] + k);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (k + j < K) w[j] = W[woff[i] + k + j];
        }
      }
      rw[i] = w;
    }
  };
  auto store_tile = [&](int buf) {
    float* As = lds + buf * (2 * NT_BK * NT_LD);
    float* Ws = As + NT_BK * NT_LD;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        As[(lk + j) * NT_LD + lrow + 32 * i] = ra[i][j];
        Ws[(lk + j) * NT_LD + lrow + 32 * i] = rw[i][j];
      }
  };

  const int wave = tid >> 6, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;
  f32x16 acc00, acc01, acc10, acc11;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc00[r] = acc01[r] = acc10[r] = acc11[r] = 0.f;

  const int nk = (K + NT_BK - 1) / NT_BK;
  load_tile(0);
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    store_tile(buf);
    __syncthreads();
    if (kt + 1 < nk) load_tile(kt + 1);
    const float* As = lds + buf * (2 * NT_BK * NT_LD);
    const float* Ws = As + NT_BK * NT_LD;
#pragma unroll
    for (int kk = 0; kk < NT_BK; kk += 2) {
      const float a0 = As[(kk + half) * NT_LD + wm * 64 + l31];
      const float a1 = As[(kk + half) * NT_LD + wm * 64 + 32 + l31];
      const float b0 = Ws[(kk + half) * NT_LD + wn * 64 + l31];
      const float b1 = Ws[(kk + half) * NT_LD + wn * 64 + 32 + l31];
      acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc00, 0, 0, 0);
      acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc01, 0, 0, 0);
      acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc10, 0, 0, 0);
      acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc11, 0, 0, 0);
    }
  }

  // epilogue: lane holds column n = ... + l31, rows by register
  auto epilogue = [&](const f32x16& acc, int tm, int tn) {
    const int n = n_blk + wn * 64 + tn * 32 + l31;
    if (n >= N) return;
    const float bv = bias ? bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m_blk + wm * 64 + tm * 32 + frag_row32(r, half);
      if (m >= M) continue;
      const long long off = ws_row_off(m, p.c_div, p.c_s1, p.c_s2) + n;
      float v = acc[r] + bv;
      if (p.act == 1) v = tanhf(v);
      if (p.act == 2) v = fmaxf(v, 0.f);
      if (T) {  // derivative of the activation from its saved OUTPUT: tanh' (default) or ReLU' (act 4)
        const float t = T[off];
        v *= p.act == 4 ? (t > 0.f ? 1.f : 0.f) : (1.f - t * t);
      }
      if (R) v += R[off];
      C[off] = v;
    }
  };
  epilogue(acc00, 0, 0);
  epilogue(acc01, 0, 1);
  epilogue(acc10, 1, 0);
  epilogue(acc11, 1, 1);
}

extern "C" int ws_gemm_nt(const ws_gemm_nt_args* a, void* stream) {
  WS_REQUIRE(a && a->A && a->C, "ws_gemm_nt: null A/C");
  WS_REQUIRE(a->groups || a->W, "ws_gemm_nt: null W");
  WS_REQUIRE(a->M > 0, "ws_gemm_nt: M=%d", a->M);
  WS_REQUIRE(a->a_div > 0 && a->c_div > 0, "ws_gemm_nt: bad row divisors");
  WS_REQUIRE(!a->stats || (a->st_div1 > 0 && a->st_div2 > 0), "ws_gemm_nt: bad stat divisors");
  const int ng = a->groups ? a->ngroups : 1;
  const int maxn = a->groups ? a->max_n : a->N;
  WS_REQUIRE(ng > 0 && maxn > 0, "ws_gemm_nt: ngroups=%d max_n=%d", ng, maxn);
  WS_REQUIRE(a->groups || (a->N > 0 && a->K > 0 && a->ldw >= ((a->vec & 8) ? a->N : a->K)), "ws_gemm_nt: bad N/K/ldw");
  // vec bit 3 (round 6): W is stored transposed, W'[n][k] = W[k * ldw + n] -- the split-bf16 kernel only, 16-byte rows of W
  WS_REQUIRE(!(a->vec & 8) || ((a->vec & 7) == 7 && !a->conv.on && !a->stats && (a->groups || (a->N % 4 == 0 && a->ldw % 4 == 0))),
             "ws_gemm_nt: transposed W (vec bit 3) needs vec 15, no conv view, no norm-on-load, N %% 4 == 0 and ldw %% 4 == 0");
  if (a->conv.on) {
    const ws_conv_view& c = a->conv;
    WS_REQUIRE((a->vec & 7) == 7 && !a->groups && !a->stats, "ws_gemm_nt: the implicit patch matrix needs the split-bf16 "
               "kernel (vec 7), no groups, no norm-on-load");
    WS_REQUIRE((c.mode == 0 || c.mode == 1) && c.H > 0 && c.W > 0 && c.C > 0 && c.C % 4 == 0 && c.Ho > 0 && c.Wo > 0 &&
                   c.k >= 1 && c.sh >= 1 && c.sw >= 1 && c.p >= 0 && a->K == c.k * c.k * c.C && a->M % (c.Ho * c.Wo) == 0,
               "ws_gemm_nt: bad conv view (C %% 4, K == k*k*C, M %% (Ho*Wo))");
    WS_REQUIRE(c.mode == 0 || (c.sh <= 2 && c.sw <= 2), "ws_gemm_nt: transposed view: strides 1 or 2");
    WS_REQUIRE(c.ldp == 0 || (c.ldp >= c.C && c.ldp % 4 == 0), "ws_gemm_nt: pixel stride ldp >= C, %% 4 (got %d)", c.ldp);
    WS_REQUIRE((long long)c.H * c.W * (c.ldp > 0 ? c.ldp : c.C) < (1LL << 31),
               "ws_gemm_nt: one image must stay below 2^31 elements");
  }
  dim3 grid((a->M + 127) / 128, (maxn + 127) / 128, ng), block(256);
  hipStream_t s = (hipStream_t)stream;
  ws_prof_begin(WS_PROF_GEMM_NT, s);
  const bool va = a->vec & 1, vw = a->vec & 2;
  if ((a->vec & 4) && va && vw)
    ws_launch_gemm_nt_bf16(a, grid, s);
  else if (va && vw)
    hipLaunchKernelGGL((gemm_nt_kernel<true, true>), grid, block, 0, s, *a);
  else if (va)
    hipLaunchKernelGGL((gemm_nt_kernel<true, false>), grid, block, 0, s, *a);
  else if (vw)
    hipLaunchKernelGGL((gemm_nt_kernel<false, true>), grid, block, 0, s, *a);
  else
    hipLaunchKernelGGL((gemm_nt_kernel<false, false>), grid, block, 0, s, *a);
  ws_prof_end(WS_PROF_GEMM_NT, s);
  return ws_check_launch("ws_gemm_nt");
}

// -------------------------------------------------------------------------------------------
// TN: weight gradients.  LDS tiles are [32 rows m][128 cols] row-major (+4 pad keeps 16-B
// alignment for ds_write_b128); the MFMA reads column slices, conflict-free.
// -------------------------------------------------------------------------------------------
#define TN_BM 32
#define TN_LD 132

template <bool VEC_A>
__global__ __launch_bounds__(256) void gemm_tn_kernel(const ws_gemm_tn_args p) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 2 * TN_BM * TN_LD];  // 67.6 KB
  const float* G = p.G;
  const float* A = p.A;
  const float* gamma = p.gamma;
  const float* beta = p.beta;
  int Nn = p.Nn, Kk = p.Kk;
  long long st_base = p.st_base, out_off = p.out_off, bout_off = p.bout_off;
  if (p.groups) {
    const ws_group_tn g = p.groups[blockIdx.z];
    G += g.g_off;
    A += g.a_off;
    gamma = g.gamma;
    beta = g.beta;
    st_base = g.st_base;
    out_off = g.out_off;
    bout_off = g.bout_off;
    Nn = g.Nn;
    Kk = g.Kk;
  }
  const int tiles_k = (Kk + 127) / 128, tiles_n = (Nn + 127) / 128;
  if ((int)blockIdx.x >= tiles_k * tiles_n) return;
  const int n_blk = (blockIdx.x / tiles_k) * 128, k_blk = (blockIdx.x % tiles_k) * 128;
  const int split = blockIdx.y;
  const int m_begin = split * p.rows_per_split;
  const int m_end = min(p.M, m_begin + p.rows_per_split);
  const int tid = threadIdx.x;
  const int lrow = tid >> 5, lc = (tid & 31) * 4;
  const bool has_norm = p.stats != nullptr;
  const bool do_bias = p.bslab != nullptr && k_blk == 0;

  // per-thread column constants
  float gm[4] = {1.f, 1.f, 1.f, 1.f}, bt[4] = {0.f, 0.f, 0.f, 0.f};
  if (has_norm) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (k_blk + lc + j < Kk) {
        gm[j] = gamma[k_blk + lc + j];
        bt[j] = beta[k_blk + lc + j];
      }
  }

  f32x4 rg[4], ra[4];
  auto load_tile = [&](int m0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + lrow + 8 * i;
      f32x4 g = {0.f, 0.f, 0.f, 0.f}, a = {0.f, 0.f, 0.f, 0.f};
      if (m < m_end) {
        const long long goff = ws_row_off(m, p.g_div, p.g_s1, p.g_s2);
        const int n = n_blk + lc;
        if (n < Nn) g = *reinterpret_cast<const f32x4*>(G + goff + n);  // Nn % 4 == 0 (checked)
        int ma = m;
        bool ok = true;
        if (p.shift_rows != 0) {
          const int t = (m / p.seq_div) % p.seq_len;
          const int t2 = t + (p.shift_rows > 0 ? 1 : -1);
          ok = (t2 >= 0) && (t2 < p.seq_len);
          ma = m + p.shift_rows;
        }
        if (ok) {
          const long long ao = ws_row_off(ma, p.a_div, p.a_s1, p.a_s2);
          const int k = k_blk + lc;
          if (VEC_A) {
            if (k < Kk) a = *reinterpret_cast<const f32x4*>(A + ao + k);
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (k + j < Kk) a[j] = A[ao + k + j];
          }
          if (has_norm) {
            const long long s = (long long)(m / p.st_div1) * p.st_m1 +
                                (long long)(m % p.st_div2) * p.st_m2 + st_base;
            const float mu = p.stats[2 * s], rs = p.stats[2 * s + 1];
#pragma unroll
            for (int j = 0; j < 4; ++j)
              a[j] = (k + j < Kk) ? (a[j] - mu) * rs * gm[j] + bt[j] : 0.f;
          }
        }
      }
      rg[i] = g;
      ra[i] = a;
    }
  };
  auto store_tile = [&](int buf) {
    float* Gs = lds + buf * (2 * TN_BM * TN_LD);
    float* As = Gs + TN_BM * TN_LD;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<f32x4*>(Gs + (lrow + 8 * i) * TN_LD + lc) = rg[i];
      *reinterpret_cast<f32x4*>(As + (lrow + 8 * i) * TN_LD + lc) = ra[i];
    }
  };

  const int wave = tid >> 6, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;
  f32x16 acc00, acc01, acc10, acc11;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc00[r] = acc01[r] = acc10[r] = acc11[r] = 0.f;
  float bsum = 0.f;

  int it = 0;
  if (m_begin < m_end) load_tile(m_begin);
  for (int m0 = m_begin; m0 < m_end; m0 += TN_BM, ++it) {
    const int buf = it & 1;
    store_tile(buf);
    __syncthreads();
    if (m0 + TN_BM < m_end) load_tile(m0 + TN_BM);
    const float* Gs = lds + buf * (2 * TN_BM * TN_LD);
    const float* As = Gs + TN_BM * TN_LD;
    if (do_bias && tid < 128) {
#pragma unroll 8
      for (int r = 0; r < TN_BM; ++r) bsum += Gs[r * TN_LD + tid];
    }
#pragma unroll
    for (int kk = 0; kk < TN_BM; kk += 2) {
      const float a0 = Gs[(kk + half) * TN_LD + wm * 64 + l31];
      const float a1 = Gs[(kk + half) * TN_LD + wm * 64 + 32 + l31];
      const float b0 = As[(kk + half) * TN_LD + wn * 64 + l31];
      const float b1 = As[(kk + half) * TN_LD + wn * 64 + 32 + l31];
      acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc00, 0, 0, 0);
      acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc01, 0, 0, 0);
      acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc10, 0, 0, 0);
      acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc11, 0, 0, 0);
    }
  }

  float* out = p.slab + (long long)split * p.slab_stride + out_off;
  auto write = [&](const f32x16& acc, int tm, int tn) {
    const int k = k_blk + wn * 64 + tn * 32 + l31;
    if (k >= Kk) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = n_blk + wm * 64 + tm * 32 + frag_row32(r, half);
      if (n < Nn) out[(long long)n * Kk + k] = acc[r];
    }
  };
  write(acc00, 0, 0);
  write(acc01, 0, 1);
  write(acc10, 1, 0);
  write(acc11, 1, 1);
  if (do_bias && tid < 128 && n_blk + tid < Nn)
    p.bslab[(long long)split * p.bslab_stride + bout_off + n_blk + tid] = bsum;
}

extern "C" int ws_gemm_tn(const ws_gemm_tn_args* a, void* stream) {
  WS_REQUIRE(a && a->G && a->A && a->slab, "ws_gemm_tn: null G/A/slab");
  WS_REQUIRE(a->M > 0 && a->nsplit > 0 && a->rows_per_split > 0, "ws_gemm_tn: bad M/split");
  WS_REQUIRE((long long)a->nsplit * a->rows_per_split >= a->M, "ws_gemm_tn: splits do not cover M");
  WS_REQUIRE(a->g_div > 0 && a->a_div > 0, "ws_gemm_tn: bad row divisors");
  WS_REQUIRE(!a->stats || (a->st_div1 > 0 && a->st_div2 > 0), "ws_gemm_tn: bad stat divisors");
  WS_REQUIRE(a->shift_rows == 0 || (a->seq_div > 0 && a->seq_len > 0), "ws_gemm_tn: bad shift");
  const int ng = a->groups ? a->ngroups : 1;
  const int maxn = a->groups ? a->max_n : a->Nn, maxk = a->groups ? a->max_k : a->Kk;
  WS_REQUIRE(ng > 0 && maxn > 0 && maxk > 0, "ws_gemm_tn: bad group dims");
  WS_REQUIRE(a->groups || (a->Nn % 4 == 0), "ws_gemm_tn: Nn must be a multiple of 4");
  if (a->conv.on) {
    const ws_conv_view& c = a->conv;
    WS_REQUIRE((a->vec & 4) && !a->groups && !a->stats && a->shift_rows == 0, "ws_gemm_tn: the implicit patch matrix "
               "needs the split-bf16 kernel, no groups, no norm-on-load, no shift");
    WS_REQUIRE(c.mode == 0 && c.H > 0 && c.W > 0 && c.C > 0 && c.Ho > 0 && c.Wo > 0 && c.k >= 1 && c.sh >= 1 &&
                   c.sw >= 1 && c.p >= 0 && a->Kk == c.k * c.k * c.C && a->M % (c.Ho * c.Wo) == 0,
               "ws_gemm_tn: bad conv view (mode 0, Kk == k*k*C, M %% (Ho*Wo))");
    WS_REQUIRE(c.ldp == 0 || (c.ldp >= c.C && c.ldp % 4 == 0), "ws_gemm_tn: pixel stride ldp >= C, %% 4 (got %d)", c.ldp);
    WS_REQUIRE(c.k <= 5 && (long long)(c.H + 2 * c.k) * c.W * (c.ldp > 0 ? c.ldp : c.C) < (1LL << 31),
               "ws_gemm_tn: conv view: k <= 5 and one image below 2^31 elements");
  }
  dim3 grid(((maxn + 127) / 128) * ((maxk + 127) / 128), a->nsplit, ng), block(256);
  hipStream_t s = (hipStream_t)stream;
  ws_prof_begin(WS_PROF_GEMM_TN, s);
  if (a->vec & 4)
    ws_launch_gemm_tn_bf16(a, grid, s);
  else if (a->vec & 1)
    hipLaunchKernelGGL((gemm_tn_kernel<true>), grid, block, 0, s, *a);
  else
    hipLaunchKernelGGL((gemm_tn_kernel<false>), grid, block, 0, s, *a);
  ws_prof_end(WS_PROF_GEMM_TN, s);
  return ws_check_launch("ws_gemm_tn");
}

// -------------------------------------------------------------------------------------------
__global__ void reduce_slabs_kernel(const float* __restrict__ slab, int nsplit, long long stride,
                                    long long count, float* __restrict__ out, int w,
                                    long long ldo) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < count;
       i += (long long)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int k = 0; k < nsplit; ++k) s += slab[k * stride + i];
    const long long o = (w > 0) ? (i / w) * ldo + (i % w) : i;
    out[o] = s;
  }
}

// Few outputs, many splits (column sums, scalar reductions): the plain kernel would walk the splits
// serially in a handful of threads.  (a) count <= 8: one workgroup per output, splits strided over 256
// threads, block sum.  (b) 64 outputs x 4 split lanes per workgroup, lanes summed in fixed order through LDS.
__global__ __launch_bounds__(256) void reduce_slabs_few_kernel(const float* __restrict__ slab, int nsplit,
                                                               long long stride, float* __restrict__ out, int w,
                                                               long long ldo) {
  __shared__ float red[16];
  const long long i = blockIdx.x;
  float s = 0.f;
  for (int k = threadIdx.x; k < nsplit; k += 256) s += slab[k * stride + i];
  s = ws_block_sum(s, red);
  if (threadIdx.x == 0) out[(w > 0) ? (i / w) * ldo + (i % w) : i] = s;
}

// YL split lanes per output column (4 or 16): with hundreds of splits (the halo weight gradients of DPCCN: 342 slabs of
// 11 520 floats) four lanes walked 86 slabs each, one dependent add after the other -- 20-40 us per launch, 518 launches per
// step; sixteen lanes walk 22 each, and the partial sums meet in LDS in a fixed order (deterministic).
template <int YL>
__global__ __launch_bounds__(64 * YL) void reduce_slabs_2d_kernel(const float* __restrict__ slab, int nsplit,
                                                                  long long stride, long long count,
                                                                  float* __restrict__ out, int w, long long ldo) {
  __shared__ float part[YL][64];
  const int x = threadIdx.x & 63, y = threadIdx.x >> 6;
  const long long i = blockIdx.x * 64LL + x;
  float s0 = 0.f, s1 = 0.f;     // two chains: the loads of consecutive rounds are in flight together
  if (i < count) {
    int k = y;
    for (; k + YL < nsplit; k += 2 * YL) {
      s0 += slab[k * stride + i];
      s1 += slab[(k + YL) * stride + i];
    }
    if (k < nsplit) s0 += slab[k * stride + i];
  }
  part[y][x] = s0 + s1;
  __syncthreads();
  if (y == 0 && i < count) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < YL; q += 4) t += (part[q][x] + part[q + 1][x]) + (part[q + 2][x] + part[q + 3][x]);
    out[(w > 0) ? (i / w) * ldo + (i % w) : i] = t;
  }
}

extern "C" int ws_reduce_slabs(const float* slab, int nsplit, long long stride, long long count,
                               float* out, int w, long long ldo, void* stream) {
  WS_REQUIRE(slab && out && nsplit > 0 && count > 0, "ws_reduce_slabs: bad args");
  hipStream_t s = (hipStream_t)stream;
  if (count <= 8 && nsplit >= 64) {
    hipLaunchKernelGGL(reduce_slabs_few_kernel, dim3((unsigned)count), dim3(256), 0, s, slab, nsplit, stride, out, w,
                       ldo);
  } else if (count <= 65536 && nsplit >= 64) {
    hipLaunchKernelGGL(reduce_slabs_2d_kernel<16>, dim3((unsigned)((count + 63) / 64)), dim3(1024), 0, s, slab, nsplit,
                       stride, count, out, w, ldo);
  } else if (count <= 16384 && nsplit >= 16) {
    hipLaunchKernelGGL(reduce_slabs_2d_kernel<4>, dim3((unsigned)((count + 63) / 64)), dim3(256), 0, s, slab, nsplit,
                       stride, count, out, w, ldo);
  } else {
    const int threads = 256;
    long long blocks = (count + threads - 1) / threads;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)blocks), dim3(threads), 0, s, slab, nsplit, stride, count,
                       out, w, ldo);
  }
  return ws_check_launch("ws_reduce_slabs");
}

__global__ void transpose_kernel(const float* __restrict__ src, int rows, int cols, long long lds_,
                                 float* __restrict__ dst) {
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: ty 0..7
  for (int i = ty; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < rows && c < cols) ? src[(long long)r * lds_ + c] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + tx;
    if (c < cols && r < rows) dst[(long long)c * rows + r] = tile[tx][i];
  }
}

extern "C" int ws_transpose(const float* src, int rows, int cols, long long lds_, float* dst,
                            void* stream) {
  WS_REQUIRE(src && dst && rows > 0 && cols > 0 && lds_ >= cols, "ws_transpose: bad args");
  dim3 grid((cols + 31) / 32, (rows + 31) / 32);
  hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, rows, cols,
                     lds_, dst);
  return ws_check_launch("ws_transpose");
}
