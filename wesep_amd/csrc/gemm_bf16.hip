// Split-bf16 ("bf16x3") MFMA variants of the two GEMM families in gemm.hip, same arguments.
//
// fp32 operands are split on the way into LDS into hi = bf16(x) and lo = bf16(x - hi); the product
// is accumulated in fp32 as  a_hi*b_hi + a_hi*b_lo + a_lo*b_hi  on v_mfma_f32_32x32x16_bf16
// (dense bf16 matrix rate, 16x the fp32 MFMA rate -> ~5.3x per fp32-equivalent product).  The
// dropped a_lo*b_lo term is <= 2^-16 relative per product; accumulation stays fp32.  Whether that
// holds the path's 1e-3 waveform bound is decided by tests/test_bsrnn_gpu.py, not assumed.
//
// LDS image: 4 planes (A_hi, A_lo, B_hi, B_lo) of [128 rows][32 k] bf16, row stride 40 elements
// (80 B): the MFMA fragment (8 consecutive k of one row per lane) is ONE ds_read_b128, and the
// 80-B stride spreads a 16-lane group over all 64 banks (conflict-free).  Single LDS buffer
// (40 KB -> 3 workgroups/CU for latency hiding), next k-tile's global loads in flight in
// registers under the current tile's MFMAs.
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#define BT_BK 32
#define BT_LD 40               // bf16 elements per LDS row
#define BT_PLANE (128 * BT_LD)  // elements per plane

__device__ __forceinline__ int frag_row32b(int reg, int half) {
  return (reg & 3) + 8 * (reg >> 2) + 4 * half;
}

__device__ __forceinline__ void split4(const f32x4& v, bf16x4& hi, bf16x4& lo) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    hi[j] = (__bf16)v[j];
    lo[j] = (__bf16)(v[j] - (float)hi[j]);
  }
}

// one 128x128x32 tile step for a wave's 64x64 sub-tile; A planes hold the "row" operand
// (MFMA A, rows -> D rows), B planes the "column" operand (MFMA B, rows -> D columns).
__device__ __forceinline__ void tile_mma(const __bf16* __restrict__ lds, int arow0, int brow0, int l31,
                                         int half, f32x16& c00, f32x16& c01, f32x16& c10, f32x16& c11) {
  const __bf16* Ah = lds;
  const __bf16* Al = lds + BT_PLANE;
  const __bf16* Bh = lds + 2 * BT_PLANE;
  const __bf16* Bl = lds + 3 * BT_PLANE;
#pragma unroll
  for (int ks = 0; ks < BT_BK; ks += 16) {
    const int ka = ks + 8 * half;
    const int ra0 = (arow0 + l31) * BT_LD + ka, ra1 = (arow0 + 32 + l31) * BT_LD + ka;
    const int rb0 = (brow0 + l31) * BT_LD + ka, rb1 = (brow0 + 32 + l31) * BT_LD + ka;
    const bf16x8 a0h = *reinterpret_cast<const bf16x8*>(Ah + ra0);
    const bf16x8 a1h = *reinterpret_cast<const bf16x8*>(Ah + ra1);
    const bf16x8 b0h = *reinterpret_cast<const bf16x8*>(Bh + rb0);
    const bf16x8 b1h = *reinterpret_cast<const bf16x8*>(Bh + rb1);
    const bf16x8 a0l = *reinterpret_cast<const bf16x8*>(Al + ra0);
    const bf16x8 a1l = *reinterpret_cast<const bf16x8*>(Al + ra1);
    const bf16x8 b0l = *reinterpret_cast<const bf16x8*>(Bl + rb0);
    const bf16x8 b1l = *reinterpret_cast<const bf16x8*>(Bl + rb1);
    c00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0h, b0h, c00, 0, 0, 0);
    c01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0h, b1h, c01, 0, 0, 0);
    c10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1h, b0h, c10, 0, 0, 0);
    c11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1h, b1h, c11, 0, 0, 0);
    c00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0h, b0l, c00, 0, 0, 0);
    c01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0h, b1l, c01, 0, 0, 0);
    c10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1h, b0l, c10, 0, 0, 0);
    c11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1h, b1l, c11, 0, 0, 0);
    c00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0l, b0h, c00, 0, 0, 0);
    c01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0l, b1h, c01, 0, 0, 0);
    c10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1l, b0h, c10, 0, 0, 0);
    c11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1l, b1h, c11, 0, 0, 0);
  }
}

// narrow column tiles (NB = 1, 2 blocks of 32 columns): the four waves split the 128 rows, each owns 32 rows x 32*NB
// columns -- a layer with 16 or 32 output channels (every DPCCN convolution) does a quarter / half of the MFMA work
// of the 128-column tile instead of multiplying zero padding
template <int NB>
__device__ __forceinline__ void tile_mma_narrow(const __bf16* __restrict__ lds, int arow0, int l31, int half,
                                                f32x16 (&c)[2]) {
  const __bf16* Ah = lds;
  const __bf16* Al = lds + BT_PLANE;
  const __bf16* Bh = lds + 2 * BT_PLANE;
  const __bf16* Bl = lds + 3 * BT_PLANE;
#pragma unroll
  for (int ks = 0; ks < BT_BK; ks += 16) {
    const int ka = ks + 8 * half;
    const int ra = (arow0 + l31) * BT_LD + ka;
    const bf16x8 ah = *reinterpret_cast<const bf16x8*>(Ah + ra);
    const bf16x8 al = *reinterpret_cast<const bf16x8*>(Al + ra);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int rb = (32 * nb + l31) * BT_LD + ka;
      const bf16x8 bh = *reinterpret_cast<const bf16x8*>(Bh + rb);
      const bf16x8 bl = *reinterpret_cast<const bf16x8*>(Bl + rb);
      c[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c[nb], 0, 0, 0);
      c[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c[nb], 0, 0, 0);
      c[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c[nb], 0, 0, 0);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// NT: C[M][N] = epi(pro(A)[M][K] * W[N][K]^T).  Requires float4-loadable A and W (vec bits 0,1).
// ---------------------------------------------------------------------------------------------
// The k-loop body is branch-free (out-of-range rows / columns / k are clamped to a valid address and
// zeroed by a select; GroupNorm is a template parameter): with exec-masked loads the compiler's
// s_waitcnt placement drained every load before the MFMAs (no overlap at all), and the
// norm-on-load variant built that way produced run-to-run mismatches on MI355X
// (tools/nt_bug_probe.py).  Pointers taken from the group table are cast to the global address
// space so they do not decay to flat loads.
typedef const __attribute__((address_space(1))) float* gfp;

// CONV: the A operand is an implicit im2col matrix (ws_conv_view, wesep_hip.h): a row is an output pixel, a k-tile's
// float4 lies inside one tap (C % 4 == 0), so the loader adds the tap's offset to the pixel's base and masks taps that
// fall outside the image -- the patch matrix (k*k times the activation) never exists in HBM.
// NB: 32-column blocks per workgroup tile (4 = the 128 x 128 tile with 2 x 2 waves of 64 x 64; 1 / 2 = narrow, above).
// (launch bounds: the SECOND number is hipcc's minimum of waves per SIMD, not workgroups per CU -- "2" let the 128 x 128 tile take
//  174 / 184 registers = two workgroups per CU where 40 KB of LDS allow three; at 3 the compiler fits it in 162 without a spill.
//  The NORM variant would spill 30 registers there and stays at 2.)
// WT (vec bit 3, round 6): W is stored TRANSPOSED -- W'[n][k] = W[k * ldw + n], i.e. the product is A [M, K] x B [K, N] with B as
// it lies in memory ("NN").  A thread loads a 4 x 4 block (four consecutive n of four consecutive k), transposes it in registers
// and writes four 8-byte k-runs into the [n][k] LDS image -- the store shape of the plain path (a first cut scattered 2-byte
// stores: 8-way bank conflicts, the GEMMs lost more than the copies had cost); everything behind the staging is the same kernel.  For the attention products of
// TF-GridNet (att x V, and d(logits) x K / d(ov) x V^T of the backward): each of them used to copy its B operand into the other
// orientation first -- four ATen copies of 0.15 - 0.57 ms per block and step (profiles/r06_c37_tfg_step_timeline.txt).
template <bool NORM, bool CONV, int NB, bool WT = false>
__global__ __launch_bounds__(256, NORM ? 2 : 3) void gemm_nt_bf16_kernel(const ws_gemm_nt_args p) {
  static_assert(!WT || (!NORM && !CONV), "the transposed-W staging exists for the plain kernel");
  __shared__ __attribute__((aligned(16))) __bf16 lds[4 * BT_PLANE];  // 40 KB
  const float* A_ = p.A;
  const float* W_ = p.W;
  const float* bias = p.bias;
  const float* gamma_ = p.gamma;
  const float* beta_ = p.beta;
  float* C = p.C;
  const float* R = p.R;
  const float* T = p.T;
  int K = p.K, N = p.N, ldw = p.ldw;
  long long st_base = p.st_base;
  if (p.groups) {
    const ws_group_nt g = p.groups[blockIdx.z];
    A_ += g.a_off;
    W_ = g.W;
    bias = g.bias;
    gamma_ = g.gamma;
    beta_ = g.beta;
    C += g.c_off;
    if (R) R += g.c_off;
    if (T) T += g.c_off;
    st_base = g.st_base;
    K = g.K;
    N = g.N;
    ldw = g.ldw;
  }
  const gfp A = (gfp)A_, W = (gfp)W_, gamma = (gfp)gamma_, beta = (gfp)beta_, stats = (gfp)p.stats;
  const int M = p.M;
  const int m_blk = blockIdx.x * 128, n_blk = blockIdx.y * (32 * NB);
  if (n_blk >= N) return;
  const int tid = threadIdx.x;
  const int lrow = tid >> 3, lk = (tid & 7) * 4;

  long long aoff[4], woff[4];
  bool vm[4], vn[4];
  float mean[4], rstd[4];
  int cbh[4], cbw[4];  // CONV: the pixel's base coordinates (mode 0: ho*sh - p; mode 1: ho + p); aoff = image base
  const ws_conv_view cv = p.conv;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m_blk + lrow + 32 * i;
    vm[i] = m < M;
    const int mm = vm[i] ? m : M - 1;
    if (CONV) {
      const int hw = cv.Ho * cv.Wo;
      const int r = mm / hw, q = mm - r * hw;
      const int ho = q / cv.Wo, wo = q - ho * cv.Wo;
      aoff[i] = (long long)r * cv.H * cv.W * (cv.ldp > 0 ? cv.ldp : cv.C);
      cbh[i] = cv.mode == 0 ? ho * cv.sh - cv.p : ho + cv.p;
      cbw[i] = cv.mode == 0 ? wo * cv.sw - cv.p : wo + cv.p;
    } else {
      aoff[i] = ws_row_off(mm, p.a_div, p.a_s1, p.a_s2);
      cbh[i] = cbw[i] = 0;
    }
    mean[i] = 0.f;
    rstd[i] = 1.f;
    if (NORM) {
      const long long s = (long long)(mm / p.st_div1) * p.st_m1 + (long long)(mm % p.st_div2) * p.st_m2 + st_base;
      mean[i] = stats[2 * s];
      rstd[i] = stats[2 * s + 1];
    }
    const int n = n_blk + lrow + 32 * i;
    vn[i] = n < N && i < NB;
    woff[i] = (long long)(n < N ? n : N - 1) * ldw;
    if constexpr (WT) {   // this thread's 4 x 4 block of the [32 k][32 NB n] tile: column quad n4, k quad k4; i = row of the block
      const int n4 = (tid >> 6) * 8 + (tid & 7), nn = n_blk + 4 * n4;
      vn[i] = tid < 64 * NB && nn < N;           // (N % 4 == 0: a quad is in or out as a whole)
      woff[i] = vn[i] ? nn : 0;
    }
  }

  // load_tile only ISSUES the loads (raw values stay in registers under the MFMAs of the current tile);
  // normalisation, range masking and the bf16 split happen in store_tile, one iteration later
  f32x4 ra[4], rw[4], gm = {1.f, 1.f, 1.f, 1.f}, bt = {0.f, 0.f, 0.f, 0.f};
  bool vk = true;
  bool vt[4] = {true, true, true, true};  // CONV: the tap of this tile's float4 lies inside the image for row i
  bool vkw[4] = {true, true, true, true}; // WT: this thread's k row of the W tile lies inside K
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  const int csh = cv.sh - 1, csw = cv.sw - 1;  // mode 1: strides are 1 or 2 -> shift / mask instead of a division
  const int cdil = cv.dil > 0 ? cv.dil : 1;
  auto load_tile = [&](int kt) {
    const int k = kt * BT_BK + lk;
    vk = k < K;                       // K % 4 == 0 on this path: a float4 is in or out as a whole
    const int kc = vk ? k : 0;
    if (NORM) {
      gm = *reinterpret_cast<const __attribute__((address_space(1))) f32x4*>(gamma + kc);
      bt = *reinterpret_cast<const __attribute__((address_space(1))) f32x4*>(beta + kc);
    }
    int ky = 0, kx = 0, cc = 0;
    if (CONV) {
      const int tap = kc / cv.C;
      cc = kc - tap * cv.C;
      ky = tap / cv.k;
      kx = (tap - ky * cv.k) * cdil;
      ky *= cdil;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (CONV) {
        int hh, ww;
        bool ok;
        if (cv.mode == 0) {
          hh = cbh[i] + ky;
          ww = cbw[i] + kx;
          ok = (unsigned)hh < (unsigned)cv.H && (unsigned)ww < (unsigned)cv.W;
        } else {
          const int hn = cbh[i] - ky, wn = cbw[i] - kx;
          hh = hn >> csh;
          ww = wn >> csw;
          ok = hn >= 0 && wn >= 0 && (hn & csh) == 0 && (wn & csw) == 0 && hh < cv.H && ww < cv.W;
        }
        vt[i] = ok;
        const long long o = ok ? aoff[i] + ((hh * cv.W + ww) * (cv.ldp > 0 ? cv.ldp : cv.C) + cc) : 0;  // one image < 2^31 elements (checked)
        ra[i] = *reinterpret_cast<const __attribute__((address_space(1))) f32x4*>(A + o);
      } else {
        ra[i] = *reinterpret_cast<const __attribute__((address_space(1))) f32x4*>(A + aoff[i] + kc);
      }
      if constexpr (WT) {
        const int kr = kt * BT_BK + 4 * ((tid >> 3) & 7) + i;
        vkw[i] = kr < K;
        rw[i] = *reinterpret_cast<const __attribute__((address_space(1))) f32x4*>(W + (long long)(vkw[i] ? kr : 0) * ldw + woff[i]);
      } else {
        if (i < NB) rw[i] = *reinterpret_cast<const __attribute__((address_space(1))) f32x4*>(W + woff[i] + kc);
      }
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f32x4 v = ra[i];
      if (NORM) v = (v - mean[i]) * rstd[i] * gm + bt;
      v = (vm[i] && vk && vt[i]) ? v : zero4;
      const f32x4 wv = (vn[i] && vk) ? rw[i] : zero4;
      bf16x4 hi, lo;
      const int o = (lrow + 32 * i) * BT_LD + lk;
      split4(v, hi, lo);
      *reinterpret_cast<bf16x4*>(lds + o) = hi;
      *reinterpret_cast<bf16x4*>(lds + BT_PLANE + o) = lo;
      if constexpr (WT) {
        if (tid < 64 * NB) {     // column j = i of the block: its four k, transposed in registers
          const int n4 = (tid >> 6) * 8 + (tid & 7), k4 = (tid >> 3) & 7;
          const f32x4 wt = {(vn[0] && vkw[0]) ? rw[0][i] : 0.f, (vn[1] && vkw[1]) ? rw[1][i] : 0.f,
                            (vn[2] && vkw[2]) ? rw[2][i] : 0.f, (vn[3] && vkw[3]) ? rw[3][i] : 0.f};
          split4(wt, hi, lo);
          const int ow = (4 * n4 + i) * BT_LD + 4 * k4;
          *reinterpret_cast<bf16x4*>(lds + 2 * BT_PLANE + ow) = hi;
          *reinterpret_cast<bf16x4*>(lds + 3 * BT_PLANE + ow) = lo;
        }
      } else if (i < NB) {
        split4(wv, hi, lo);
        *reinterpret_cast<bf16x4*>(lds + 2 * BT_PLANE + o) = hi;
        *reinterpret_cast<bf16x4*>(lds + 3 * BT_PLANE + o) = lo;
      }
    }
  };

  const int wave = tid >> 6, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;
  f32x16 acc00, acc01, acc10, acc11;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc00[r] = acc01[r] = acc10[r] = acc11[r] = 0.f;
  f32x16 accn[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) accn[0][r] = accn[1][r] = 0.f;

  const int nk = (K + BT_BK - 1) / BT_BK;
  load_tile(0);
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();  // previous tile's fragment reads are done
    store_tile();
    __syncthreads();
    load_tile(min(kt + 1, nk - 1));  // branch-free: the last iteration reloads its own tile (unused)
    __builtin_amdgcn_sched_barrier(0);  // issue the loads HERE: they must fly under the MFMAs below
    if constexpr (NB == 4) tile_mma(lds, wm * 64, wn * 64, l31, half, acc00, acc01, acc10, acc11);
    else tile_mma_narrow<NB>(lds, wave * 32, l31, half, accn);
    __builtin_amdgcn_sched_barrier(0);
  }

  const bool flat_c = p.c_div >= M;  // (m / c_div) == 0 for every row: no division needed
  // Vector epilogue (round 2): the accumulators hold one COLUMN per lane; stored that way a row costs 4-byte accesses
  // behind per-element branches, and the residual / activation-derivative loads cannot move above the stores (C may
  // alias R) -- one exposed memory round trip per register (measured on gemm_b2p: 0.4 of 0.59 ms).  When every row piece
  // is 16-byte addressable the tile is staged through the (now free) LDS planes, 10 KB per wave, and a lane moves
  // float4 pieces of whole rows: all R / T loads of a pass are issued before its first store; a narrow tile (16 output
  // channels) becomes 1 KB of contiguous stores per wave instead of 64-byte rows.
  const bool vec_epi = (N & 3) == 0 && (p.c_s1 & 3) == 0 && (p.c_s2 & 3) == 0 && ((size_t)C & 15) == 0 &&
                       (!R || ((size_t)R & 15) == 0) && (!T || ((size_t)T & 15) == 0);
  if (vec_epi) {
    constexpr int W = NB == 4 ? 64 : 32 * NB;  // columns per pass
    constexpr int LDW = W + 4;                 // 32 x 68 floats = 8.5 KB of the wave's 10 KB
    constexpr int NPASS = NB == 4 ? 2 : 1;
    constexpr int C4 = W / 4, NI = 32 * C4 / 64;
    __syncthreads();                            // the last tile's fragment reads are done: the planes are free
    float* stg = reinterpret_cast<float*>(lds) + wave * 2560;
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
#pragma unroll
      for (int tn = 0; tn < W / 32; ++tn) {
        const f32x16& acc = NB == 4 ? (pass == 0 ? (tn ? acc01 : acc00) : (tn ? acc11 : acc10)) : accn[tn];
#pragma unroll
        for (int r = 0; r < 16; ++r) stg[frag_row32b(r, half) * LDW + tn * 32 + l31] = acc[r];
      }
      __syncthreads();
      const int row0 = NB == 4 ? wm * 64 + pass * 32 : wave * 32, col0 = NB == 4 ? wn * 64 : 0;
      f32x4 v[NI], tv[NI], rv[NI];
      long long off[NI];
#pragma unroll
      for (int q = 0; q < NI; ++q) {
        const int idx = lane + 64 * q, rr = idx / C4, c4 = idx - rr * C4;
        const int m = m_blk + row0 + rr, n = n_blk + col0 + 4 * c4;
        const bool ok = m < M && n < N;
        off[q] = ok ? (flat_c ? (long long)m * p.c_s2 : ws_row_off(m, p.c_div, p.c_s1, p.c_s2)) + n : -1;
        v[q] = *reinterpret_cast<const f32x4*>(stg + rr * LDW + 4 * c4);
        if (bias && ok) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[q][j] += bias[n + j];
        }
        if (T && ok) tv[q] = *reinterpret_cast<const f32x4*>(T + off[q]);
        if (R && ok) rv[q] = *reinterpret_cast<const f32x4*>(R + off[q]);
      }
#pragma unroll
      for (int q = 0; q < NI; ++q) {
        if (off[q] < 0) continue;
        f32x4 o = v[q];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (p.act == 1) o[j] = tanhf(o[j]);
          if (p.act == 2) o[j] = fmaxf(o[j], 0.f);
          if (T) o[j] *= p.act == 4 ? (tv[q][j] > 0.f ? 1.f : 0.f) : (1.f - tv[q][j] * tv[q][j]);
          if (R) o[j] += rv[q][j];
        }
        *reinterpret_cast<f32x4*>(C + off[q]) = o;
      }
      if (pass + 1 < NPASS) __syncthreads();    // staging area free for the second half of the wave's tile
    }
    return;
  }
  // (row0, col0) of a 32 x 32 accumulator block inside the workgroup tile
  auto epilogue = [&](const f32x16& acc, int tm, int tn) {
    const int col0 = NB == 4 ? wn * 64 + tn * 32 : tn * 32, row0 = NB == 4 ? wm * 64 + tm * 32 : wave * 32;
    const int n = n_blk + col0 + l31;
    if (n >= N) return;
    const float bv = bias ? bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m_blk + row0 + frag_row32b(r, half);
      if (m >= M) continue;
      const long long off = (flat_c ? (long long)m * p.c_s2 : ws_row_off(m, p.c_div, p.c_s1, p.c_s2)) + n;
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
  if constexpr (NB == 4) {
    epilogue(acc00, 0, 0);
    epilogue(acc01, 0, 1);
    epilogue(acc10, 1, 0);
    epilogue(acc11, 1, 1);
  } else {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) epilogue(accn[nb], 0, nb);
  }
}

// Column-tile width of a launch: ungrouped launches without norm-on-load and at most 64 output columns (the
// convolutions and 1x1 convolutions of DPCCN / the ResNet stem) use the narrow tiles; everything else -- in particular
// every launch of the pBSRNN path (grouped and / or normalising) -- keeps the 128-column tile of round 1.
int ws_gemm_nt_bf16_nb(const ws_gemm_nt_args* a) {
  if (a->groups || a->stats) return 4;
  return a->N <= 32 ? 1 : (a->N <= 64 ? 2 : 4);
}

int ws_launch_gemm_nt_bf16(const ws_gemm_nt_args* a, dim3 grid, hipStream_t s) {
  const int nb = ws_gemm_nt_bf16_nb(a);
  if (nb != 4) grid.y = (a->N + 32 * nb - 1) / (32 * nb);
#define WS_NT_LAUNCH(NORM_, CONV_)                                                                        \
  do {                                                                                                    \
    if (nb == 1) hipLaunchKernelGGL((gemm_nt_bf16_kernel<NORM_, CONV_, 1>), grid, dim3(256), 0, s, *a);      \
    else if (nb == 2) hipLaunchKernelGGL((gemm_nt_bf16_kernel<NORM_, CONV_, 2>), grid, dim3(256), 0, s, *a); \
    else hipLaunchKernelGGL((gemm_nt_bf16_kernel<NORM_, CONV_, 4>), grid, dim3(256), 0, s, *a);              \
  } while (0)
  if (a->vec & 8) {   // W stored [K][N] (ws_gemm_nt checks: no conv view, no norm-on-load, N % 4 == 0, ldw % 4 == 0)
    if (nb == 1) hipLaunchKernelGGL((gemm_nt_bf16_kernel<false, false, 1, true>), grid, dim3(256), 0, s, *a);
    else if (nb == 2) hipLaunchKernelGGL((gemm_nt_bf16_kernel<false, false, 2, true>), grid, dim3(256), 0, s, *a);
    else hipLaunchKernelGGL((gemm_nt_bf16_kernel<false, false, 4, true>), grid, dim3(256), 0, s, *a);
  } else if (a->conv.on)
    WS_NT_LAUNCH(false, true);
  else if (a->stats)
    hipLaunchKernelGGL((gemm_nt_bf16_kernel<true, false, 4>), grid, dim3(256), 0, s, *a);
  else
    WS_NT_LAUNCH(false, false);
#undef WS_NT_LAUNCH
  return 0;
}

// ---------------------------------------------------------------------------------------------
// TN: slab[n][k] = sum_m G[m][n] * pro(A[m'][k]).  The MFMA wants 8 consecutive m per lane for a
// fixed n (and k), i.e. the TRANSPOSE of both global tiles: each thread owns one column (n or k)
// and gathers 16 consecutive rows with scalar loads (coalesced across lanes), then writes two
// 16-B fragments per plane.  Row metadata (offsets, validity, norm stats) for the 32 rows of a
// tile is computed once by 32 threads into LDS instead of 16x per thread.
// ---------------------------------------------------------------------------------------------
struct RowMeta {
  long long goff, aoff;
  float mean, rstd;
  int gvalid, avalid;
  int tapmask, pad_;  // CONV: bit (ky*k + kx) = that tap of the row's pixel lies inside the image; aoff then addresses
                      // tap (0, 0) of the pixel (possibly before the image: only dereferenced through a valid tap)
};

// CONV: A is the implicit im2col matrix (mode 0) of the image p.A points to; this thread's column is one (tap, channel).
// NARROW: at most 32 gradient columns per tile (16 / 32 output channels): the four waves split the 128 A columns and
// each runs one 32 x 32 accumulator -- a quarter of the MFMA work of the 128 x 128 tile, which is what bounds the
// weight gradients of the DPCCN convolutions (M = 4.1 M rows, Nn = 16).
template <bool CONV, bool NARROW>
__global__ __launch_bounds__(256, 3) void gemm_tn_bf16_kernel(const ws_gemm_tn_args p) {
  __shared__ __attribute__((aligned(16))) __bf16 lds[4 * BT_PLANE];  // 40 KB
  __shared__ RowMeta meta[2][32];
  __shared__ float bred[128];
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
  constexpr int TNW = NARROW ? 32 : 128;  // gradient columns per tile
  const int tiles_k = (Kk + 127) / 128, tiles_n = (Nn + TNW - 1) / TNW;
  if ((int)blockIdx.x >= tiles_k * tiles_n) return;
  const int n_blk = (blockIdx.x / tiles_k) * TNW, k_blk = (blockIdx.x % tiles_k) * 128;
  const int split = blockIdx.y;
  const int m_begin = split * p.rows_per_split;
  const int m_end = min(p.M, m_begin + p.rows_per_split);
  const int tid = threadIdx.x;
  const int col = tid & 127, mg = (tid >> 7) * 16;  // this thread: column `col`, rows mg..mg+15
  const bool has_norm = p.stats != nullptr;
  const bool do_bias = p.bslab != nullptr && k_blk == 0;
  const bool gcol_ok = col < TNW && n_blk + col < Nn, acol_ok = k_blk + col < Kk;
  float gm = 1.f, bt = 0.f;
  if (has_norm && acol_ok) {
    gm = gamma[k_blk + col];
    bt = beta[k_blk + col];
  }

  auto make_meta = [&](int m0, int slot) {
    if (tid < 32) {
      const int m = m0 + tid;
      RowMeta r;
      r.gvalid = m < m_end;
      r.avalid = r.gvalid;
      r.goff = r.aoff = 0;
      r.mean = 0.f;
      r.rstd = 1.f;
      r.tapmask = r.pad_ = 0;
      if (CONV && r.gvalid) {
        const ws_conv_view cv = p.conv;
        const int hw = cv.Ho * cv.Wo;
        const int rr = m / hw, q = m - rr * hw;
        const int ho = q / cv.Wo, wo = q - ho * cv.Wo;
        const int bh = ho * cv.sh - cv.p, bw = wo * cv.sw - cv.p, dl = cv.dil > 0 ? cv.dil : 1;
        r.goff = ws_row_off(m, p.g_div, p.g_s1, p.g_s2);
        const int cld = cv.ldp > 0 ? cv.ldp : cv.C;
        r.aoff = (long long)rr * cv.H * cv.W * cld + (long long)(bh * cv.W + bw) * cld;
        int mask = 0;
        for (int ky = 0; ky < cv.k; ++ky)
          for (int kx = 0; kx < cv.k; ++kx)
            if ((unsigned)(bh + ky * dl) < (unsigned)cv.H && (unsigned)(bw + kx * dl) < (unsigned)cv.W)
              mask |= 1 << (ky * cv.k + kx);
        r.tapmask = mask;
      } else if (r.gvalid) {
        r.goff = ws_row_off(m, p.g_div, p.g_s1, p.g_s2);
        int ma = m;
        if (p.shift_rows != 0) {
          const int t = (m / p.seq_div) % p.seq_len;
          const int t2 = t + (p.shift_rows > 0 ? 1 : -1);
          r.avalid = (t2 >= 0) && (t2 < p.seq_len);
          ma = m + p.shift_rows;
        }
        if (r.avalid) r.aoff = ws_row_off(ma, p.a_div, p.a_s1, p.a_s2);
        if (has_norm) {
          const long long s = (long long)(m / p.st_div1) * p.st_m1 + (long long)(m % p.st_div2) * p.st_m2 + st_base;
          r.mean = p.stats[2 * s];
          r.rstd = p.stats[2 * s + 1];
        }
      }
      meta[slot][tid] = r;
    }
  };

  // Branch-free tile loads (see gemm_nt_bf16_kernel): invalid rows carry offset 0 in the meta table and
  // out-of-range columns are clamped, so every load is unconditional; masking, normalisation and the
  // bf16 split happen in store_tile one iteration later, with the loads in flight under the MFMAs.
  const gfp Gg = (gfp)G + n_blk + (gcol_ok ? col : 0);
  gfp Ag = (gfp)A + k_blk + (acol_ok ? col : 0);
  int ctap = 0, ctapo = 0;  // CONV: this thread's tap and its element offset from a pixel's tap (0, 0); Ag = its channel
  if (CONV) {
    const int kc = acol_ok ? k_blk + col : 0;
    ctap = kc / p.conv.C;
    const int dl = p.conv.dil > 0 ? p.conv.dil : 1;
    const int cky = (ctap / p.conv.k) * dl, ckx = (ctap % p.conv.k) * dl;
    ctapo = (cky * p.conv.W + ckx) * (p.conv.ldp > 0 ? p.conv.ldp : p.conv.C);
    Ag = (gfp)A + (kc - ctap * p.conv.C);
  }
  float rg[16], ra[16];
  unsigned tapok = 0xffffu;  // CONV: bit j = the tap lies inside the image for row j of the tile in registers
  auto load_tile = [&](int slot) {
    unsigned okb = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const RowMeta& r = meta[slot][mg + j];
      rg[j] = Gg[r.goff];
      if (CONV) {
        const bool ok = (r.tapmask >> ctap) & 1;   // (0 for rows past the split's end)
        okb |= ok ? 1u << j : 0u;
        // unconditional load; a masked tap reads element 0 of the thread's channel -- NOT "its tap of pixel (0, 0)":
        // with a dilated tap of a one-row image (ECAPA) that address lies k/2 * dil rows behind a small tensor's end
        ra[j] = Ag[ok ? r.aoff + ctapo : 0];
      } else {
        ra[j] = Ag[r.aoff];
      }
    }
    if (CONV) tapok = okb;
  };
  auto store_tile = [&](int slot) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const RowMeta& r = meta[slot][mg + j];
      rg[j] = (r.gvalid && gcol_ok) ? rg[j] : 0.f;
      float a = ra[j];
      if (has_norm) a = (a - r.mean) * r.rstd * gm + bt;
      ra[j] = (r.avalid && acol_ok && (!CONV || ((tapok >> j) & 1u))) ? a : 0.f;
    }
#pragma unroll
    for (int h8 = 0; h8 < 2; ++h8) {
      bf16x8 ghi, glo, ahi, alo;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float g = rg[h8 * 8 + j], a = ra[h8 * 8 + j];
        ghi[j] = (__bf16)g;
        glo[j] = (__bf16)(g - (float)ghi[j]);
        ahi[j] = (__bf16)a;
        alo[j] = (__bf16)(a - (float)ahi[j]);
      }
      const int o = col * BT_LD + mg + h8 * 8;
      *reinterpret_cast<bf16x8*>(lds + o) = ghi;
      *reinterpret_cast<bf16x8*>(lds + BT_PLANE + o) = glo;
      *reinterpret_cast<bf16x8*>(lds + 2 * BT_PLANE + o) = ahi;
      *reinterpret_cast<bf16x8*>(lds + 3 * BT_PLANE + o) = alo;
    }
  };

  const int wave = tid >> 6, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;
  f32x16 acc00, acc01, acc10, acc11;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc00[r] = acc01[r] = acc10[r] = acc11[r] = 0.f;
  float bsum = 0.f;

  int it = 0;
  if (m_begin < m_end) {
    make_meta(m_begin, 0);
    __syncthreads();
    load_tile(0);
  }
  for (int m0 = m_begin; m0 < m_end; m0 += 32, ++it) {
    // meta of the NEXT tile (rows past m_end are marked invalid, offset 0), so the prefetch below is
    // unconditional; slot it&1 still describes the tile now in registers
    make_meta(m0 + 32, (it + 1) & 1);
    __syncthreads();  // previous tile's fragment reads done; next meta visible
    store_tile(it & 1);
    if (do_bias) {  // after masking: invalid rows contribute 0
#pragma unroll
      for (int j = 0; j < 16; ++j) bsum += rg[j];
    }
    __syncthreads();
    load_tile((it + 1) & 1);
    __builtin_amdgcn_sched_barrier(0);  // the loads are issued here and fly under the MFMAs
    if constexpr (NARROW) {
      // D[n][k]: rows = the 32 gradient columns (G planes, rows 0..31), columns = this wave's 32 A columns
#pragma unroll
      for (int ks = 0; ks < BT_BK; ks += 16) {
        const int ra_ = l31 * BT_LD + ks + 8 * half, rb_ = (wave * 32 + l31) * BT_LD + ks + 8 * half;
        const bf16x8 gh = *reinterpret_cast<const bf16x8*>(lds + ra_);
        const bf16x8 gl = *reinterpret_cast<const bf16x8*>(lds + BT_PLANE + ra_);
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(lds + 2 * BT_PLANE + rb_);
        const bf16x8 al = *reinterpret_cast<const bf16x8*>(lds + 3 * BT_PLANE + rb_);
        acc00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gh, ah, acc00, 0, 0, 0);
        acc00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gh, al, acc00, 0, 0, 0);
        acc00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gl, ah, acc00, 0, 0, 0);
      }
    } else {
      tile_mma(lds, wm * 64, wn * 64, l31, half, acc00, acc01, acc10, acc11);
    }
    __builtin_amdgcn_sched_barrier(0);
  }

  float* out = p.slab + (long long)split * p.slab_stride + out_off;
  auto write = [&](const f32x16& acc, int tm, int tn) {
    const int k = k_blk + (NARROW ? wave * 32 : wn * 64 + tn * 32) + l31;
    if (k >= Kk) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = n_blk + (NARROW ? 0 : wm * 64 + tm * 32) + frag_row32b(r, half);
      if (n < Nn) out[(long long)n * Kk + k] = acc[r];
    }
  };
  write(acc00, 0, 0);
  if constexpr (!NARROW) {
    write(acc01, 0, 1);
    write(acc10, 1, 0);
    write(acc11, 1, 1);
  }
  if (do_bias) {
    __syncthreads();
    if (tid >= 128) bred[col] = bsum;
    __syncthreads();
    if (tid < 128 && gcol_ok)
      p.bslab[(long long)split * p.bslab_stride + bout_off + n_blk + col] = bsum + bred[col];
  }
}

int ws_launch_gemm_tn_bf16(const ws_gemm_tn_args* a, dim3 grid, hipStream_t s) {
  // narrow gradient tiles for ungrouped launches with at most 32 gradient columns (see the kernel); the grouped /
  // normalising launches of the pBSRNN path keep the 128 x 128 tile
  const bool narrow = !a->groups && !a->stats && a->Nn <= 32;
  if (narrow) grid.x = ((a->Nn + 31) / 32) * ((a->Kk + 127) / 128);
  if (a->conv.on && narrow)
    hipLaunchKernelGGL((gemm_tn_bf16_kernel<true, true>), grid, dim3(256), 0, s, *a);
  else if (a->conv.on)
    hipLaunchKernelGGL((gemm_tn_bf16_kernel<true, false>), grid, dim3(256), 0, s, *a);
  else if (narrow)
    hipLaunchKernelGGL((gemm_tn_bf16_kernel<false, true>), grid, dim3(256), 0, s, *a);
  else
    hipLaunchKernelGGL((gemm_tn_bf16_kernel<false, false>), grid, dim3(256), 0, s, *a);
  return 0;
}
