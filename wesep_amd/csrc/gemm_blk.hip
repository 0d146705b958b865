// Split-bf16 GEMMs between the plain Z layout and the blocked layout BL (include/wesep_hip.h) --
// the dense layers around the BLSTM recurrence of ResRNN (wesep/models/bsrnn.py:38-46) and their
// gradients.  All are HBM-bound by construction (huge M = positions, small weights), so the design
// goal is: every activation byte crosses HBM once, as part of a fully used, contiguous wave-level
// access, and lands directly in MFMA fragment registers.
//
//   ws_gemm_p2b  plain rows -> BL   C[(b,i)][n] = sum_k pro(A[pos(b,i)][k]) W[n][k] + bias[n]
//                (x-projection with GroupNorm-on-load; d(hcat) = dout Wp).  K = 128.
//   ws_gemm_b2p  BL -> plain rows   C[pos(b,i)][n] = sum_k A[(b,i)][k] W[n][k] + bias[n] + R
//                (projection + residual; d(xn) = dgates Wcat).  N = 128.
//   ws_gemm_tnb  BL x BL -> weights out[g][a] = sum_(b,i) G[(b,i)][g] A[(b',i)][a]
//                (dW_ih | dW_hh in one pass over dgates, dW_proj, bias gradients).
//
// BL(C): rows in blocks of 32 (block b = tile*L + step: the 32 sequences of an LSTM workgroup at
// one step), element (b, i, c) at  b*32*C + ((c>>2)*32 + i)*4 + (c&3).  A 32-lane group that
// moves 16 bytes per lane for consecutive slots i touches one contiguous 512-byte run; a
// [32 slots x 128 columns] sub-block is one contiguous 16 KB.  Products are split-bf16 (hi/lo,
// 3 MFMAs, fp32 accumulate) on v_mfma_f32_32x32x16_bf16; weights are pre-split and pre-ordered
// into fragment order by ws_pack_w.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));   // 32 FP8 operand bytes of v_mfma_scale_f32_32x32x64_f8f6f4
typedef __fp16 h16x2 __attribute__((ext_vector_type(2)));  // the operand type of __builtin_amdgcn_fdot2 / cvt_pkrtz

__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ void split8(const float* v, bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    hi[j] = (__bf16)v[j];
    lo[j] = (__bf16)(v[j] - (float)hi[j]);
  }
}

// BLS (split-bf16 storage, wesep_hip.h): 8 consecutive k of a slot = two 16-byte cells of packed hi|lo elements
// <-> the hi / lo MFMA fragments; both directions are 8 v_perm_b32 (a 16-bit 2x2 transpose per element pair)
__device__ __forceinline__ void unpack8(const u32x4& c0, const u32x4& c1, bf16x8& hi, bf16x8& lo) {
  u32x4 h, l;
  h[0] = __builtin_amdgcn_perm(c0[1], c0[0], WS_SEL_HI16);
  h[1] = __builtin_amdgcn_perm(c0[3], c0[2], WS_SEL_HI16);
  h[2] = __builtin_amdgcn_perm(c1[1], c1[0], WS_SEL_HI16);
  h[3] = __builtin_amdgcn_perm(c1[3], c1[2], WS_SEL_HI16);
  l[0] = __builtin_amdgcn_perm(c0[1], c0[0], WS_SEL_LO16);
  l[1] = __builtin_amdgcn_perm(c0[3], c0[2], WS_SEL_LO16);
  l[2] = __builtin_amdgcn_perm(c1[1], c1[0], WS_SEL_LO16);
  l[3] = __builtin_amdgcn_perm(c1[3], c1[2], WS_SEL_LO16);
  hi = __builtin_bit_cast(bf16x8, h);
  lo = __builtin_bit_cast(bf16x8, l);
}
__device__ __forceinline__ void pack8(const bf16x8& hi, const bf16x8& lo, u32x4& c0, u32x4& c1) {
  const u32x4 h = __builtin_bit_cast(u32x4, hi), l = __builtin_bit_cast(u32x4, lo);
  c0[0] = __builtin_amdgcn_perm(h[0], l[0], WS_SEL_LO16);
  c0[1] = __builtin_amdgcn_perm(h[0], l[0], WS_SEL_HI16);
  c0[2] = __builtin_amdgcn_perm(h[1], l[1], WS_SEL_LO16);
  c0[3] = __builtin_amdgcn_perm(h[1], l[1], WS_SEL_HI16);
  c1[0] = __builtin_amdgcn_perm(h[2], l[2], WS_SEL_LO16);
  c1[1] = __builtin_amdgcn_perm(h[2], l[2], WS_SEL_HI16);
  c1[2] = __builtin_amdgcn_perm(h[3], l[3], WS_SEL_LO16);
  c1[3] = __builtin_amdgcn_perm(h[3], l[3], WS_SEL_HI16);
}

// position (row of the plain layout) of slot i of block b; `valid` false for padded slots
__device__ __forceinline__ long long seq_pos(const ws_seqmap& sm, int b, int i, bool& valid) {
  const int tile = b / sm.L, step = b - tile * sm.L;
  const int seq = tile * 32 + i;
  const int nv = sm.nvalid > 0 ? sm.nvalid : sm.nseq;  // (ABI v15: sequences >= nvalid are padding)
  valid = seq < nv;
  const int s = valid ? seq : nv - 1;
  return (long long)(s / sm.sq_div) * sm.sq_s1 + (long long)(s % sm.sq_div) * sm.sq_s2 +
         (long long)step * sm.step_rows;
}

// ---------------------------------------------------------------------------------------------
// weight packing: W'[n][k] = trans ? W[k*ldw + n] : W[n*ldw + k], split into bf16 hi/lo, in
// 16-byte units of 8 consecutive k for the MFMA lane that will load them:
//   element j of unit u*64 + lane = part( W'[32*nt + (lane&31)][16*ks + 8*(lane>>5) + j] )
//   order 0 (p2b): u = (nt*(K/16) + ks)*2 + part      order 1 (b2p): u = (ks*(N/32) + nt)*2 + part
// ---------------------------------------------------------------------------------------------
__global__ void pack_w_kernel(const float* __restrict__ W, int N, int K, long long ldw, int trans,
                              int order, __bf16* __restrict__ out) {
  // one thread per 16-byte unit pair (hi, lo): 8 consecutive k of one row n
  const int nks = K / 16, nnt = N / 32;
  const int units = N * K / 8;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= units) return;
  const int lane = idx & 63;
  const int r = idx >> 6;
  int nt, ks;
  if (order == 0) {
    ks = r % nks;
    nt = r / nks;
  } else {
    nt = r % nnt;
    ks = r / nnt;
  }
  const int n = 32 * nt + (lane & 31), k0 = 16 * ks + 8 * (lane >> 5);
  bf16x8 hi, lo;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float v = trans ? W[(long long)(k0 + j) * ldw + n] : W[(long long)n * ldw + k0 + j];
    hi[j] = (__bf16)v;
    lo[j] = (__bf16)(v - (float)hi[j]);
  }
  const long long u = (order == 0 ? ((long long)nt * nks + ks) : ((long long)ks * nnt + nt)) * 2;
  bf16x8* o = reinterpret_cast<bf16x8*>(out);
  o[u * 64 + lane] = hi;
  o[(u + 1) * 64 + lane] = lo;
}

// fp16 variant (ABI v15): the same units, each element of W' scaled by 2^8 and split into fp16 hi = fp16(256 w) and
// lo = fp16(256 w - hi) (22 bits; the scale keeps the lo term of ordinary weights out of fp16's subnormals) -- the B operand
// of ws_gemm_b2p with a_fmt = 2, whose A operand (scaled-fp16 d(gates)) then needs no conversion: v_mfma_f32_32x32x16_f16.
#define WS_PACK16_SCALE 256.f
__global__ void pack_w16_kernel(const float* __restrict__ W, int N, int K, long long ldw, int trans, int order,
                                _Float16* __restrict__ out) {
  const int nks = K / 16, nnt = N / 32;
  const int units = N * K / 8;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= units) return;
  const int lane = idx & 63;
  const int r = idx >> 6;
  int nt, ks;
  if (order == 0) {
    ks = r % nks;
    nt = r / nks;
  } else {
    nt = r % nnt;
    ks = r / nnt;
  }
  const int n = 32 * nt + (lane & 31), k0 = 16 * ks + 8 * (lane >> 5);
  f16x8 hi, lo;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float v = WS_PACK16_SCALE * (trans ? W[(long long)(k0 + j) * ldw + n] : W[(long long)n * ldw + k0 + j]);
    hi[j] = (_Float16)v;
    lo[j] = (_Float16)(v - (float)hi[j]);
  }
  const long long u = (order == 0 ? ((long long)nt * nks + ks) : ((long long)ks * nnt + nt)) * 2;
  f16x8* o = reinterpret_cast<f16x8*>(out);
  o[u * 64 + lane] = hi;
  o[(u + 1) * 64 + lane] = lo;
}

extern "C" int ws_pack_w_f16(const float* W, int N, int K, long long ldw, int trans, int order, float* out,
                             void* stream) {
  WS_REQUIRE(W && out && N > 0 && K > 0 && N % 32 == 0 && K % 16 == 0, "ws_pack_w_f16: N %% 32, K %% 16 (N=%d K=%d)", N, K);
  WS_REQUIRE(order == 0 || order == 1, "ws_pack_w_f16: order");
  hipLaunchKernelGGL(pack_w16_kernel, dim3((N * K / 8 + 255) / 256), dim3(256), 0, (hipStream_t)stream, W, N, K, ldw,
                     trans, order, reinterpret_cast<_Float16*>(out));
  return ws_check_launch("ws_pack_w_f16");
}

// fp16 hi + FP8 lo variant (ABI v20; N = 128, b2p order): the B operand of ws_gemm_b2p with a_fmt = 3 -- per stage of 64 k (24 KB):
// the sixteen fp16 hi fragments [ks 4][nt 4][lane] of ws_pack_w_f16, then per column tile nt ONE fragment of
// v_mfma_scale_f32_32x32x64_f8f6f4 with the e4m3 codes of the residuals 256 w - hi: a lane's 32 bytes = its 8 k of each of the
// stage's four k-steps in order (byte 8 i + j <-> k = 64 st + 16 i + 8 (lane >> 5) + j: the order the A operand is built in, in
// registers, from the four fp16 fragments), as two 16-byte pieces [nt][piece][lane]; one exponent per fragment (the largest code in
// [128, 256)): the E8M0 bytes of a stage's four fragments form one dword behind the last stage (at 24 KB * K / 64).
__global__ __launch_bounds__(256) void pack_w16f8_kernel(const float* __restrict__ W, int K, long long ldw, int trans,
                                                         unsigned char* __restrict__ out) {
  const int st = blockIdx.x, nt = threadIdx.x >> 6, lane = threadIdx.x & 63, n = 32 * nt + (lane & 31);
  unsigned char* sb = out + (long long)st * 24576;
  float res[32];
  float mx = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f16x8 hi;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = 64 * st + 16 * i + 8 * (lane >> 5) + j;
      const float v = WS_PACK16_SCALE * (trans ? W[(long long)k * ldw + n] : W[(long long)n * ldw + k]);
      hi[j] = (_Float16)v;
      res[8 * i + j] = v - (float)hi[j];
      mx = fmaxf(mx, fabsf(res[8 * i + j]));
    }
    reinterpret_cast<f16x8*>(sb)[(i * 4 + nt) * 64 + lane] = hi;
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  int E = mx > 0.f ? ((__float_as_int(mx) >> 23) & 255) - 127 - 7 : 0;
  E = max(E, -126);
  const float inv = __int_as_float((127 - E) << 23);
  unsigned int* o8 = reinterpret_cast<unsigned int*>(sb + 16384 + nt * 2048 + lane * 16);
#pragma unroll
  for (int j = 0; j < 32; j += 4) {
    int c = 0;
    c = __builtin_amdgcn_cvt_pk_fp8_f32(res[j] * inv, res[j + 1] * inv, c, false);
    c = __builtin_amdgcn_cvt_pk_fp8_f32(res[j + 2] * inv, res[j + 3] * inv, c, true);
    o8[(j >> 4) * 256 + ((j & 15) >> 2)] = (unsigned int)c;
  }
  if (lane == 0) (out + (long long)(K / 64) * 24576)[st * 4 + nt] = (unsigned char)(127 + E);
}

extern "C" int ws_pack_w_f16f8(const float* W, int N, int K, long long ldw, int trans, float* out, void* stream) {
  WS_REQUIRE(W && out && N == 128 && K > 0 && K % 64 == 0, "ws_pack_w_f16f8: N = 128, K %% 64 (N=%d K=%d)", N, K);
  hipLaunchKernelGGL(pack_w16f8_kernel, dim3(K / 64), dim3(256), 0, (hipStream_t)stream, W, K, ldw, trans,
                     reinterpret_cast<unsigned char*>(out));
  return ws_check_launch("ws_pack_w_f16f8");
}

extern "C" int ws_pack_w(const float* W, int N, int K, long long ldw, int trans, int order, float* out,
                         void* stream) {
  WS_REQUIRE(W && out && N > 0 && K > 0 && N % 32 == 0 && K % 16 == 0, "ws_pack_w: N %% 32, K %% 16 (N=%d K=%d)",
             N, K);
  WS_REQUIRE(order == 0 || order == 1, "ws_pack_w: order");
  hipLaunchKernelGGL(pack_w_kernel, dim3((N * K / 8 + 255) / 256), dim3(256), 0, (hipStream_t)stream, W, N, K, ldw,
                     trans, order, reinterpret_cast<__bf16*>(out));
  return ws_check_launch("ws_pack_w");
}

// ---------------------------------------------------------------------------------------------
// p2b: one wave = one block (32 positions); its activation fragments (K = 128: 8 k-steps x
// {hi, lo}) stay in registers for the whole sweep over N; weight tiles (2 x 32 columns = 32 KB of
// fragments per stage) go global -> registers -> LDS once per workgroup and are read by all 8
// waves.  Output orientation D[m = column][n = slot]: a lane holds 4 consecutive columns of its
// slot = one 16-byte BL cell, 32 lanes = 512 contiguous bytes.
// ---------------------------------------------------------------------------------------------
#define P2B_K 128
__global__ __launch_bounds__(512, 4) void gemm_p2b_kernel(const ws_gemm_p2b_args p) {
  __shared__ __attribute__((aligned(16))) u32x4 wl[2][2048];  // 2 stages x 32 KB
  if (p.run_if && *p.run_if == 0u) return;  // predicated fall-back launch (wesep_hip.h): uniform
  const int tid = threadIdx.x, lane = tid & 63, i = lane & 31, half = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nblk = ((p.sm.nseq + 31) / 32) * p.sm.L;
  const int b = blockIdx.x * 8 + w;
  const bool active = b < nblk;
  const int bb = active ? b : nblk - 1;
  bool valid;
  const long long pos = seq_pos(p.sm, bb, i, valid);

  // ---- activation fragments ---------------------------------------------------------------
  bf16x8 xh[8], xl[8];
  {
    const float* arow = p.A + pos * p.lda;
    float mean = 0.f, rstd = 1.f;
    if (p.stats) {
      const long long s = (pos / p.st_div1) * p.st_m1 + (pos % p.st_div2) * p.st_m2 + p.st_base;
      mean = p.stats[2 * s];
      rstd = p.stats[2 * s + 1];
    }
    float* eb = p.A_bl ? p.A_bl + (long long)bb * (32 * P2B_K) + i * 4 : nullptr;
    unsigned short* eb16 = p.A_bl16 ? reinterpret_cast<unsigned short*>(p.A_bl16) + (long long)bb * (32 * P2B_K) + i * 4 : nullptr;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int k0 = 16 * ks + 8 * half;
      f32x4 v0 = *reinterpret_cast<const f32x4*>(arow + k0);
      f32x4 v1 = *reinterpret_cast<const f32x4*>(arow + k0 + 4);
      if (p.stats) {
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(p.gamma + k0), g1 = *reinterpret_cast<const f32x4*>(p.gamma + k0 + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.beta + k0), b1 = *reinterpret_cast<const f32x4*>(p.beta + k0 + 4);
        v0 = (v0 - mean) * rstd * g0 + b0;
        v1 = (v1 - mean) * rstd * g1 + b1;
      }
      if (!valid) {
        v0 = f32x4{0.f, 0.f, 0.f, 0.f};
        v1 = v0;
      }
      const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
      split8(v, xh[ks], xl[ks]);
      if (eb && active) {  // the (normalised) operand itself, in BL(K) as split pairs (BLS): the fused recurrence
        u32x4 c0, c1;      // and the weight-gradient pass take their fragments from it without converting again
        pack8(xh[ks], xl[ks], c0, c1);
        *reinterpret_cast<u32x4*>(eb + (k0 / 4) * 128) = c0;
        *reinterpret_cast<u32x4*>(eb + (k0 / 4 + 1) * 128) = c1;
      }
      if (eb16 && active) {  // and as fp16 in BLH(K): the 2-byte A operand of ws_gemm_tnb (a_fmt = 1, ABI v16)
        const f16x2 h0 = {(_Float16)v[0], (_Float16)v[1]}, h1 = {(_Float16)v[2], (_Float16)v[3]};
        const f16x2 h2 = {(_Float16)v[4], (_Float16)v[5]}, h3 = {(_Float16)v[6], (_Float16)v[7]};
        *reinterpret_cast<u32x2*>(eb16 + (k0 / 4) * 128) = u32x2{__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1)};
        *reinterpret_cast<u32x2*>(eb16 + (k0 / 4 + 1) * 128) = u32x2{__builtin_bit_cast(unsigned, h2), __builtin_bit_cast(unsigned, h3)};
      }
    }
  }

  // ---- sweep over N in stages of 64 columns ---------------------------------------------------
  const int nstage = p.N / 64;
  if (nstage == 0) return;  // N = 0: the call only relays the (normalised) operand into BL(K)
  const u32x4* wsrc = reinterpret_cast<const u32x4*>(p.Wpack);  // 2048 units of 16 B per stage
  u32x4 wreg[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) wreg[q] = wsrc[tid + 512 * q];
#pragma unroll
  for (int q = 0; q < 4; ++q) wl[0][tid + 512 * q] = wreg[q];
  __syncthreads();
  float* cblk = p.C + (long long)bb * 32 * p.N + i * 4;
  float cmax = 0.f;  // max |C| of this lane (p.amax: the scale source of WS_GATES_H2F; fmaxf drops NaN)
  for (int st = 0; st < nstage; ++st) {
    const int cur = st & 1;
    if (st + 1 < nstage) {
#pragma unroll
      for (int q = 0; q < 4; ++q) wreg[q] = wsrc[(long long)(st + 1) * 2048 + tid + 512 * q];
    }
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
      const u32x4* wt = &wl[cur][t2 * 1024 + lane];  // [ks][part][lane]
      f32x16 acc;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const bf16x8 ah = __builtin_bit_cast(bf16x8, wt[(ks * 2) * 64]);
        const bf16x8 al = __builtin_bit_cast(bf16x8, wt[(ks * 2 + 1) * 64]);
        if (ks == 0) {
          const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          acc = mfma32(ah, xh[ks], zero);
        } else {
          acc = mfma32(ah, xh[ks], acc);
        }
        acc = mfma32(al, xh[ks], acc);
        acc = mfma32(ah, xl[ks], acc);
      }
      if (active) {
        const int n0 = st * 64 + t2 * 32;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int n = n0 + 8 * j + 4 * half;  // 4 consecutive columns = BL cell (n/4, i)
          f32x4 v = {acc[4 * j], acc[4 * j + 1], acc[4 * j + 2], acc[4 * j + 3]};
          if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
          if (!valid) v = f32x4{0.f, 0.f, 0.f, 0.f};
          *reinterpret_cast<f32x4*>(cblk + (n / 4) * 128) = v;
          cmax = fmaxf(fmaxf(cmax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
        }
      }
    }
    if (st + 1 < nstage) {
#pragma unroll
      for (int q = 0; q < 4; ++q) wl[cur ^ 1][tid + 512 * q] = wreg[q];
    }
    __syncthreads();
  }
  if (p.amax) {  // one atomic per wave: non-negative floats order like their bit patterns
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cmax = fmaxf(cmax, __shfl_xor(cmax, o, 64));
    if (lane == 0) atomicMax(p.amax, __float_as_uint(cmax));
  }
}

extern "C" int ws_gemm_p2b(const ws_gemm_p2b_args* a, void* stream) {
  WS_REQUIRE(a && a->A && ((a->Wpack && a->C) || (a->N == 0 && (a->A_bl || a->A_bl16))), "ws_gemm_p2b: null pointer");
  WS_REQUIRE(a->K == P2B_K, "ws_gemm_p2b: K must be %d (got %d)", P2B_K, a->K);
  WS_REQUIRE(a->N >= 0 && a->N % 64 == 0, "ws_gemm_p2b: N %% 64 (N=%d)", a->N);
  WS_REQUIRE(a->lda >= a->K && a->lda % 4 == 0, "ws_gemm_p2b: lda");
  WS_REQUIRE(a->sm.nseq > 0 && a->sm.L > 0 && a->sm.sq_div > 0, "ws_gemm_p2b: bad sequence map");
  WS_REQUIRE(!a->stats || (a->gamma && a->beta && a->st_div1 > 0 && a->st_div2 > 0), "ws_gemm_p2b: norm args");
  const int nblk = ((a->sm.nseq + 31) / 32) * a->sm.L;
  hipStream_t s = (hipStream_t)stream;
  const bool timed = a->run_if == nullptr;  // a predicated fall-back launch is normally empty: not a sample of this kind
  if (timed) ws_prof_begin(WS_PROF_GEMM_NT, s);
  hipLaunchKernelGGL(gemm_p2b_kernel, dim3((nblk + 7) / 8), dim3(512), 0, s, *a);
  if (timed) ws_prof_end(WS_PROF_GEMM_NT, s);
  return ws_check_launch("ws_gemm_p2b");
}

// ---------------------------------------------------------------------------------------------
// b2p: N = 128.  One wave = one block; activations come straight from BL into A-operand
// fragments (lane = slot, 8 consecutive k = two 16-byte cells, each 512 contiguous bytes per 32
// lanes) and are split in registers; weights (4 k-steps x 4 column tiles x {hi, lo} = 32 KB per
// stage) are shared through LDS.  Output orientation D[m = slot][n = column]: a 32-lane group
// writes one full 128-byte row segment per register.
// ---------------------------------------------------------------------------------------------
// A16 (a_fmt = 1, ABI v15): A holds bf16 elements in BLH(K) -- d(gates) of WS_GATES_H2: a lane's 8 consecutive k are two
// 8-byte cells = the hi fragment itself; no lo term, two MFMAs per product instead of three, half the A bytes.
// A16 = 2 (a_fmt = 2): A holds fp16 elements scaled by S = ws_dgates_scale(*p.amax) -- d(gates) of WS_GATES_H2F -- and Wpack
// is a ws_pack_w_f16 pack (fp16 hi / lo of 256 w): the A cells ARE the MFMA fragments (no conversion) and a product is
// a (w_hi + w_lo) on v_mfma_f32_32x32x16_f16 -- the operand's 11 bits times the weight's 22; the epilogue multiplies by
// 1 / (256 S) (a power of two: exact).
// A16 = 3 (a_fmt = 3, ABI v20): a_fmt 2 with the lo term on v_mfma_scale_f32_32x32x64_f8f6f4 -- Wpack from ws_pack_w_f16f8 (24 KB
// per stage through LDS instead of 32), the A operand of the term = e4m3 of the stage's four fp16 fragments / 256, converted in
// registers: per stage and column tile four fp16 MFMAs + one FP8 MFMA (K = 64, twice the rate) instead of eight.
// (launch bounds: the SECOND number is hipcc's minimum of waves per SIMD, not workgroups per CU.  With "2" (rounds 1-5) the kernel took
//  129 .. 168 registers -- three waves per SIMD, i.e. ONE 512-thread workgroup per CU where 66 KB of LDS allow two.  Round 6: "4" =
//  128 registers; a_fmt 2 fits as it was (0.69 -> 0.58 ms per launch alone, 4.5 ms per training step: profiles/r06_c28_*), the
//  other formats once the weight stages go to LDS without a register stop and a_fmt 0's activations are requested half a stage
//  ahead -- no spills in any of the four.)
template <int A16>
__global__ __launch_bounds__(512, 4) void gemm_b2p_kernel(const ws_gemm_b2p_args p) {
  __shared__ __attribute__((aligned(16))) u32x4 wl[2][2048];
  __shared__ long long posl[8][32];
  const int tid = threadIdx.x, lane = tid & 63, i = lane & 31, half = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nblk = ((p.sm.nseq + 31) / 32) * p.sm.L;
  const int b = blockIdx.x * 8 + w;
  const bool active = b < nblk;
  const int bb = active ? b : nblk - 1;
  {
    bool valid;
    const long long pos = seq_pos(p.sm, bb, i, valid);
    if (half == 0) posl[w][i] = valid ? pos : -1;
  }
  const int K = p.K, nstage = K / 64;
  // A-operand source: cell (quad, slot) of block bb; lane reads quads 4ks + 2half, +1 (cell = 16 B, A16: 8 B; `ab` in
  // units of half a cell-element pair: floats for BLS, 2-byte elements viewed through the same index formula for A16)
  typedef typename std::conditional<A16 != 0, u32x2, f32x4>::type acell;
  typedef typename std::conditional<A16 != 0, unsigned short, float>::type aelem;
  constexpr int SU = A16 == 3 ? 1536 : 2048, NQ = SU / 512;    // 16-byte units of a weight stage; per thread
  const int* wsc = reinterpret_cast<const int*>(reinterpret_cast<const unsigned char*>(p.Wpack) + (long long)nstage * 24576);
  int sc_n = A16 == 3 ? wsc[0] : 0;                            // A16 = 3: the four fragment exponents of the next stage
  const float inv_s = A16 >= 2 ? ws_dgates_scale_inv(*p.amax) * (1.f / WS_PACK16_SCALE) : 1.f;
  const aelem* ab = reinterpret_cast<const aelem*>(p.A) + (long long)bb * 32 * K + i * 4 + 2 * half * 128;
  // weight stages: global -> LDS without a register stop (buffer_load ... lds: lane l of a wave lands at base + 16 l) -- the
  // 12 / 16 registers a stage used to wait in are what keeps the other formats from a second workgroup per CU
  const __amdgpu_buffer_rsrc_t wsr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.Wpack), 0, (unsigned)nstage * SU * 16u,
                                                                        0x00020000);
  auto wstage = [&](int st, int buf) {   // this wave's share of stage st: units w * 64 * NQ + 64 q + lane
#pragma unroll
    for (int q = 0; q < NQ; ++q)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wsr, (__attribute__((address_space(3))) void*)&wl[buf][(w * NQ + q) * 64], 16, lane * 16,
                                               (st * SU + (w * NQ + q) * 64) * 16, 0, 0);
  };
  // activations: requested one stage ahead -- the 16-byte cells of a_fmt 0 half a stage ahead (two k-steps: 16 registers in flight
  // instead of 32, with the stage's 32 the difference between one and two workgroups per CU)
  constexpr int HS = A16 == 0 ? 2 : 1, KH = 4 / HS;   // halves per stage; k-steps per half
  acell an[2 * KH];
  auto load_a = [&](int st, int hs) {
    const aelem* a2 = ab + ((long long)st * 16 + hs * KH * 4) * 128;  // 16 quads per stage
#pragma unroll
    for (int ks = 0; ks < KH; ++ks) {
      an[2 * ks] = *reinterpret_cast<const acell*>(a2 + (4 * ks) * 128);
      an[2 * ks + 1] = *reinterpret_cast<const acell*>(a2 + (4 * ks + 1) * 128);
    }
  };
  wstage(0, 0);
  load_a(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  f32x16 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  for (int st = 0; st < nstage; ++st) {
    const int cur = st & 1;
    const int sc = __builtin_amdgcn_readfirstlane(sc_n);
    v8i a8;   // A16 = 3: e4m3 of the stage's A fragments / 256
#pragma unroll
    for (int hs = 0; hs < HS; ++hs) {
    acell ac[2 * KH];
#pragma unroll
    for (int q = 0; q < 2 * KH; ++q) ac[q] = an[q];
    if (hs == 0 && st + 1 < nstage) {
      wstage(st + 1, cur ^ 1);     // (the other buffer: last read one stage ago, behind that stage's barrier)
      if constexpr (A16 == 3) sc_n = wsc[st + 1];
    }
    if (hs + 1 < HS) load_a(st, hs + 1);
    else if (st + 1 < nstage) load_a(st + 1, 0);
#pragma unroll
    for (int ksl = 0; ksl < KH; ++ksl) {
      const int ks = hs * KH + ksl;
      bf16x8 ah, al;  // A arrives as split pairs (BLS): h from the recurrences, d(gates) from BPTT -- or as bf16 (A16)
      if constexpr (A16 == 1) {
        const u32x2 c0 = __builtin_bit_cast(u32x2, ac[2 * ksl]), c1 = __builtin_bit_cast(u32x2, ac[2 * ksl + 1]);
        ah = __builtin_bit_cast(bf16x8, u32x4{c0[0], c0[1], c1[0], c1[1]});
      } else if constexpr (A16 == 3) {
        const u32x2 c0 = __builtin_bit_cast(u32x2, ac[2 * ksl]), c1 = __builtin_bit_cast(u32x2, ac[2 * ksl + 1]);
        const f16x8 a16 = __builtin_bit_cast(f16x8, u32x4{c0[0], c0[1], c1[0], c1[1]});
        typedef short s16x2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {     // (|fp16| / 256 < 256: inside e4m3, which has no infinity)
          s16x2 c = {0, 0};
          c = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(c, f16x2{a16[4 * h2], a16[4 * h2 + 1]}, 256.f, false);
          c = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(c, f16x2{a16[4 * h2 + 2], a16[4 * h2 + 3]}, 256.f, true);
          a8[2 * ks + h2] = __builtin_bit_cast(int, c);
        }
        const u32x4* wt = &wl[cur][ks * 256 + lane];  // hi fragments [ks][nt][lane]
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
          acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a16, __builtin_bit_cast(f16x8, wt[nt * 64]), acc[nt], 0, 0, 0);
        if (ks == 3) {
          const u32x4* w8 = &wl[cur][1024 + lane];    // FP8 fragments [nt][piece][lane]
          auto frag = [&](int nt) {
            return __builtin_shufflevector(__builtin_bit_cast(i32x4, w8[(2 * nt) * 64]), __builtin_bit_cast(i32x4, w8[(2 * nt + 1) * 64]),
                                           0, 1, 2, 3, 4, 5, 6, 7);
          };
          acc[0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, frag(0), acc[0], 0, 0, 0, 127 + 8, 0, sc);
          acc[1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, frag(1), acc[1], 0, 0, 0, 127 + 8, 1, sc);
          acc[2] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, frag(2), acc[2], 0, 0, 0, 127 + 8, 2, sc);
          acc[3] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, frag(3), acc[3], 0, 0, 0, 127 + 8, 3, sc);
        }
        continue;
      } else if constexpr (A16 == 2) {
        const u32x2 c0 = __builtin_bit_cast(u32x2, ac[2 * ksl]), c1 = __builtin_bit_cast(u32x2, ac[2 * ksl + 1]);
        const f16x8 a16 = __builtin_bit_cast(f16x8, u32x4{c0[0], c0[1], c1[0], c1[1]});
        const u32x4* wt = &wl[cur][ks * 512 + lane];  // [nt][part][lane]
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a16, __builtin_bit_cast(f16x8, wt[(nt * 2) * 64]), acc[nt], 0, 0, 0);
          acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a16, __builtin_bit_cast(f16x8, wt[(nt * 2 + 1) * 64]), acc[nt], 0, 0, 0);
        }
        continue;
      } else {
        unpack8(__builtin_bit_cast(u32x4, ac[2 * ksl]), __builtin_bit_cast(u32x4, ac[2 * ksl + 1]), ah, al);
        if (p.a16_out && active) {  // the operand once more as fp16 in BLH(K) (ABI v16): hi + lo is the fp32 value
          unsigned short* o16 = reinterpret_cast<unsigned short*>(p.a16_out) + (long long)bb * 32 * K + i * 4 +
                                (long long)(st * 16 + 4 * ks + 2 * half) * 128;
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            const u32x4 cell = __builtin_bit_cast(u32x4, ac[2 * ksl + c]);
            _Float16 h[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
              h[j] = (_Float16)(__uint_as_float(cell[j] & 0xffff0000u) + __uint_as_float(cell[j] << 16));
            const f16x2 p0 = {h[0], h[1]}, p1 = {h[2], h[3]};
            *reinterpret_cast<u32x2*>(o16 + c * 128) = u32x2{__builtin_bit_cast(unsigned, p0), __builtin_bit_cast(unsigned, p1)};
          }
        }
      }
      const u32x4* wt = &wl[cur][ks * 512 + lane];  // [nt][part][lane]
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const bf16x8 bh = __builtin_bit_cast(bf16x8, wt[(nt * 2) * 64]);
        const bf16x8 bl = __builtin_bit_cast(bf16x8, wt[(nt * 2 + 1) * 64]);
        acc[nt] = mfma32(ah, bh, acc[nt]);
        if constexpr (A16 != 1) acc[nt] = mfma32(al, bh, acc[nt]);
        acc[nt] = mfma32(ah, bl, acc[nt]);
      }
    }
    }   // hs
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next stage's weights have landed (and its activations: needed anyway)
    __syncthreads();
  }

  // Epilogue through LDS (the weight stages are done: their 64 KB are reused, 8 KB per wave): the accumulators hold one
  // COLUMN per lane; written out that way a row costs 64 four-byte accesses per lane, each behind its own validity
  // branch, and the residual loads could not be moved above the stores (C may alias R) -- 64 exposed memory round trips
  // per wave, 0.4 of the projection's 0.59 ms.  Staged as [32 rows][64 columns] per pass, a lane moves 16-byte pieces of
  // whole rows: all residual loads of a pass are issued before its first store.
  float* stg = reinterpret_cast<float*>(&wl[0][0]) + w * 2048;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int nt = 2 * pass + t;
      const float bv = p.bias ? p.bias[nt * 32 + i] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = (r & 3) + 8 * (r >> 2) + 4 * half;
        stg[m * 64 + t * 32 + i] = (A16 >= 2 ? acc[nt][r] * inv_s : acc[nt][r]) + bv;
      }
    }
    __syncthreads();  // (uniform: every wave runs both passes; only the wave's own 8 KB are exchanged)
    const int c4 = lane & 15, r0 = lane >> 4;  // 16 lanes x 16 B = the 64 columns of one row; 4 rows per access
    f32x4 v[8];
    long long off[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int m = r0 + 4 * k;
      const long long pos = posl[w][m];
      off[k] = (active && pos >= 0) ? pos * p.ldc + pass * 64 + 4 * c4 : -1;
      v[k] = *reinterpret_cast<const f32x4*>(stg + m * 64 + 4 * c4);
    }
    if (p.R) {
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (off[k] >= 0) v[k] += *reinterpret_cast<const f32x4*>(p.R + off[k]);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (off[k] >= 0) *reinterpret_cast<f32x4*>(p.C + off[k]) = v[k];
    __syncthreads();  // staging area free for the next pass
  }
}

extern "C" int ws_gemm_b2p(const ws_gemm_b2p_args* a, void* stream) {
  WS_REQUIRE(a && a->A && a->Wpack && a->C, "ws_gemm_b2p: null pointer");
  WS_REQUIRE(a->N == 128, "ws_gemm_b2p: N must be 128 (got %d)", a->N);
  WS_REQUIRE(a->a_fmt >= 0 && a->a_fmt <= 3 && (a->a_fmt < 2 || a->amax), "ws_gemm_b2p: a_fmt %d (2 / 3 need amax)", a->a_fmt);
  WS_REQUIRE(a->a_fmt != 3 || !a->a16_out, "ws_gemm_b2p: a16_out goes with a_fmt 0");
  WS_REQUIRE(a->K > 0 && a->K % 64 == 0, "ws_gemm_b2p: K %% 64 (K=%d)", a->K);
  WS_REQUIRE(a->ldc >= a->N && a->ldc % 4 == 0, "ws_gemm_b2p: ldc >= N and ldc %% 4 == 0 (16-byte row pieces)");
  WS_REQUIRE(a->sm.nseq > 0 && a->sm.L > 0 && a->sm.sq_div > 0, "ws_gemm_b2p: bad sequence map");
  const int nblk = ((a->sm.nseq + 31) / 32) * a->sm.L;
  hipStream_t s = (hipStream_t)stream;
  ws_prof_begin(WS_PROF_GEMM_NT, s);
  if (a->a_fmt == 3)
    hipLaunchKernelGGL(gemm_b2p_kernel<3>, dim3((nblk + 7) / 8), dim3(512), 0, s, *a);
  else if (a->a_fmt == 2)
    hipLaunchKernelGGL(gemm_b2p_kernel<2>, dim3((nblk + 7) / 8), dim3(512), 0, s, *a);
  else if (a->a_fmt == 1)
    hipLaunchKernelGGL(gemm_b2p_kernel<1>, dim3((nblk + 7) / 8), dim3(512), 0, s, *a);
  else
    hipLaunchKernelGGL(gemm_b2p_kernel<0>, dim3((nblk + 7) / 8), dim3(512), 0, s, *a);
  ws_prof_end(WS_PROF_GEMM_NT, s);
  return ws_check_launch("ws_gemm_b2p");
}

// ---------------------------------------------------------------------------------------------
// tnb: out[g][a] = sum over blocks b in the split, slots i of G[(b,i)][g] * A[(b + shift,i)][a].
// The contraction index is the slot.  A [32 slots x 128 columns] sub-block of BL is one contiguous
// 16 KB: the workgroup reads it with fully coalesced 64-byte-per-thread loads (a thread owns one
// "group" = (quad, 4 consecutive slots) = a 4x4 register block), transposes that block in registers
// for free and writes each column's 4 consecutive slots as one 8-byte bf16 group into a
// [column][slot] LDS image (row stride 80 B: conflict-free 16-byte fragment reads).
// One workgroup (8 waves, 2 x 4) owns 128 G columns x ALL A columns (TA tiles of 128; TA = 3 for
// [dW_ih | dW_hh], 1 otherwise), so G -- the big operand -- crosses L2/HBM exactly once.  Global
// loads run TWO blocks ahead of the MFMAs (the block loop is HBM-latency-bound otherwise).
// ---------------------------------------------------------------------------------------------
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
#define TB_LD 40

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

template <int TA, bool ASUM>
__global__ __launch_bounds__(512, 2) void gemm_tnb_kernel(const ws_gemm_tnb_args p) {
  constexpr int NCOL = 128 * (1 + TA);   // LDS columns: G tile, then the A tiles
  constexpr int PLANE = NCOL * TB_LD;    // bf16 elements per part
  constexpr int NG = (1 + TA) / 2;       // groups per thread (512 threads, 256 groups per 128 columns)
  constexpr int TN = TA == 1 ? 1 : TA;   // 32-column MFMA tiles per wave along A (wave owns 32*TA columns)
  // two LDS images [buf][hi | lo][column][slot]: block b+1 is converted and written while block b's
  // fragments are read, one barrier per block (TA = 3: 2 x 80 KB = the whole LDS of the CU)
  // two distinct objects (not one [2][..] array) so alias analysis can move image-A reads past image-B writes
  __shared__ __attribute__((aligned(16))) __bf16 ldsA[2 * PLANE];
  __shared__ __attribute__((aligned(16))) __bf16 ldsB[2 * PLANE];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 2, wn = w & 3;
  // x = split: workgroups that share the A operand (same split, different G tile) differ by a
  // multiple of gridDim.x (a multiple of 8 when nsplit is) in dispatch order -> same XCD -> one L2
  const int gt = blockIdx.y, split = blockIdx.x;
  const int L = p.L;
  const int b_begin = split * p.blocks_per_split;
  const int b_end = min(p.nblk, b_begin + p.blocks_per_split);

  // this thread's groups: g = tid + 512*r -> part (0 = G, 1.. = A tile), quad, slot group
  const float* gbase[NG];
  long long gstride[NG];
  int gshift[NG], gcol[NG];
#pragma unroll
  for (int r = 0; r < NG; ++r) {
    const int g = tid + 512 * r;
    const int part = g >> 8, quad = (g >> 3) & 31, sg = g & 7;
    gcol[r] = part * 128 + 4 * quad;
    if (part == 0) {
      gbase[r] = p.G + (long long)((p.g_off + gt * 128) / 4 + quad) * 128 + sg * 16;
      gstride[r] = 32LL * p.g_width;
      gshift[r] = 0;
    } else {
      const int c0 = (part - 1) * 128;  // first column of this A tile in Acat
      const bool s1 = c0 >= p.a0_cols;
      const float* base = s1 ? p.A1 : p.A0;
      const int off = s1 ? p.a1_off + c0 - p.a0_cols : p.a0_off + c0;
      gbase[r] = base + (long long)(off / 4 + quad) * 128 + sg * 16;
      gstride[r] = 32LL * (s1 ? p.a1_width : p.a0_width);
      gshift[r] = s1 ? p.a1_shift : p.a0_shift;
    }
  }
  const int sg = tid & 7;

  f32x4 rq[2][NG][4];  // two blocks in flight
  bool use[2][NG];
  auto load_block = [&](int b, int slot) {
    const int tile = b / L, step = b - tile * L;
#pragma unroll
    for (int r = 0; r < NG; ++r) {
      const int sa = step + gshift[r];
      use[slot][r] = sa >= 0 && sa < L;
      const float* src = gbase[r] + (long long)(use[slot][r] ? b + gshift[r] : b) * gstride[r];
      // (non-temporal loads of G and / or A measured neutral, alone and beside a recurrence: tools/overlap_probe.py)
#pragma unroll
      for (int j = 0; j < 4; ++j) rq[slot][r][j] = *reinterpret_cast<const f32x4*>(src + 4 * j);
    }
  };
  float csum[NG][4];
#pragma unroll
  for (int r = 0; r < NG; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) csum[r][c] = 0.f;
  // one piece = one (group r, column c) of a block: 4 slots -> hi/lo bf16x4 -> LDS; pc = 4*r + c.
  // Both operands arrive as split pairs (BLS): the 4x4 register transpose and the hi / lo separation are the same
  // four v_perm_b32; the column sums (bias gradients) are v_dot2c_f32_bf16 with (1, 1).  Only groups r >= 1 can be a
  // shifted operand (the launcher refuses a0_shift != 0), so only they are masked at the sequence ends.
  auto store_piece = [&](int slot, __bf16* lds, unsigned live, int pc) {  // live = 0: tail iteration, sums untouched
    const int r = pc >> 2, c = pc & 3;
    unsigned e[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float f = rq[slot][r][j][c];  // (bit_cast of a vector-element lvalue reads element 0 with this compiler)
      e[j] = __float_as_uint(f);
    }
    if (r > 0) {
      const unsigned m = use[slot][r] ? 0xffffffffu : 0u;  // uniform
#pragma unroll
      for (int j = 0; j < 4; ++j) e[j] &= m;
    }
    if (r == 0 || ASUM) {
      const bf16x2 ones = __builtin_bit_cast(bf16x2, live);  // 0x3f803f80 = (1, 1)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        csum[r][c] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, e[j]), ones, csum[r][c], false);
    }
    uint2 hi, lo;
    hi.x = __builtin_amdgcn_perm(e[1], e[0], WS_SEL_HI16);
    hi.y = __builtin_amdgcn_perm(e[3], e[2], WS_SEL_HI16);
    lo.x = __builtin_amdgcn_perm(e[1], e[0], WS_SEL_LO16);
    lo.y = __builtin_amdgcn_perm(e[3], e[2], WS_SEL_LO16);
    const int o = (gcol[r] + c) * TB_LD + 4 * sg;
    *reinterpret_cast<uint2*>(lds + o) = hi;
    *reinterpret_cast<uint2*>(lds + PLANE + o) = lo;
  };
  auto store_block = [&](int slot, __bf16* lds, unsigned live) {
#pragma unroll
    for (int pc = 0; pc < 4 * NG; ++pc) store_piece(slot, lds, live, pc);
  };

  f32x16 acc[2][TN];
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int f = 0; f < TN; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[e][f][r] = 0.f;

  // The block loop body is ONE basic block (tail iterations reload / reconvert the last block instead of
  // branching), so the conversion of block b+1 (VALU + LDS writes) can be interleaved with the MFMAs
  // of block b by the scheduler hints below -- both waves of a SIMD are in phase after each barrier,
  // so without it the two phases simply add up.
  const int nb = b_end - b_begin;
  if (nb > 0) {
    load_block(b_begin, 0);
    load_block(min(b_begin + 1, b_end - 1), 1);
    store_block(0, ldsA, 0x3f803f80u);
  }
  __syncthreads();
  for (int ib = 0; ib < nb; ib += 2) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int i = ib + s;
      if (i < nb) {  // uniform; only the odd tail skips
        // registers of slot s (block i) were consumed by the previous store: refill them 2 blocks ahead
        load_block(b_begin + min(i + 2, nb - 1), s);
        __builtin_amdgcn_sched_barrier(0);
        // block i+1 (register slot s^1) -> the other LDS image, one piece per MFMA pair of this
        // block; scheduling fences keep each piece beside its pair (a wave issues in order, so VALU
        // work only overlaps the matrix pipe when it sits between MFMAs in program order)
        const __bf16* lds = s ? ldsB : ldsA;
        __bf16* ldsw = s ? ldsA : ldsB;
        const unsigned live = i + 1 < nb ? 0x3f803f80u : 0u;
        constexpr int NSUB = 6 * TN, NPC = 4 * NG, EVERY = NSUB / NPC;
        auto lda = [&](int ks, int f, bf16x8& h, bf16x8& l) {
          const int ra = (128 + wn * 32 * TN + f * 32 + l31) * TB_LD + ks + 8 * half;
          h = *reinterpret_cast<const bf16x8*>(lds + ra);
          l = *reinterpret_cast<const bf16x8*>(lds + PLANE + ra);
        };
        bf16x8 ah, al, ahn, aln;
        lda(0, 0, ah, al);
        int sub = 0;
#pragma unroll
        for (int ks = 0; ks < 32; ks += 16) {
          bf16x8 gh[2], gl[2];
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int rg = (wm * 64 + e * 32 + l31) * TB_LD + ks + 8 * half;
            gh[e] = *reinterpret_cast<const bf16x8*>(lds + rg);
            gl[e] = *reinterpret_cast<const bf16x8*>(lds + PLANE + rg);
          }
#pragma unroll
          for (int f = 0; f < TN; ++f) {
            const bool last = ks == 16 && f == TN - 1;
#pragma unroll
            for (int term = 0; term < 3; ++term, ++sub) {
              if (term == 0 && !last) lda(f + 1 < TN ? ks : 16, f + 1 < TN ? f + 1 : 0, ahn, aln);
#pragma unroll
              for (int e = 0; e < 2; ++e)
                acc[e][f] = mfma32(term == 1 ? gl[e] : gh[e], term == 2 ? al : ah, acc[e][f]);
              if (sub % EVERY == 0 && sub / EVERY < NPC) store_piece(s ^ 1, ldsw, live, sub / EVERY);
              __builtin_amdgcn_sched_barrier(0);
            }
            ah = ahn; al = aln;
          }
        }
        __syncthreads();  // image s fully read, image s^1 fully written
      }
    }
  }

  const int ncols = p.a0_cols + p.a1_cols;
  float* out = p.slab + (long long)split * p.slab_stride;
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int f = 0; f < TN; ++f) {
      const int acol = wn * 32 * TN + f * 32 + l31;
      if (acol < ncols) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int grow = gt * 128 + wm * 64 + e * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          out[(long long)grow * ncols + acol] = acc[e][f][r];
        }
      }
    }
  // column sums: of G (bias gradient) and, on request, of the A columns (gt == 0 only)
#pragma unroll
  for (int r = 0; r < NG; ++r) {
    const int part = (tid + 512 * r) >> 8;
    float* dst = part == 0 ? p.bslab : (gt == 0 && (r == 0 || ASUM) ? p.aslab : nullptr);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float t = csum[r][c];
      t += __shfl_xor(t, 1, 64);
      t += __shfl_xor(t, 2, 64);
      t += __shfl_xor(t, 4, 64);
      if (dst && sg == 0) {
        if (part == 0)
          dst[(long long)split * p.bslab_stride + gt * 128 + gcol[r] + c] = t;
        else
          dst[(long long)split * p.aslab_stride + gcol[r] - 128 + c] = t;
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------
// tnb with G as bf16 elements in BLH (g_fmt = 1, ABI v15: d(gates) of WS_GATES_H2), A columns = 384 ([xn | h]).
// Same tiling, LDS images and MFMA schedule as gemm_tnb_kernel<3, .>; what changes:
//   * the G tile of a block is 8 KB instead of 16: every thread loads ONE 16-byte piece (quad tid >> 4, slots
//     2 (tid & 15), + 1), separates its four columns with four v_perm_b32 and writes each as one bf16 pair; there is no lo
//     plane of G, so a product is two MFMAs (G_hi A_hi + G_hi A_lo) instead of three;
//   * the three A tiles (768 groups of 64 B) are spread as one whole group per thread (tiles 0 and 1) plus one half group
//     (2 slots, tile 2): 112 bytes and 7 loads per thread and block instead of 128 / 8.
// ---------------------------------------------------------------------------------------------
// GFMT = 2 (g_fmt = 2): G holds fp16 elements scaled by S = ws_dgates_scale(*p.amax) -- d(gates) of WS_GATES_H2F: every value
// is split into its bf16 hi + lo terms in registers (exact: 11 = 8 + 3 bits), both planes of the G image are written and a
// product is the three split-pair MFMAs again; slab and bslab leave the kernel multiplied by 1 / S (exact).
// GD: blocks of G in flight per thread (2 or 4).  G is the operand that comes from HBM (the A tiles are re-read by the eight
// workgroups of a split: L2), and a wave's loads return in order: the A loads are issued FIRST, so they do not wait behind
// the HBM round trip of G, and with GD = 4 the 16 bytes of G per thread and block are requested four blocks ahead (4 more
// registers per block in flight) -- the block loop is bound by the round trip of its prefetch (profiles/r03_tnb_experiments.md).
template <bool ASUM, int GFMT, int GD, int AF = 0>
__global__ __launch_bounds__(512, 2) void gemm_tnb16_kernel(const ws_gemm_tnb_args p) {
  // AF = 1 (a_fmt 1, ABI v16; with GFMT 3 only): A0 / A1 hold fp16 elements in BLH -- 8-byte cells, ONE LDS plane, ONE MFMA
  // per product; a workgroup loads 32 KB per block instead of 56
  static_assert(AF == 0 || (GFMT == 3 && !ASUM), "fp16 A operand: scaled-fp16 G on the fp16 instruction, no column sums of A");
  constexpr int TA = 3, TN = 3;
  // GFMT 3 = the scaled-fp16 G of g_fmt 2 on v_mfma_f32_32x32x16_f16: G needs no split (its 11 bits ARE an fp16), the A
  // operand's bf16 hi / lo terms convert exactly to fp16 after a power-of-two lift (x 2^6: one packed exponent add per
  // BLS word; keeps lo terms of |x| >= 5e-4 out of the fp16 denormals, |x| < 1023 finite) -> G A_hi + G A_lo, TWO MFMAs
  // per product instead of three (G_hi A_hi + G_hi A_lo + G_lo A_hi on the bf16 instruction)
  constexpr bool F16 = GFMT == 3;
  constexpr int NTERM = AF ? 1 : (GFMT == 2 ? 3 : 2);
  const float inv_s = GFMT >= 2 ? ws_dgates_scale_inv(*p.amax) : 1.f;
  const float inv_out = (F16 && !AF) ? inv_s * 0.015625f : inv_s;   // (the 2^6 lift belongs to the BLS -> fp16 conversion)
  constexpr int NCOL = 128 * (1 + TA);
  constexpr int PLANE = NCOL * TB_LD;
  __shared__ __attribute__((aligned(16))) __bf16 ldsA[(AF ? 1 : 2) * PLANE];   // (AF: no lo plane at all -- 80 KB per workgroup)
  __shared__ __attribute__((aligned(16))) __bf16 ldsB[(AF ? 1 : 2) * PLANE];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 2, wn = w & 3;
  const int gt = blockIdx.y, split = blockIdx.x;
  const int L = p.L;
  const int b_begin = split * p.blocks_per_split;
  const int b_end = min(p.nblk, b_begin + p.blocks_per_split);

  // ---- G piece: quad qg of the tile, slots 2 sp and 2 sp + 1 (two 8-byte cells, adjacent) ----
  const int qg = tid >> 4, sp = tid & 15;
  const unsigned short* gsrc = reinterpret_cast<const unsigned short*>(p.G) +
                               ((long long)((p.g_off + gt * 128) / 4 + qg) * 32 + 2 * sp) * 4;
  const long long gstep = 32LL * p.g_width;  // 2-byte elements per block
  // ---- A pieces: r = 0: whole group tid of tiles 0 / 1; r = 1: half group (tid >> 1, slots 2 (tid & 1) ..) of tile 2 ----
  const float* abase[2];
  long long astride[2];
  int ashift[2], acol[2], aslot[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int g = r == 0 ? tid : 512 + (tid >> 1);
    const int tile = g >> 8, quad = (g >> 3) & 31, sg = g & 7;
    const int c0 = tile * 128;  // first column of this A tile in Acat
    const bool s1 = c0 >= p.a0_cols;
    const float* base = s1 ? p.A1 : p.A0;
    const int off = s1 ? p.a1_off + c0 - p.a0_cols : p.a0_off + c0;
    aslot[r] = 4 * sg + (r == 1 ? 2 * (tid & 1) : 0);
    // (AF: the same index formula in 2-byte elements; abase then counts halves through a float pointer's address)
    abase[r] = AF ? reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(base) +
                                                   (long long)(off / 4 + quad) * 128 + aslot[r] * 4)
                  : base + (long long)(off / 4 + quad) * 128 + aslot[r] * 4;
    astride[r] = 32LL * (s1 ? p.a1_width : p.a0_width);
    ashift[r] = s1 ? p.a1_shift : p.a0_shift;
    acol[r] = 128 + c0 + 4 * quad;
  }

  u32x4 gq[GD];       // GD blocks in flight (ring slot = block index mod GD)
  f32x4 aq[2][AF ? 3 : 6];  // [slot][4 cells of r = 0, 2 cells of r = 1]: two blocks in flight (AF: 8-byte cells, two per register)
  bool use[2][2];
  auto load_g = [&](int b, int gslot) { gq[gslot] = *reinterpret_cast<const u32x4*>(gsrc + (long long)b * gstep); };
  auto load_block = [&](int b, int slot) {
    const int tile = b / L, step = b - tile * L;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int sa = step + ashift[r];
      use[slot][r] = sa >= 0 && sa < L;
      if constexpr (AF) {
        const unsigned short* src = reinterpret_cast<const unsigned short*>(abase[r]) +
                                    (long long)(use[slot][r] ? b + ashift[r] : b) * astride[r];
#pragma unroll
        for (int j = 0; j < (r == 0 ? 2 : 1); ++j) aq[slot][2 * r + j] = *reinterpret_cast<const f32x4*>(src + 8 * j);
      } else {
        const float* src = abase[r] + (long long)(use[slot][r] ? b + ashift[r] : b) * astride[r];
#pragma unroll
        for (int j = 0; j < (r == 0 ? 4 : 2); ++j) aq[slot][4 * r + j] = *reinterpret_cast<const f32x4*>(src + 4 * j);
      }
    }
  };
  float gsum[4] = {0.f, 0.f, 0.f, 0.f}, asum[2][4];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) asum[r][c] = 0.f;
  // pieces of a block: pc 0..3 = column pc of the whole A group, 4..7 = column pc - 4 of the half group, 8..9 = the G
  // columns (0, 1) / (2, 3).  live = 0: tail iteration, the column sums stay untouched
  auto store_piece = [&](int slot, int gslot, __bf16* lds, unsigned live, int pc) {
    const bf16x2 ones = __builtin_bit_cast(bf16x2, live);  // 0x3f803f80 = (1, 1)
    if (pc >= 8) {
      const u32x4 d = gq[gslot];
      const unsigned lo_ = pc == 8 ? d[0] : d[1], hi_ = pc == 8 ? d[2] : d[3];  // slot 2sp / slot 2sp + 1
      const int col = 4 * qg + 2 * (pc - 8);
      if constexpr (F16) {
        const unsigned c_even = __builtin_amdgcn_perm(hi_, lo_, WS_SEL_LO16), c_odd = __builtin_amdgcn_perm(hi_, lo_, WS_SEL_HI16);
        *reinterpret_cast<unsigned*>(lds + col * TB_LD + 2 * sp) = c_even;
        *reinterpret_cast<unsigned*>(lds + (col + 1) * TB_LD + 2 * sp) = c_odd;
        const h16x2 ones_h = __builtin_bit_cast(h16x2, live ? 0x3c003c00u : 0u);  // (1, 1) in fp16
        gsum[2 * (pc - 8)] = __builtin_amdgcn_fdot2(__builtin_bit_cast(h16x2, c_even), ones_h, gsum[2 * (pc - 8)], false);
        gsum[2 * (pc - 8) + 1] = __builtin_amdgcn_fdot2(__builtin_bit_cast(h16x2, c_odd), ones_h, gsum[2 * (pc - 8) + 1], false);
        return;
      }
      if constexpr (GFMT == 2) {
        // four fp16 values: (slot 2sp, col), (slot 2sp, col + 1), (slot 2sp + 1, col), (slot 2sp + 1, col + 1)
        const f16x2 a = __builtin_bit_cast(f16x2, lo_), b = __builtin_bit_cast(f16x2, hi_);
        const float x[4] = {(float)a[0], (float)a[1], (float)b[0], (float)b[1]};
        __bf16 h[4], l[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          h[j] = (__bf16)x[j];
          l[j] = (__bf16)(x[j] - (float)h[j]);
        }
        const bf16x2 he = {h[0], h[2]}, ho = {h[1], h[3]}, le = {l[0], l[2]}, lodd = {l[1], l[3]};
        *reinterpret_cast<bf16x2*>(lds + col * TB_LD + 2 * sp) = he;
        *reinterpret_cast<bf16x2*>(lds + (col + 1) * TB_LD + 2 * sp) = ho;
        *reinterpret_cast<bf16x2*>(lds + PLANE + col * TB_LD + 2 * sp) = le;
        *reinterpret_cast<bf16x2*>(lds + PLANE + (col + 1) * TB_LD + 2 * sp) = lodd;
        if (live) {
          gsum[2 * (pc - 8)] += x[0] + x[2];
          gsum[2 * (pc - 8) + 1] += x[1] + x[3];
        }
        return;
      }
      const unsigned c_even = __builtin_amdgcn_perm(hi_, lo_, WS_SEL_LO16), c_odd = __builtin_amdgcn_perm(hi_, lo_, WS_SEL_HI16);
      *reinterpret_cast<unsigned*>(lds + col * TB_LD + 2 * sp) = c_even;
      *reinterpret_cast<unsigned*>(lds + (col + 1) * TB_LD + 2 * sp) = c_odd;
      gsum[2 * (pc - 8)] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, c_even), ones, gsum[2 * (pc - 8)], false);
      gsum[2 * (pc - 8) + 1] =
          __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, c_odd), ones, gsum[2 * (pc - 8) + 1], false);
      return;
    }
    const int r = pc >> 2, c = pc & 3;
    const unsigned m = use[slot][r] ? 0xffffffffu : 0u;  // uniform: the shifted operand at the sequence ends
    if constexpr (AF) {
      // register j of this group holds the cells of slots 2j, 2j + 1: words (cols 0|1, cols 2|3) of each; column c of the
      // group's slots -> one 16-bit lane of four (two) words
      const unsigned sel = (c & 1) ? WS_SEL_HI16 : WS_SEL_LO16;
      const int wi = c >> 1;
      const u32x4 q0 = __builtin_bit_cast(u32x4, aq[slot][2 * r]);
      const unsigned x = __builtin_amdgcn_perm(q0[2 + wi], q0[wi], sel) & m;          // slots 0, 1
      const int o = (acol[r] + c) * TB_LD + aslot[r];
      if (r == 0) {
        const u32x4 q1 = __builtin_bit_cast(u32x4, aq[slot][1]);
        const unsigned y = __builtin_amdgcn_perm(q1[2 + wi], q1[wi], sel) & m;        // slots 2, 3
        *reinterpret_cast<uint2*>(lds + o) = uint2{x, y};
      } else {
        *reinterpret_cast<unsigned*>(lds + o) = x;
      }
      return;
    }
    unsigned e[4];
#pragma unroll
    for (int j = 0; j < (r == 0 ? 4 : 2); ++j) {
      const float f = aq[slot][4 * r + j][c];  // (bit_cast of a vector-element lvalue reads element 0 with this compiler)
      e[j] = __float_as_uint(f) & m;
    }
    if (ASUM) {
#pragma unroll
      for (int j = 0; j < (r == 0 ? 4 : 2); ++j)
        asum[r][c] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, e[j]), ones, asum[r][c], false);
    }
    const int o = (acol[r] + c) * TB_LD + aslot[r];
    if constexpr (F16) {
      unsigned hw[2], lw[2];
#pragma unroll
      for (int j = 0; j < (r == 0 ? 2 : 1); ++j) {
        const unsigned u0 = e[2 * j] + 0x03000300u, u1 = e[2 * j + 1] + 0x03000300u;  // hi and lo terms x 2^6 (zeros turn
        // into 2^-121, which converts to 0)
        hw[j] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(__uint_as_float(u0 & 0xffff0000u), __uint_as_float(u1 & 0xffff0000u)));
        lw[j] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(__uint_as_float(u0 << 16), __uint_as_float(u1 << 16)));
      }
      if (r == 0) {
        *reinterpret_cast<uint2*>(lds + o) = uint2{hw[0], hw[1]};
        *reinterpret_cast<uint2*>(lds + PLANE + o) = uint2{lw[0], lw[1]};
      } else {
        *reinterpret_cast<unsigned*>(lds + o) = hw[0];
        *reinterpret_cast<unsigned*>(lds + PLANE + o) = lw[0];
      }
      return;
    }
    if (r == 0) {
      uint2 hi, lo;
      hi.x = __builtin_amdgcn_perm(e[1], e[0], WS_SEL_HI16);
      hi.y = __builtin_amdgcn_perm(e[3], e[2], WS_SEL_HI16);
      lo.x = __builtin_amdgcn_perm(e[1], e[0], WS_SEL_LO16);
      lo.y = __builtin_amdgcn_perm(e[3], e[2], WS_SEL_LO16);
      *reinterpret_cast<uint2*>(lds + o) = hi;
      *reinterpret_cast<uint2*>(lds + PLANE + o) = lo;
    } else {
      *reinterpret_cast<unsigned*>(lds + o) = __builtin_amdgcn_perm(e[1], e[0], WS_SEL_HI16);
      *reinterpret_cast<unsigned*>(lds + PLANE + o) = __builtin_amdgcn_perm(e[1], e[0], WS_SEL_LO16);
    }
  };
  constexpr int NPC = 10;
  auto store_block = [&](int slot, int gslot, __bf16* lds, unsigned live) {
#pragma unroll
    for (int pc = 0; pc < NPC; ++pc) store_piece(slot, gslot, lds, live, pc);
  };

  f32x16 acc[2][TN];
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int f = 0; f < TN; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[e][f][r] = 0.f;

  const int nb = b_end - b_begin;
  if (nb > 0) {
    load_block(b_begin, 0);
    load_block(min(b_begin + 1, b_end - 1), 1);
#pragma unroll
    for (int k = 0; k < GD; ++k) load_g(min(b_begin + k, b_end - 1), k);
    store_block(0, 0, ldsA, 0x3f803f80u);
  }
  __syncthreads();
  for (int ib = 0; ib < nb; ib += GD) {
#pragma unroll
    for (int sg4 = 0; sg4 < GD; ++sg4) {
      const int s = sg4 & 1;
      const int i = ib + sg4;
      if (i < nb) {  // uniform; only the tail skips
        // refill: the A registers of slot s held block i (converted one iteration ago) -> block i + 2; the G ring slot of
        // block i -> block i + GD.  A first: its loads come back from L2 and must not queue behind G's HBM round trip
        load_block(b_begin + min(i + 2, nb - 1), s);
        load_g(b_begin + min(i + GD, nb - 1), sg4);
        __builtin_amdgcn_sched_barrier(0);
        const __bf16* lds = s ? ldsB : ldsA;
        __bf16* ldsw = s ? ldsA : ldsB;
        const unsigned live = i + 1 < nb ? 0x3f803f80u : 0u;
        auto lda = [&](int ks, int f, bf16x8& h, bf16x8& l) {
          const int ra = (128 + wn * 32 * TN + f * 32 + l31) * TB_LD + ks + 8 * half;
          h = *reinterpret_cast<const bf16x8*>(lds + ra);
          if constexpr (!AF) l = *reinterpret_cast<const bf16x8*>(lds + PLANE + ra);
        };
        bf16x8 ah, al, ahn, aln;
        lda(0, 0, ah, al);
        int sub = 0;
#pragma unroll
        for (int ks = 0; ks < 32; ks += 16) {
          bf16x8 gh[2], gl[2];
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            gh[e] = *reinterpret_cast<const bf16x8*>(lds + (wm * 64 + e * 32 + l31) * TB_LD + ks + 8 * half);
            if constexpr (GFMT == 2) gl[e] = *reinterpret_cast<const bf16x8*>(lds + PLANE + (wm * 64 + e * 32 + l31) * TB_LD + ks + 8 * half);
          }
#pragma unroll
          for (int f = 0; f < TN; ++f) {
            const bool last = ks == 16 && f == TN - 1;
#pragma unroll
            for (int term = 0; term < NTERM; ++term, ++sub) {
              if (term == 0 && !last) lda(f + 1 < TN ? ks : 16, f + 1 < TN ? f + 1 : 0, ahn, aln);
              // terms: G_hi A_hi, G_hi A_lo and (GFMT 2) G_lo A_hi
#pragma unroll
              for (int e = 0; e < 2; ++e) {
                if constexpr (F16)
                  acc[e][f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, gh[e]),
                                                                     __builtin_bit_cast(f16x8, term == 1 ? al : ah), acc[e][f], 0, 0, 0);
                else
                  acc[e][f] = mfma32(term == 2 ? gl[e] : gh[e], term == 1 ? al : ah, acc[e][f]);
              }
              if constexpr (NTERM == 1) {   // six MFMA groups per block, ten pieces: two per group
                if (2 * sub < NPC) store_piece(s ^ 1, (sg4 + 1) % GD, ldsw, live, 2 * sub);
                if (2 * sub + 1 < NPC) store_piece(s ^ 1, (sg4 + 1) % GD, ldsw, live, 2 * sub + 1);
              } else if (sub < NPC) {
                store_piece(s ^ 1, (sg4 + 1) % GD, ldsw, live, sub);
              }
              __builtin_amdgcn_sched_barrier(0);
            }
            ah = ahn; al = aln;
          }
        }
        __syncthreads();  // image s fully read, image s^1 fully written
      }
    }
  }

  const int ncols = p.a0_cols + p.a1_cols;
  float* out = p.slab + (long long)split * p.slab_stride;
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int f = 0; f < TN; ++f) {
      const int ac_ = wn * 32 * TN + f * 32 + l31;
      if (ac_ < ncols) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int grow = gt * 128 + wm * 64 + e * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          out[(long long)grow * ncols + ac_] = GFMT >= 2 ? acc[e][f][r] * inv_out : acc[e][f][r];
        }
      }
    }
  // column sums of G (bias gradient): over the 16 slot pairs of a quad = lanes sharing tid >> 4
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    float t = gsum[c];
    t += __shfl_xor(t, 1, 64);
    t += __shfl_xor(t, 2, 64);
    t += __shfl_xor(t, 4, 64);
    t += __shfl_xor(t, 8, 64);
    if (p.bslab && sp == 0) p.bslab[(long long)split * p.bslab_stride + gt * 128 + 4 * qg + c] = GFMT >= 2 ? t * inv_s : t;
  }
  if (ASUM && gt == 0 && p.aslab) {  // column sums of Acat: whole groups over the 8 slot groups, half groups over 16 halves
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float t = asum[r][c];
        t += __shfl_xor(t, 1, 64);
        t += __shfl_xor(t, 2, 64);
        t += __shfl_xor(t, 4, 64);
        if (r == 1) t += __shfl_xor(t, 8, 64);
        const bool first = r == 0 ? (tid & 7) == 0 : (tid & 15) == 0;
        if (first) p.aslab[(long long)split * p.aslab_stride + acol[r] - 128 + c] = t;
      }
  }
}

extern "C" int ws_gemm_tnb(const ws_gemm_tnb_args* a, void* stream) {
  WS_REQUIRE(a && a->G && a->A0 && a->slab, "ws_gemm_tnb: null pointer");
  WS_REQUIRE(a->g_cols > 0 && a->g_cols % 128 == 0 && a->g_off % 4 == 0 && a->g_width % 4 == 0,
             "ws_gemm_tnb: G column range");
  WS_REQUIRE(a->a0_cols > 0 && a->a0_cols % 128 == 0 && a->a0_off % 4 == 0 && a->a0_width % 4 == 0,
             "ws_gemm_tnb: A0 column range");
  WS_REQUIRE(a->a1_cols == 0 || (a->A1 && a->a1_cols % 128 == 0 && a->a1_off % 4 == 0 && a->a1_width % 4 == 0),
             "ws_gemm_tnb: A1 column range");
  const int ta = (a->a0_cols + a->a1_cols) / 128;
  WS_REQUIRE(ta == 1 || ta == 3, "ws_gemm_tnb: A columns must total 128 or 384 (got %d)", ta * 128);
  WS_REQUIRE(a->a0_shift == 0, "ws_gemm_tnb: only A1 can be shifted (a0_shift = %d)", a->a0_shift);
  WS_REQUIRE(a->g_fmt == 0 || ((a->g_fmt == 1 || a->g_fmt == 2) && ta == 3),
             "ws_gemm_tnb: g_fmt 1 / 2 (2-byte G) are built for 384 A columns");
  WS_REQUIRE(a->g_fmt != 2 || a->amax, "ws_gemm_tnb: g_fmt = 2 (scaled fp16 G) needs amax");
  WS_REQUIRE(a->a_fmt == 0 || (a->a_fmt == 1 && a->g_fmt == 2 && !a->aslab && ta == 3),
             "ws_gemm_tnb: a_fmt = 1 (fp16 A operands) is built for g_fmt = 2, 384 A columns, no aslab");
  WS_REQUIRE(a->nblk > 0 && a->L > 0 && a->nblk % a->L == 0 && a->nsplit > 0 && a->blocks_per_split > 0 &&
                 (long long)a->nsplit * a->blocks_per_split >= a->nblk,
             "ws_gemm_tnb: bad block split");
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(a->nsplit, a->g_cols / 128), block(512);
  ws_prof_begin(WS_PROF_GEMM_TN, s);
  // WS_TNB_GDEPTH=2|4 (diagnostics): blocks of the 2-byte G operand in flight per thread; both depths give the same bits
  static const int gdepth = [] { const char* e = getenv("WS_TNB_GDEPTH"); return e && atoi(e) == 2 ? 2 : 4; }();
  // WS_TNB_F16=0 (A/B runs): the scaled-fp16 G on the bf16 instruction (3 terms) instead of the fp16 one (2 terms)
  const char* f16env = getenv("WS_TNB_F16");  // read per call: the tests run both forms in one process
  const bool f16mm = !(f16env && atoi(f16env) == 0);
  if (a->a_fmt == 1)
    hipLaunchKernelGGL((gemm_tnb16_kernel<false, 3, 4, 1>), grid, block, 0, s, *a);
  else if (a->g_fmt == 2 && f16mm && a->aslab)
    hipLaunchKernelGGL((gemm_tnb16_kernel<true, 3, 4>), grid, block, 0, s, *a);
  else if (a->g_fmt == 2 && f16mm)
    hipLaunchKernelGGL((gemm_tnb16_kernel<false, 3, 4>), grid, block, 0, s, *a);
  else if (a->g_fmt == 2 && a->aslab)
    hipLaunchKernelGGL((gemm_tnb16_kernel<true, 2, 4>), grid, block, 0, s, *a);
  else if (a->g_fmt == 2 && gdepth == 2)
    hipLaunchKernelGGL((gemm_tnb16_kernel<false, 2, 2>), grid, block, 0, s, *a);
  else if (a->g_fmt == 2)
    hipLaunchKernelGGL((gemm_tnb16_kernel<false, 2, 4>), grid, block, 0, s, *a);
  else if (a->g_fmt && a->aslab)
    hipLaunchKernelGGL((gemm_tnb16_kernel<true, 1, 4>), grid, block, 0, s, *a);
  else if (a->g_fmt && gdepth == 2)
    hipLaunchKernelGGL((gemm_tnb16_kernel<false, 1, 2>), grid, block, 0, s, *a);
  else if (a->g_fmt)
    hipLaunchKernelGGL((gemm_tnb16_kernel<false, 1, 4>), grid, block, 0, s, *a);
  else if (ta == 3 && a->aslab)
    hipLaunchKernelGGL((gemm_tnb_kernel<3, true>), grid, block, 0, s, *a);
  else if (ta == 3)
    hipLaunchKernelGGL((gemm_tnb_kernel<3, false>), grid, block, 0, s, *a);
  else
    hipLaunchKernelGGL((gemm_tnb_kernel<1, true>), grid, block, 0, s, *a);
  ws_prof_end(WS_PROF_GEMM_TN, s);
  return ws_check_launch("ws_gemm_tnb");
}
