// race_repro -- standalone (no torch, no Python) reproducer for the cross-stream disturbance of profiles/r02_kernel_race.md:
// `ws_gemm_b2p` (the library's BL -> plain GEMM: MFMA + LDS weight staging + in-loop HBM loads) runs on stream 0, a
// VICTIM kernel defined in this file runs on stream 1, and every victim launch is compared bit for bit with the same
// launch made while the GPU was otherwise idle.  The victims isolate one ingredient each:
//
//   fft512     radix-2 Stockham FFT, 512 points per workgroup through LDS (the class r02 found: STFT / iSTFT / rocFFT)
//   fft_priv   the same butterflies, one transform per THREAD in registers (no LDS, no barrier)
//   lds_perm   nine LDS permutation passes with barriers, no arithmetic
//   lds_pkadd  the LDS passes with one packed add / subtract per pass;  lds_imad: with integer multiply-adds instead
//   fma_chain  4 096 dependent FMAs per thread, no LDS;  pk_chain: the same as v_pk_fma_f32
//   ring_step  RCCL-shaped: dst[i] = a[i] + b[i] over 64 MB by 32 workgroups (a ring all-reduce step's kernel shape)
//   copy       dst[i] = a[i], all CUs
//
//   a_*        2 048 dependent instructions of ONE kind per thread from inline asm: v_pk_fma_f32, v_pk_mul_f32,
//              v_pk_add_f32, v_fma_f64, v_fma_f32 -- all operands in VGPRs;  s_*: the constant operand in SGPRs
//
//   race_repro [--trials 20] [--mask none|halves|interleave] [--aggr b2p|clone<bits>|own1|own3|own5|own6|own7|copy|none]
//              [--victims a,b,...] [--own-reps 400]
//   own<bits>: aggressors defined here (bit 0 MFMA, bit 1 LDS staging + barriers, bit 2 global loads in the loop,
//              bit 3 v_mov_b64 register moves in the loop)
//
// --mask restricts the two streams to disjoint CU sets (hipExtStreamCreateWithCUMask): "halves" = CUs 0..127 vs 128..255
// of the mask, "interleave" = even vs odd bits.  A disturbance that survives disjoint CUs is not CU-local (LDS, register
// file, instruction cache); one that does not is.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <random>
#include <string>
#include <vector>

#include "../../include/wesep_hip.h"

// gemm_b2p restated with one ingredient removable at a time (b2p_clone.hip, compiled with the library's flags)
extern "C" int b2p_clone_launch(int flags, const ws_gemm_b2p_args* a, hipStream_t s);

#define HIP_OK(x)                                                        \
  do {                                                                   \
    hipError_t e_ = (x);                                                 \
    if (e_ != hipSuccess) {                                              \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));            \
      exit(2);                                                           \
    }                                                                    \
  } while (0)
#define WS_OK_(x)                                                        \
  do {                                                                   \
    int r_ = (x);                                                        \
    if (r_ != WS_OK) {                                                   \
      fprintf(stderr, "%s (rc=%d): %s\n", #x, r_, ws_last_error());      \
      exit(3);                                                           \
    }                                                                    \
  } while (0)

// ---- victims ------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) fft512(const float2* __restrict__ in, const float2* __restrict__ tw,
                                              float2* __restrict__ out, int nrows) {
  __shared__ float2 buf[2][512];
  const int t = threadIdx.x;
  for (int row = blockIdx.x; row < nrows; row += gridDim.x) {
    buf[0][t] = in[(size_t)row * 512 + t];
    buf[0][t + 256] = in[(size_t)row * 512 + t + 256];
    __syncthreads();
    int cur = 0;
    for (int ns = 1; ns < 512; ns <<= 1) {
      const int k = t & (ns - 1);
      const float2 w = tw[k * (256 / ns)];
      const float2 a = buf[cur][t], b0 = buf[cur][t + 256];
      const float2 b = {b0.x * w.x - b0.y * w.y, b0.x * w.y + b0.y * w.x};
      const int j0 = ((t - k) << 1) + k;
      buf[cur ^ 1][j0] = {a.x + b.x, a.y + b.y};
      buf[cur ^ 1][j0 + ns] = {a.x - b.x, a.y - b.y};
      cur ^= 1;
      __syncthreads();
    }
    out[(size_t)row * 512 + t] = buf[cur][t];
    out[(size_t)row * 512 + t + 256] = buf[cur][t + 256];
    __syncthreads();
  }
}

__global__ void __launch_bounds__(256) fft_priv(const float2* __restrict__ in, const float2* __restrict__ tw,
                                                float2* __restrict__ out, int nrows) {
  // 16-point transform per thread, all in registers; rows of 512 = 32 threads x 16 points
  const size_t gt = (size_t)blockIdx.x * 256 + threadIdx.x, n = (size_t)nrows * 32;
  for (size_t i = gt; i < n; i += (size_t)gridDim.x * 256) {
    float2 v[16], u[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = in[i * 16 + k];
#pragma unroll
    for (int ns = 1; ns < 16; ns <<= 1) {
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int k = t & (ns - 1);
        const float2 w = tw[k * (256 / ns)];
        const float2 a = v[t], b0 = v[t + 8];
        const float2 b = {b0.x * w.x - b0.y * w.y, b0.x * w.y + b0.y * w.x};
        const int j0 = ((t - k) << 1) + k;
        u[j0] = {a.x + b.x, a.y + b.y};
        u[j0 + ns] = {a.x - b.x, a.y - b.y};
      }
#pragma unroll
      for (int k = 0; k < 16; ++k) v[k] = u[k];
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) out[i * 16 + k] = v[k];
  }
}

__global__ void __launch_bounds__(256) lds_perm(const float2* __restrict__ in, const float2*, float2* __restrict__ out,
                                                int nrows) {
  __shared__ float2 buf[2][512];
  const int t = threadIdx.x;
  for (int row = blockIdx.x; row < nrows; row += gridDim.x) {
    buf[0][t] = in[(size_t)row * 512 + t];
    buf[0][t + 256] = in[(size_t)row * 512 + t + 256];
    __syncthreads();
    int cur = 0;
    for (int ns = 1; ns < 512; ns <<= 1) {
      const int k = t & (ns - 1), j0 = ((t - k) << 1) + k;
      buf[cur ^ 1][j0] = buf[cur][t];
      buf[cur ^ 1][j0 + ns] = buf[cur][t + 256];
      cur ^= 1;
      __syncthreads();
    }
    out[(size_t)row * 512 + t] = buf[cur][t];
    out[(size_t)row * 512 + t + 256] = buf[cur][t + 256];
    __syncthreads();
  }
}

__global__ void __launch_bounds__(256) fma_chain(const float2* __restrict__ in, const float2*, float2* __restrict__ out,
                                                 int nrows) {
  const size_t gt = (size_t)blockIdx.x * 256 + threadIdx.x, n = (size_t)nrows * 512;
  for (size_t i = gt; i < n; i += (size_t)gridDim.x * 256) {
    float2 v = in[i];
    float a = v.x, b = v.y;
    for (int k = 0; k < 2048; ++k) {
      a = fmaf(a, 0.99993f, b * 1e-4f);
      b = fmaf(b, 0.99991f, -a * 1e-4f);
    }
    out[i] = {a, b};
  }
}

// fma_chain in packed form: float2 lanes so that hipcc emits v_pk_fma_f32 / v_pk_mul_f32 (no LDS)
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(256) pk_chain(const float2* __restrict__ in, const float2*, float2* __restrict__ out,
                                                int nrows) {
  const size_t gt = (size_t)blockIdx.x * 256 + threadIdx.x, n = (size_t)nrows * 512;
  for (size_t i = gt; i < n; i += (size_t)gridDim.x * 256) {
    const float2 v = in[i];
    f32x2_t a = {v.x, v.y}, b = {v.y, v.x};
    const f32x2_t c0 = {0.99993f, 0.99991f}, c1 = {1e-4f, -1e-4f};
    for (int k = 0; k < 1024; ++k) {
      a = __builtin_elementwise_fma(a, c0, b * c1);
      b = __builtin_elementwise_fma(b, c0, a * c1);
    }
    out[i] = {a.x + b.y, a.y + b.x};
  }
}

// one instruction each, from inline asm (nothing the compiler selects): 2 048 dependent ops per thread, no LDS, no loads
#define ASM_CHAIN(NAME, INSTR, T)                                                                                     \
  __global__ void __launch_bounds__(256) NAME(const float2* __restrict__ in, const float2*, float2* __restrict__ out, \
                                              int nrows) {                                                            \
    const size_t gt = (size_t)blockIdx.x * 256 + threadIdx.x, n = (size_t)nrows * 512;                                \
    for (size_t i = gt; i < n; i += (size_t)gridDim.x * 256) {                                                        \
      const float2 v = in[i];                                                                                         \
      T a, b, c;                                                                                                      \
      set3(v, a, b, c);                                                                                               \
      for (int k = 0; k < 256; ++k)                                                                                   \
        asm volatile(INSTR "\n" INSTR "\n" INSTR "\n" INSTR "\n" INSTR "\n" INSTR "\n" INSTR "\n" INSTR                \
                     : "+v"(a)                                                                                        \
                     : "v"(b), "v"(c));                                                                               \
      out[i] = get2(a);                                                                                               \
    }                                                                                                                 \
  }
__device__ __forceinline__ void set3(float2 v, f32x2_t& a, f32x2_t& b, f32x2_t& c) {
  a = {v.x, v.y}, b = {0.99993f, 0.99991f}, c = {v.y * 1e-4f, -v.x * 1e-4f};
}
__device__ __forceinline__ void set3(float2 v, double& a, double& b, double& c) {
  a = v.x + 1e-3 * v.y, b = 0.99993, c = v.y * 1e-4;
}
__device__ __forceinline__ void set3(float2 v, float& a, float& b, float& c) { a = v.x, b = 0.99993f, c = v.y * 1e-4f; }
__device__ __forceinline__ float2 get2(f32x2_t a) { return {a.x, a.y}; }
__device__ __forceinline__ float2 get2(double a) { return {(float)a, (float)(a * 1e3)}; }
__device__ __forceinline__ float2 get2(float a) { return {a, -a}; }
ASM_CHAIN(a_pk_fma, "v_pk_fma_f32 %0, %0, %1, %2", f32x2_t)
ASM_CHAIN(a_pk_mul, "v_pk_mul_f32 %0, %0, %1", f32x2_t)
ASM_CHAIN(a_pk_add, "v_pk_add_f32 %0, %0, %2", f32x2_t)
ASM_CHAIN(a_fma_f64, "v_fma_f64 %0, %0, %1, %2", double)
ASM_CHAIN(a_fma_f32, "v_fma_f32 %0, %0, %1, %2", float)
// packed FP32 with a cross-half operand selection (op_sel): what hipcc emits for complex arithmetic (a.x + b.y, ...)
ASM_CHAIN(o_pk_add, "v_pk_add_f32 %0, %0, %2 op_sel:[0,1] op_sel_hi:[1,0]", f32x2_t)
ASM_CHAIN(o_pk_mul, "v_pk_mul_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]", f32x2_t)
ASM_CHAIN(o_pk_fma, "v_pk_fma_f32 %0, %0, %1, %2 op_sel:[0,1,0] op_sel_hi:[1,0,1]", f32x2_t)
ASM_CHAIN(o_pk_fma2, "v_pk_fma_f32 %0, %0, %1, %2 op_sel:[0,0,1] op_sel_hi:[1,1,0]", f32x2_t)
ASM_CHAIN(o_pk_mov, "v_pk_mov_b32 %0, %0, %0 op_sel:[1,0]", f32x2_t)
// the only operand selection in RCCL 2.26's gfx950 fp32 reduction kernels (FuncPreMulSum<float> = ncclAvg / premul-sum;
// FuncSum<float> has plain v_pk_add_f32 only): the scalar factor's low half broadcast to both lanes through src0
ASM_CHAIN(r_pk_fma, "v_pk_fma_f32 %0, %1, %0, %2 op_sel_hi:[0,1,1]", f32x2_t)
// and the mirror images of the failing form, for the record: selection on src0, and src1 low half broadcast
ASM_CHAIN(o_pk_add0, "v_pk_add_f32 %0, %2, %0 op_sel:[1,0] op_sel_hi:[0,1]", f32x2_t)
ASM_CHAIN(o_pk_mul_b, "v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]", f32x2_t)
// the same with the constant operand in an SGPR (pair): what hipcc emits for a uniform twiddle / scale factor
#define ASM_CHAIN_S(NAME, INSTR, T)                                                                                   \
  __global__ void __launch_bounds__(256) NAME(const float2* __restrict__ in, const float2*, float2* __restrict__ out, \
                                              int nrows) {                                                            \
    const size_t gt = (size_t)blockIdx.x * 256 + threadIdx.x, n = (size_t)nrows * 512;                                \
    for (size_t i = gt; i < n; i += (size_t)gridDim.x * 256) {                                                        \
      const float2 v = in[i];                                                                                         \
      T a, b, c;                                                                                                      \
      set3(v, a, b, c);                                                                                               \
      for (int k = 0; k < 256; ++k)                                                                                   \
        asm volatile(INSTR "\n" INSTR "\n" INSTR "\n" INSTR "\n" INSTR "\n" INSTR "\n" INSTR "\n" INSTR                \
                     : "+v"(a)                                                                                        \
                     : "s"(b), "v"(c));                                                                               \
      out[i] = get2(a);                                                                                               \
    }                                                                                                                 \
  }
// the pair hipcc emits for pk_chain: t = b * c1; (NOP); a = a * c0 + t -- the fma's addend is the multiply's result.
// hipcc separates the two by one wait state (s_nop 0 or an independent SALU instruction).
#define MULFMA(NAME, NOP, CK)                                                                                         \
  __global__ void __launch_bounds__(256) NAME(const float2* __restrict__ in, const float2*, float2* __restrict__ out, \
                                              int nrows) {                                                            \
    const size_t gt = (size_t)blockIdx.x * 256 + threadIdx.x, n = (size_t)nrows * 512;                                \
    const f32x2_t c0 = {0.99993f, 0.99991f}, c1 = {1e-4f, -1e-4f};                                                    \
    for (size_t i = gt; i < n; i += (size_t)gridDim.x * 256) {                                                        \
      const float2 v = in[i];                                                                                         \
      f32x2_t a = {v.x, v.y}, b = {v.y, v.x}, t;                                                                      \
      for (int k = 0; k < 128; ++k)                                                                                   \
        asm volatile("v_pk_mul_f32 %2, %1, %4\n" NOP "v_pk_fma_f32 %0, %0, %3, %2\n" NOP                              \
                     "v_pk_mul_f32 %2, %0, %4\n" NOP "v_pk_fma_f32 %1, %1, %3, %2\n" NOP                              \
                     "v_pk_mul_f32 %2, %1, %4\n" NOP "v_pk_fma_f32 %0, %0, %3, %2\n" NOP                              \
                     "v_pk_mul_f32 %2, %0, %4\n" NOP "v_pk_fma_f32 %1, %1, %3, %2\n" NOP                              \
                     : "+v"(a), "+v"(b), "=&v"(t)                                                                     \
                     : CK(c0), CK(c1));                                                                               \
      out[i] = {a.x + b.y, a.y + b.x};                                                                                \
    }                                                                                                                 \
  }
MULFMA(mf_s_n0, "", "s")
MULFMA(mf_s_n1, "s_nop 0\n", "s")
MULFMA(mf_s_n2, "s_nop 1\n", "s")
MULFMA(mf_s_n4, "s_nop 3\n", "s")
MULFMA(mf_s_n8, "s_nop 7\n", "s")
MULFMA(mf_v_n0, "", "v")
MULFMA(mf_v_n1, "s_nop 0\n", "v")
ASM_CHAIN_S(s_pk_fma, "v_pk_fma_f32 %0, %0, %1, %2", f32x2_t)
ASM_CHAIN_S(s_pk_mul, "v_pk_mul_f32 %0, %0, %1", f32x2_t)
ASM_CHAIN_S(s_pk_add, "v_pk_add_f32 %0, %0, %1", f32x2_t)
ASM_CHAIN_S(s_fma_f64, "v_fma_f64 %0, %0, %1, %2", double)
ASM_CHAIN_S(s_fma_f32, "v_fma_f32 %0, %0, %1, %2", float)

// self-contained aggressors (no library): 512 threads, 64 KB of LDS, two workgroups per CU like gemm_b2p.
// WHAT bit 0: MFMAs, bit 1: LDS staging + reads + barriers, bit 2: global loads inside the loop
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
template <int WHAT>
__global__ void __launch_bounds__(512, 2) own_aggr(const float4* __restrict__ src, float* __restrict__ dst, size_t n,
                                                   int reps) {
  __shared__ float4 stage[4096];
  f32x16_t acc0 = {}, acc1 = {}, acc2 = {}, acc3 = {};
  const int t = threadIdx.x;
  size_t i = ((size_t)blockIdx.x * 512 + t) & (n - 1);  // n is a power of two
  float4 u = {1.f + t, 2.f, 3.f, 4.f}, v = {4.f, 3.f, 2.f, 1.f + t};
  if (WHAT & 2) {
    for (int j = 0; j < 8; ++j) stage[j * 512 + t] = u;
    __syncthreads();
  }
  for (int r = 0; r < reps; ++r) {
    if (WHAT & 4) {
      float4 g[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) g[j] = src[(i + (size_t)j * 65536) & (n - 1)];
      i = (i + 8 * 65536 + 512 * 977) & (n - 1);
      if (WHAT & 2) {
#pragma unroll
        for (int j = 0; j < 8; ++j) stage[j * 512 + t] = g[j];
        __syncthreads();
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) u.x += g[j].x, v.y += g[j].y;
      }
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (WHAT & 2) {
        u = stage[(j * 256 + t * 17) & 4095];
        v = stage[(j * 256 + t * 33 + 7) & 4095];
      }
      if (WHAT & 1) {
        bf16x8_t a, b;
        __builtin_memcpy(&a, &u, 16);
        __builtin_memcpy(&b, &v, 16);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc1, 0, 0, 0);
        if (WHAT & 32) {  // gemm_b2p's LDS : MFMA ratio -- a fresh pair of 128-bit LDS reads in front of EVERY group of three
          // MFMAs (eight reads per twelve MFMAs), four accumulator tiles
          float4 u2 = stage[(j * 256 + t * 17 + 64) & 4095], v2 = stage[(j * 256 + t * 33 + 71) & 4095];
          float4 u3 = stage[(j * 256 + t * 17 + 128) & 4095], v3 = stage[(j * 256 + t * 33 + 135) & 4095];
          float4 u4 = stage[(j * 256 + t * 17 + 192) & 4095], v4 = stage[(j * 256 + t * 33 + 199) & 4095];
          bf16x8_t a2, b2, a3, b3, a4, b4;
          __builtin_memcpy(&a2, &u2, 16);
          __builtin_memcpy(&b2, &v2, 16);
          __builtin_memcpy(&a3, &u3, 16);
          __builtin_memcpy(&b3, &v3, 16);
          __builtin_memcpy(&a4, &u4, 16);
          __builtin_memcpy(&b4, &v4, 16);
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, b, acc0, 0, 0, 0);
          if (WHAT & 64) {  // the FIRST MFMA of a group needs only the first read of its pair: it issues while the second
            // 128-bit LDS read is still in flight (s_waitcnt lgkmcnt(1)), as in gemm_b2p's loop
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, a2, acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b2, a2, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, a3, acc2, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b3, acc2, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b3, a3, acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a4, a4, acc3, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a4, b4, acc3, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b4, a4, acc3, 0, 0, 0);
          } else {
          acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, acc1, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b2, a2, acc1, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, a2, acc1, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b3, acc2, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b3, a3, acc2, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, a3, acc2, 0, 0, 0);
          acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a4, b4, acc3, 0, 0, 0);
          acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b4, a4, acc3, 0, 0, 0);
          acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a4, a4, acc3, 0, 0, 0);
          }
        } else if (WHAT & 16) {  // gemm_b2p's density: four accumulator tiles x three dependent MFMAs per pair of LDS reads
          acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc2, 0, 0, 0);
          acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc3, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, b, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, a, acc1, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, b, acc2, 0, 0, 0);
          acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, a, acc3, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc1, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, a, acc2, 0, 0, 0);
          acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, b, acc3, 0, 0, 0);
        }
      } else {
        acc0[j] += u.x * v.y;
      }
    }
    if (WHAT & 8) {  // 64-bit register moves, as hipcc emits them for gemm_b2p's prefetch rotation
      double d0, d1;
      __builtin_memcpy(&d0, &u, 8);
      __builtin_memcpy(&d1, &v, 8);
      asm volatile("v_mov_b64 %0, %2\n v_mov_b64 %1, %3\n v_mov_b64 %0, %1\n v_mov_b64 %1, %2\n"
                   "v_mov_b64 %0, %3\n v_mov_b64 %1, %0\n v_mov_b64 %0, %2\n v_mov_b64 %1, %3"
                   : "=&v"(d0), "=&v"(d1)
                   : "v"(d0), "v"(d1));
      __builtin_memcpy(&u, &d0, 8);
      __builtin_memcpy(&v, &d1, 8);
    }
    if (WHAT & 2) __syncthreads();
  }
  float s = u.x + v.y;
  for (int k = 0; k < 16; ++k) s += acc0[k] + acc1[k] + acc2[k] + acc3[k];
  dst[(size_t)blockIdx.x * 512 + t] = s;
}

// SCLK estimate: shader-clock ticks (s_memtime) per 100 MHz reference tick (s_memrealtime) over a busy loop
__global__ void clock_probe(unsigned long long* out) {
  const unsigned long long c0 = clock64(), w0 = wall_clock64();
  float a = threadIdx.x;
  for (int k = 0; k < 200000; ++k) a = fmaf(a, 0.9999f, 1e-3f);
  const unsigned long long c1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0) out[0] = c1 - c0, out[1] = w1 - w0, out[2] = (unsigned long long)a;
}

// LDS round trips with ONE packed add per pass (no barrier-free stretch, no multiplies)
__global__ void __launch_bounds__(256) lds_pkadd(const float2* __restrict__ in, const float2*, float2* __restrict__ out,
                                                 int nrows) {
  __shared__ float2 buf[2][512];
  const int t = threadIdx.x;
  for (int row = blockIdx.x; row < nrows; row += gridDim.x) {
    buf[0][t] = in[(size_t)row * 512 + t];
    buf[0][t + 256] = in[(size_t)row * 512 + t + 256];
    __syncthreads();
    int cur = 0;
    for (int ns = 1; ns < 512; ns <<= 1) {
      const int k = t & (ns - 1), j0 = ((t - k) << 1) + k;
      const float2 a = buf[cur][t], b = buf[cur][t + 256];
      buf[cur ^ 1][j0] = {a.x + b.x, a.y + b.y};
      buf[cur ^ 1][j0 + ns] = {a.x - b.x, a.y - b.y};
      cur ^= 1;
      __syncthreads();
    }
    out[(size_t)row * 512 + t] = buf[cur][t];
    out[(size_t)row * 512 + t + 256] = buf[cur][t + 256];
    __syncthreads();
  }
}

// the fft512 butterflies with the LDS exchange but every value kept in integer registers between passes: the
// arithmetic is integer multiply-add (v_mad_u32 / v_mul_lo), same LDS traffic and barriers
__global__ void __launch_bounds__(256) lds_imad(const float2* __restrict__ in, const float2*, float2* __restrict__ out,
                                                int nrows) {
  __shared__ uint2 buf[2][512];
  const int t = threadIdx.x;
  const uint2* inu = reinterpret_cast<const uint2*>(in);
  uint2* outu = reinterpret_cast<uint2*>(out);
  for (int row = blockIdx.x; row < nrows; row += gridDim.x) {
    buf[0][t] = inu[(size_t)row * 512 + t];
    buf[0][t + 256] = inu[(size_t)row * 512 + t + 256];
    __syncthreads();
    int cur = 0;
    for (int ns = 1; ns < 512; ns <<= 1) {
      const int k = t & (ns - 1), j0 = ((t - k) << 1) + k;
      const uint2 a = buf[cur][t], b = buf[cur][t + 256];
      buf[cur ^ 1][j0] = {a.x * 2654435761u + b.x, a.y * 40503u + b.y};
      buf[cur ^ 1][j0 + ns] = {a.x - b.x * 2246822519u, a.y - b.y * 3266489917u};
      cur ^= 1;
      __syncthreads();
    }
    outu[(size_t)row * 512 + t] = buf[cur][t];
    outu[(size_t)row * 512 + t + 256] = buf[cur][t + 256];
    __syncthreads();
  }
}

__global__ void __launch_bounds__(512) ring_step(const float4* __restrict__ a, const float4* __restrict__ b,
                                                 float4* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 512 + threadIdx.x; i < n; i += (size_t)gridDim.x * 512) {
    const float4 x = a[i], y = b[i];
    dst[i] = {x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w};
  }
}

__global__ void copy_kernel(const float4* __restrict__ a, float4* __restrict__ dst, size_t n, int reps) {
  for (int r = 0; r < reps; ++r)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = a[i];
}

// ---- host ---------------------------------------------------------------------------------------------------------
static float* drandom(size_t n, float scale, unsigned seed) {
  const size_t blk = n < (1u << 20) ? n : (1u << 20);
  std::vector<float> h(blk);
  std::mt19937 rng(seed);
  std::normal_distribution<float> nd(0.f, scale);
  for (auto& v : h) v = nd(rng);
  float* p = nullptr;
  HIP_OK(hipMalloc(reinterpret_cast<void**>(&p), n * 4));
  for (size_t o = 0; o < n; o += blk)
    HIP_OK(hipMemcpy(p + o, h.data(), (n - o < blk ? n - o : blk) * 4, hipMemcpyHostToDevice));
  return p;
}

int main(int argc, char** argv) {
  std::map<std::string, std::string> kv;
  for (int i = 1; i + 1 < argc; i += 2) kv[argv[i]] = argv[i + 1];
  auto get = [&](const char* k, const char* d) { return kv.count(k) ? kv[k] : std::string(d); };
  const int trials = atoi(get("--trials", "20").c_str());
  const std::string mask = get("--mask", "none"), aggr = get("--aggr", "b2p");
  const std::string victims = get("--victims", "fft512,fft_priv,lds_perm,lds_pkadd,lds_imad,fma_chain,pk_chain,a_pk_fma,a_pk_mul,a_pk_add,a_fma_f64,"
                                   "a_fma_f32,s_pk_fma,s_pk_mul,s_pk_add,s_fma_f64,s_fma_f32,o_pk_add,o_pk_mul,o_pk_fma,o_pk_fma2,o_pk_mov,r_pk_fma,o_pk_add0,o_pk_mul_b,mf_s_n0,mf_s_n1,mf_s_n2,mf_s_n4,mf_s_n8,mf_v_n0,mf_v_n1,ring_step,copy");
  hipDeviceProp_t prop;
  HIP_OK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  hipStream_t s0, s1;
  if (mask == "none") {
    HIP_OK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    HIP_OK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  } else {
    const int words = (ncu + 31) / 32;
    std::vector<uint32_t> m0(words, 0), m1(words, 0);
    for (int c = 0; c < ncu; ++c) {
      const bool first = mask == "halves" ? c < ncu / 2 : (c & 1) == 0;
      (first ? m0 : m1)[c / 32] |= 1u << (c % 32);
    }
    HIP_OK(hipExtStreamCreateWithCUMask(&s0, words, m0.data()));
    HIP_OK(hipExtStreamCreateWithCUMask(&s1, words, m1.data()));
  }
  printf("# %s, %d CUs; aggressor %s on stream 0, victims on stream 1, CU mask %s, %d trials\n", prop.name, ncu,
         aggr.c_str(), mask.c_str(), trials);

  // aggressor: ws_gemm_b2p at the band-view shape of the training step (K = 256 -> N = 128, 41 steps)
  const int nseq = 8192, L = 41, K = 256, N = 128;
  const size_t pos = (size_t)nseq * L;
  float* A = drandom(pos * K, 0.5f, 1);
  float* W = drandom((size_t)N * K, 0.05f, 2);
  float* Wp = nullptr;
  HIP_OK(hipMalloc(reinterpret_cast<void**>(&Wp), (size_t)N * K * 4));
  float* bias = drandom(N, 0.1f, 3);
  float* Rres = drandom(pos * N, 1.0f, 4);
  float *Cagg = nullptr, *Cref = nullptr;
  HIP_OK(hipMalloc(reinterpret_cast<void**>(&Cagg), pos * N * 4));
  WS_OK_(ws_pack_w(W, N, K, K, 0, 1, Wp, s0));
  // --data zero|ones: the aggressor's operands (A and the packed weights) all zero bits / all bf16 1.0 instead of random values
  // (same instruction stream, no toggling in the MFMA / LDS data paths): is the disturbance data- (= power-) dependent?
  const std::string adata = get("--data", "random");
  if (adata != "random") {
    HIP_OK(hipStreamSynchronize(s0));
    const int byte = adata == "zero" ? 0 : 0x3f;     // 0x3f3f3f3f: bf16 pairs 0x3f3f = 0.746 (a constant, non-zero pattern)
    HIP_OK(hipMemset(A, byte, pos * K * 4));
    HIP_OK(hipMemset(Wp, byte, (size_t)N * K * 4));
    HIP_OK(hipMemset(Rres, 0, pos * N * 4));
    HIP_OK(hipDeviceSynchronize());
  }
  ws_gemm_b2p_args g = {};
  g.A = A, g.Wpack = Wp, g.bias = bias, g.R = Rres, g.C = Cagg;
  g.sm.nseq = nseq, g.sm.L = L, g.sm.sq_div = 1 << 30, g.sm.sq_s1 = 0, g.sm.sq_s2 = L, g.sm.step_rows = 1;
  g.ldc = N, g.N = N, g.K = K;
  const size_t cpn = (size_t)1 << 25;  // 512 MB of float4 for the copy aggressor
  float4 *cpa = nullptr, *cpd = nullptr;
  if (aggr == "copy") {
    HIP_OK(hipMalloc(reinterpret_cast<void**>(&cpa), cpn * 16));
    HIP_OK(hipMalloc(reinterpret_cast<void**>(&cpd), cpn * 16));
    HIP_OK(hipMemset(cpa, 1, cpn * 16));
  }
  const int own_reps = atoi(get("--own-reps", "400").c_str());
  {
    unsigned long long* cp = nullptr;
    HIP_OK(hipMalloc(reinterpret_cast<void**>(&cp), 32));
    hipLaunchKernelGGL(clock_probe, dim3(1), dim3(64), 0, s1, cp);
    unsigned long long h[3];
    HIP_OK(hipStreamSynchronize(s1));
    HIP_OK(hipMemcpy(h, cp, 24, hipMemcpyDeviceToHost));
    printf("# shader clock during a busy loop: %.0f MHz (s_memtime / s_memrealtime x 100 MHz)\n", 100.0 * h[0] / h[1]);
  }
  auto aggressor = [&]() {
    if (aggr == "b2p") WS_OK_(ws_gemm_b2p(&g, s0));
    if (aggr.rfind("clone", 0) == 0 && b2p_clone_launch(atoi(aggr.c_str() + 5), &g, s0)) {   // tools/cbench/b2p_clone.hip
      fprintf(stderr, "aggressor %s: variant not built\n", aggr.c_str());
      exit(4);
    }
    if (aggr == "copy") hipLaunchKernelGGL(copy_kernel, dim3(1024), dim3(256), 0, s0, cpa, cpd, cpn, 1);
    const float4* osrc = reinterpret_cast<const float4*>(A);
    size_t on = 1;
    while (on * 2 <= pos * K / 4) on *= 2;
    // --own-wpc N: own_aggr workgroups per CU in the grid (default 2 = the kernel's launch bounds; with 1, half of every CU's
    // registers and wave slots stay free for the victim whatever the aggressor's register count)
    const dim3 og(atoi(get("--own-wpc", "2").c_str()) * ncu), ob(512);
    if (aggr == "own1") hipLaunchKernelGGL(own_aggr<1>, og, ob, 0, s0, osrc, Cagg, on, own_reps);
    if (aggr == "own3") hipLaunchKernelGGL(own_aggr<3>, og, ob, 0, s0, osrc, Cagg, on, own_reps);
    if (aggr == "own5") hipLaunchKernelGGL(own_aggr<5>, og, ob, 0, s0, osrc, Cagg, on, own_reps);
    if (aggr == "own6") hipLaunchKernelGGL(own_aggr<6>, og, ob, 0, s0, osrc, Cagg, on, own_reps);
    if (aggr == "own9") hipLaunchKernelGGL(own_aggr<9>, og, ob, 0, s0, osrc, Cagg, on, own_reps);
    if (aggr == "own15") hipLaunchKernelGGL(own_aggr<15>, og, ob, 0, s0, osrc, Cagg, on, own_reps);
    if (aggr == "own7") hipLaunchKernelGGL(own_aggr<7>, og, ob, 0, s0, osrc, Cagg, on, own_reps);
    if (aggr == "own19") hipLaunchKernelGGL(own_aggr<19>, og, ob, 0, s0, osrc, Cagg, on, own_reps);
    if (aggr == "own17") hipLaunchKernelGGL(own_aggr<17>, og, ob, 0, s0, osrc, Cagg, on, own_reps);
    if (aggr == "own23") hipLaunchKernelGGL(own_aggr<23>, og, ob, 0, s0, osrc, Cagg, on, own_reps);
    if (aggr == "own35") hipLaunchKernelGGL(own_aggr<35>, og, ob, 0, s0, osrc, Cagg, on, own_reps);
    if (aggr == "own39") hipLaunchKernelGGL(own_aggr<39>, og, ob, 0, s0, osrc, Cagg, on, own_reps);
    if (aggr == "own99") hipLaunchKernelGGL(own_aggr<99>, og, ob, 0, s0, osrc, Cagg, on, own_reps);
  };
  std::vector<float> cref;
  if (aggr != "none") {
    aggressor();
    HIP_OK(hipStreamSynchronize(s0));
    if (aggr == "b2p") {
      cref.resize(pos * N);
      HIP_OK(hipMemcpy(cref.data(), Cagg, pos * N * 4, hipMemcpyDeviceToHost));
    }
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0));
    HIP_OK(hipEventCreate(&e1));
    HIP_OK(hipEventRecord(e0, s0));
    aggressor();
    HIP_OK(hipEventRecord(e1, s0));
    HIP_OK(hipEventSynchronize(e1));
    float ms = 0;
    HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    printf("# aggressor alone: %.3f ms per launch\n", ms);
  }
  (void)Cref;

  // victims' data
  const int nrows = 8192;  // 8192 x 512 complex = 32 MB in, 32 MB out per launch
  const size_t vn = (size_t)nrows * 512;
  float2* vin = reinterpret_cast<float2*>(drandom(vn * 2, 1.0f, 7));
  float2* vin2 = reinterpret_cast<float2*>(drandom(vn * 2, 1.0f, 8));
  std::vector<float2> twh(256);
  for (int k = 0; k < 256; ++k) twh[k] = {(float)cos(-2.0 * M_PI * k / 512), (float)sin(-2.0 * M_PI * k / 512)};
  float2* tw = nullptr;
  HIP_OK(hipMalloc(reinterpret_cast<void**>(&tw), 256 * 8));
  HIP_OK(hipMemcpy(tw, twh.data(), 256 * 8, hipMemcpyHostToDevice));
  const int reps = 6;  // victim launches per trial, each into its own output slab
  float2* vout = nullptr;
  HIP_OK(hipMalloc(reinterpret_cast<void**>(&vout), vn * 8 * reps));
  std::vector<float2> ref(vn), got(vn);

  struct Vic {
    const char* name;
    int grid;
  };
  const Vic all[] = {{"fft512", 2048}, {"fft_priv", 1024}, {"lds_perm", 2048}, {"fma_chain", 2048}, {"lds_pkadd", 2048}, {"lds_imad", 2048},
                     {"pk_chain", 2048},  {"a_pk_fma", 2048},  {"a_pk_mul", 2048}, {"a_pk_add", 2048},
                     {"a_fma_f64", 2048}, {"a_fma_f32", 2048}, {"s_pk_fma", 2048},  {"s_pk_mul", 2048},
                     {"s_pk_add", 2048},  {"s_fma_f64", 2048}, {"s_fma_f32", 2048}, {"o_pk_add", 2048}, {"o_pk_mul", 2048}, {"o_pk_fma", 2048}, {"o_pk_fma2", 2048}, {"o_pk_mov", 2048}, {"r_pk_fma", 2048}, {"o_pk_add0", 2048}, {"o_pk_mul_b", 2048}, {"mf_s_n0", 2048}, {"mf_s_n1", 2048}, {"mf_s_n2", 2048}, {"mf_s_n4", 2048}, {"mf_s_n8", 2048}, {"mf_v_n0", 2048}, {"mf_v_n1", 2048}, {"ring_step", 32},
                     {"copy", 1024}};
  for (const Vic& v : all) {
    if (("," + victims + ",").find(std::string(",") + v.name + ",") == std::string::npos) continue;
    auto launch = [&](float2* out) {
      const std::string n = v.name;
      if (n == "fft512") hipLaunchKernelGGL(fft512, dim3(v.grid), dim3(256), 0, s1, vin, tw, out, nrows);
      if (n == "fft_priv") hipLaunchKernelGGL(fft_priv, dim3(v.grid), dim3(256), 0, s1, vin, tw, out, nrows);
      if (n == "lds_perm") hipLaunchKernelGGL(lds_perm, dim3(v.grid), dim3(256), 0, s1, vin, tw, out, nrows);
      if (n == "lds_pkadd") hipLaunchKernelGGL(lds_pkadd, dim3(v.grid), dim3(256), 0, s1, vin, tw, out, nrows);
      if (n == "lds_imad") hipLaunchKernelGGL(lds_imad, dim3(v.grid), dim3(256), 0, s1, vin, tw, out, nrows);
      if (n == "a_pk_fma") hipLaunchKernelGGL(a_pk_fma, dim3(v.grid), dim3(256), 0, s1, vin, tw, out, nrows);
      if (n == "a_pk_mul") hipLaunchKernelGGL(a_pk_mul, dim3(v.grid), dim3(256), 0, s1, vin, tw, out, nrows);
      if (n == "a_pk_add") hipLaunchKernelGGL(a_pk_add, dim3(v.grid), dim3(256), 0, s1, vin, tw, out, nrows);
      if (n == "a_fma_f64") hipLaunchKernelGGL(a_fma_f64, dim3(v.grid), dim3(256), 0, s1, vin, tw, out, nrows);
      if (n == "a_fma_f32") hipLaunchKernelGGL(a_fma_f32, dim3(v.grid), dim3(256), 0, s1, vin, tw, out, nrows);
      if (n == "s_pk_fma") hipLaunchKernelGGL(s_pk_fma, dim3(v.grid), dim3(256), 0, s1, vin, tw, out, nrows);
      if (n == "s_pk_mul") hipLaunchKernelGGL(s_pk_mul, dim3(v.grid), dim3(256), 0, s1, vin, tw, out, nrows);
      if (n == "s_pk_add") hipLaunchKernelGGL(s_pk_add, dim3(v.grid), dim3(256), 0, s1, vin, tw, out, nrows);
      if (n == "s_fma_f64") hipLaunchKernelGGL(s_fma_f64, dim3(v.grid), dim3(256), 0, s1, vin, tw, out, nrows);
      if (n == "s_fma_f32") hipLaunchKernelGGL(s_fma_f32, dim3(v.grid), dim3(256), 0, s1, vin, tw, out, nrows);
      if (n == "mf_s_n0") hipLaunchKernelGGL(mf_s_n0, dim3(v.grid), dim3(256), 0, s1, vin, tw, out, nrows);
      if (n == "mf_s_n1") hipLaunchKernelGGL(mf_s_n1, dim3(v.grid), dim3(256), 0, s1, vin, tw, out, nrows);
      if (n == "mf_s_n2") hipLaunchKernelGGL(mf_s_n2, dim3(v.grid), dim3(256), 0, s1, vin, tw, out, nrows);
      if (n == "mf_s_n4") hipLaunchKernelGGL(mf_s_n4, dim3(v.grid), dim3(256), 0, s1, vin, tw, out, nrows);
      if (n == "mf_s_n8") hipLaunchKernelGGL(mf_s_n8, dim3(v.grid), dim3(256), 0, s1, vin, tw, out, nrows);
      if (n == "mf_v_n0") hipLaunchKernelGGL(mf_v_n0, dim3(v.grid), dim3(256), 0, s1, vin, tw, out, nrows);
      if (n == "mf_v_n1") hipLaunchKernelGGL(mf_v_n1, dim3(v.grid), dim3(256), 0, s1, vin, tw, out, nrows);
      if (n == "o_pk_add") hipLaunchKernelGGL(o_pk_add, dim3(v.grid), dim3(256), 0, s1, vin, tw, out, nrows);
      if (n == "o_pk_mul") hipLaunchKernelGGL(o_pk_mul, dim3(v.grid), dim3(256), 0, s1, vin, tw, out, nrows);
      if (n == "o_pk_fma") hipLaunchKernelGGL(o_pk_fma, dim3(v.grid), dim3(256), 0, s1, vin, tw, out, nrows);
      if (n == "o_pk_fma2") hipLaunchKernelGGL(o_pk_fma2, dim3(v.grid), dim3(256), 0, s1, vin, tw, out, nrows);
      if (n == "o_pk_mov") hipLaunchKernelGGL(o_pk_mov, dim3(v.grid), dim3(256), 0, s1, vin, tw, out, nrows);
      if (n == "r_pk_fma") hipLaunchKernelGGL(r_pk_fma, dim3(v.grid), dim3(256), 0, s1, vin, tw, out, nrows);
      if (n == "o_pk_add0") hipLaunchKernelGGL(o_pk_add0, dim3(v.grid), dim3(256), 0, s1, vin, tw, out, nrows);
      if (n == "o_pk_mul_b") hipLaunchKernelGGL(o_pk_mul_b, dim3(v.grid), dim3(256), 0, s1, vin, tw, out, nrows);
      if (n == "pk_chain") hipLaunchKernelGGL(pk_chain, dim3(v.grid), dim3(256), 0, s1, vin, tw, out, nrows);
      if (n == "fma_chain") hipLaunchKernelGGL(fma_chain, dim3(v.grid), dim3(256), 0, s1, vin, tw, out, nrows);
      if (n == "ring_step")
        hipLaunchKernelGGL(ring_step, dim3(v.grid), dim3(512), 0, s1, reinterpret_cast<const float4*>(vin),
                           reinterpret_cast<const float4*>(vin2), reinterpret_cast<float4*>(out), vn / 2);
      if (n == "copy")
        hipLaunchKernelGGL(copy_kernel, dim3(v.grid), dim3(256), 0, s1, reinterpret_cast<const float4*>(vin),
                           reinterpret_cast<float4*>(out), vn / 2, 1);
    };
    // reference: alone, twice (self-consistency)
    launch(vout);
    launch(vout + vn);
    HIP_OK(hipStreamSynchronize(s1));
    HIP_OK(hipMemcpy(ref.data(), vout, vn * 8, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(got.data(), vout + vn, vn * 8, hipMemcpyDeviceToHost));
    const bool self_ok = memcmp(ref.data(), got.data(), vn * 8) == 0;
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0));
    HIP_OK(hipEventCreate(&e1));
    HIP_OK(hipEventRecord(e0, s1));
    launch(vout);
    HIP_OK(hipEventRecord(e1, s1));
    HIP_OK(hipEventSynchronize(e1));
    float vms = 0;
    HIP_OK(hipEventElapsedTime(&vms, e0, e1));
    int bad_launches = 0, bad_trials = 0, aggr_bad = 0;
    size_t bad_elems = 0, first_bad = 0;
    float2 first_got = {0, 0}, first_want = {0, 0};
    for (int tr = 0; tr < trials; ++tr) {
      HIP_OK(hipMemsetAsync(vout, 0xff, vn * 8 * reps, s1));
      HIP_OK(hipStreamSynchronize(s1));
      for (int r = 0; r < reps; ++r) {
        aggressor();
        launch(vout + (size_t)r * vn);
      }
      HIP_OK(hipStreamSynchronize(s0));
      HIP_OK(hipStreamSynchronize(s1));
      bool trial_bad = false;
      for (int r = 0; r < reps; ++r) {
        HIP_OK(hipMemcpy(got.data(), vout + (size_t)r * vn, vn * 8, hipMemcpyDeviceToHost));
        if (memcmp(ref.data(), got.data(), vn * 8) == 0) continue;
        ++bad_launches, trial_bad = true;
        for (size_t i = 0; i < vn; ++i)
          if (memcmp(&ref[i], &got[i], 8) != 0) {
            if (!bad_elems) first_bad = i, first_got = got[i], first_want = ref[i];
            ++bad_elems;
          }
      }
      bad_trials += trial_bad;
      if (aggr == "b2p" && tr == trials - 1) {
        std::vector<float> c(pos * N);
        HIP_OK(hipMemcpy(c.data(), Cagg, pos * N * 4, hipMemcpyDeviceToHost));
        aggr_bad = memcmp(c.data(), cref.data(), pos * N * 4) != 0;
      }
    }
    printf("%-10s alone %.3f ms, self-consistent %s | beside the aggressor: %d of %d trials, %d of %d launches differ, "
           "%zu elements",
           v.name, vms, self_ok ? "yes" : "NO", bad_trials, trials, bad_launches, trials * reps, bad_elems);
    if (bad_elems)
      printf("; first at row %zu col %zu: got (%g, %g) want (%g, %g)", first_bad / 512, first_bad % 512, first_got.x,
             first_got.y, first_want.x, first_want.y);
    printf("; aggressor output %s\n", aggr == "b2p" ? (aggr_bad ? "CHANGED" : "intact") : "n/a");
    fflush(stdout);
  }
  return 0;
}
