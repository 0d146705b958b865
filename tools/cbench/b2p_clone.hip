// b2p_clone -- the library's gemm_b2p_kernel<0> (wesep_amd/csrc/gemm_blk.hip: BL -> plain GEMM, split-bf16 operands, MFMA +
// LDS weight staging + in-loop operand loads + staged epilogue) restated HERE with compile-time switches that remove one
// ingredient each, for tools/cbench/race_repro: which part of the one known aggressor of the packed-FP32 disturbance
// (profiles/r03_kernel_race.md) is needed to disturb `v_pk_add_f32 ... op_sel:[0,1] op_sel_hi:[1,0]` on the same CU?
// Compiled with the LIBRARY's flags (-O3, packed FP32 off), linked into race_repro; `--aggr clone<bits>` selects a variant.
//   bit 0 (1)    no MFMAs (one scalar FMA per product keeps the operands live)
//   bit 1 (2)    weights not staged through LDS (fragments straight from the prefetch registers, no in-loop barrier)
//   bit 2 (4)    no in-loop loads of the A operand (the first stage's cells are reused)
//   bit 3 (8)    no in-loop loads of the weights
//   bit 4 (16)   no epilogue (no LDS staging of the accumulators, no residual loads, one 4-byte store per thread)
//   bit 5 (32)   no v_perm unpack of the BLS cells (raw bits as fragments)
//   bit 6 (64)   one MFMA per product instead of three
//   bit 7 (128)  __launch_bounds__(512, 1) instead of (512, 2)
//   bit 8 (256)  accumulators re-zeroed every stage and summed into one scalar (no long MFMA accumulation chains)
//   bit 9 (512)  the LDS reads of the weight fragments in a bank-conflicting pattern (lane stride 17 cells) instead of the
//                conflict-free contiguous one
//   bit 10 (1024) the LDS reads stay but do not feed the MFMAs (their values go into a scalar; MFMA operands from registers)
//   bit 11 (2048) no LDS WRITES inside the loop (the first stage's image is read over and over; the barrier stays)
//   bit 13 (8192) ONE accumulator tile instead of four (a quarter of the MFMAs per LDS read and of the accumulator registers)
//   bit 14 (16384) with bit 13: the single tile's work four times over (the MFMA and LDS-read COUNT of four tiles, the register
//                footprint of one);   bit 15 (32768) TWO accumulator tiles
//   bit 16 (65536) with bit 11: no barrier inside the loop either;   bit 17 (131072) the A operand's fragments from LDS as well
//                (both MFMA operands are fresh LDS reads, as in race_repro's own_aggr)
//   bit 12 (4096) long-lived workgroups: a grid of 2 x CUs workgroups that each repeat the main loop 48 times, instead of
//                1 312 workgroups of ~50 us (no wave launch / retirement churn beside the victim)
#include <hip/hip_runtime.h>

#include "../../include/wesep_hip.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define SEL_LO16 0x05040100u
#define SEL_HI16 0x07060302u

__device__ __forceinline__ void unpack8(const u32x4& c0, const u32x4& c1, bf16x8& hi, bf16x8& lo) {
  u32x4 h, l;
  h[0] = __builtin_amdgcn_perm(c0[1], c0[0], SEL_HI16);
  h[1] = __builtin_amdgcn_perm(c0[3], c0[2], SEL_HI16);
  h[2] = __builtin_amdgcn_perm(c1[1], c1[0], SEL_HI16);
  h[3] = __builtin_amdgcn_perm(c1[3], c1[2], SEL_HI16);
  l[0] = __builtin_amdgcn_perm(c0[1], c0[0], SEL_LO16);
  l[1] = __builtin_amdgcn_perm(c0[3], c0[2], SEL_LO16);
  l[2] = __builtin_amdgcn_perm(c1[1], c1[0], SEL_LO16);
  l[3] = __builtin_amdgcn_perm(c1[3], c1[2], SEL_LO16);
  hi = __builtin_bit_cast(bf16x8, h);
  lo = __builtin_bit_cast(bf16x8, l);
}

__device__ __forceinline__ long long seq_pos(const ws_seqmap& sm, int b, int i, bool& valid) {
  const int tile = b / sm.L, step = b - tile * sm.L;
  const int seq = tile * 32 + i;
  valid = seq < sm.nseq;
  const int s = valid ? seq : sm.nseq - 1;
  return (long long)(s / sm.sq_div) * sm.sq_s1 + (long long)(s % sm.sq_div) * sm.sq_s2 + (long long)step * sm.step_rows;
}

template <int F>
__device__ __forceinline__ f32x16 prod(bf16x8 a, bf16x8 b, f32x16 c) {
  if constexpr (F & 1) {
    c[0] = __builtin_fmaf((float)a[0], (float)b[0], c[0]);
    return c;
  } else {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
}

template <int F>
__device__ __forceinline__ void clone_body(const ws_gemm_b2p_args& p) {
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
  const float* ab = p.A + (long long)bb * 32 * K + i * 4 + 2 * half * 128;
  const u32x4* wsrc = reinterpret_cast<const u32x4*>(p.Wpack);
  u32x4 wreg[4];
  f32x4 an[8];
#pragma unroll
  for (int q = 0; q < 4; ++q) wreg[q] = wsrc[tid + 512 * q];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    an[2 * ks] = *reinterpret_cast<const f32x4*>(ab + (4 * ks) * 128);
    an[2 * ks + 1] = *reinterpret_cast<const f32x4*>(ab + (4 * ks + 1) * 128);
  }
  if constexpr (!(F & 2)) {
#pragma unroll
    for (int q = 0; q < 4; ++q) wl[0][tid + 512 * q] = wreg[q];
  }
  __syncthreads();

  f32x16 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float side = 0.f;

  const int nrep = (F & 4096) ? 48 : 1;
  for (int rep = 0; rep < nrep; ++rep)
  for (int st = 0; st < nstage; ++st) {
    const int cur = (F & 2048) ? 0 : (st & 1);
    f32x4 ac[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) ac[q] = an[q];
    u32x4 wcur[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) wcur[q] = wreg[q];
    if (st + 1 < nstage) {
      if constexpr (!(F & 8)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) wreg[q] = wsrc[(long long)(st + 1) * 2048 + tid + 512 * q];
      }
      if constexpr (!(F & 4)) {
        const float* a2 = ab + (long long)(st + 1) * 16 * 128;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          an[2 * ks] = *reinterpret_cast<const f32x4*>(a2 + (4 * ks) * 128);
          an[2 * ks + 1] = *reinterpret_cast<const f32x4*>(a2 + (4 * ks + 1) * 128);
        }
      }
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8 ah, al;
      if constexpr (F & 32) {
        ah = __builtin_bit_cast(bf16x8, ac[2 * ks]);
        al = __builtin_bit_cast(bf16x8, ac[2 * ks + 1]);
      } else {
        unpack8(__builtin_bit_cast(u32x4, ac[2 * ks]), __builtin_bit_cast(u32x4, ac[2 * ks + 1]), ah, al);
      }
      const u32x4* wt = &wl[cur][ks * 512 + lane];
      if constexpr (F & 131072) {
        ah = __builtin_bit_cast(bf16x8, wt[((ks + 1) & 3) * 64]);
        al = __builtin_bit_cast(bf16x8, wt[((ks + 2) & 3) * 64 + 256]);
      }
#pragma unroll
      for (int nt4 = 0; nt4 < ((F & 16384) ? 4 : ((F & 8192) ? 1 : ((F & 32768) ? 2 : 4))); ++nt4) {
        const int nt = (F & 16384) ? 0 : nt4;
        bf16x8 bh, bl;
        if constexpr (F & 2) {  // no LDS: some fragment of the prefetch registers (the values do not matter here)
          bh = __builtin_bit_cast(bf16x8, wcur[nt]);
          bl = __builtin_bit_cast(bf16x8, wcur[(nt + ks) & 3]);
        } else if constexpr (F & 512) {
          const u32x4* base = &wl[cur][0];
          bh = __builtin_bit_cast(bf16x8, base[(ks * 512 + (nt * 2) * 64 + lane * 17) & 2047]);
          bl = __builtin_bit_cast(bf16x8, base[(ks * 512 + (nt * 2 + 1) * 64 + lane * 17 + 5) & 2047]);
        } else {
          bh = __builtin_bit_cast(bf16x8, wt[(nt4 * 2) * 64]);
          bl = __builtin_bit_cast(bf16x8, wt[(nt4 * 2 + 1) * 64]);
        }
        if constexpr (F & 1024) {
          const u32x4 t0 = __builtin_bit_cast(u32x4, bh), t1 = __builtin_bit_cast(u32x4, bl);
          side += __uint_as_float((t0[0] ^ t1[3]) & 0x3f7fffffu);      // the LDS values are consumed, but not by an MFMA
          bh = __builtin_bit_cast(bf16x8, wcur[nt]);
          bl = __builtin_bit_cast(bf16x8, wcur[(nt + ks) & 3]);
        }
        acc[nt] = prod<F>(ah, bh, acc[nt]);
        if constexpr (!(F & 64)) {
          acc[nt] = prod<F>(al, bh, acc[nt]);
          acc[nt] = prod<F>(ah, bl, acc[nt]);
        }
      }
    }
    if constexpr (F & 256) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          side += acc[t][r];
          acc[t][r] = 0.f;
        }
    }
    if constexpr (!(F & 2)) {
      if constexpr (!(F & 2048)) {
        if (st + 1 < nstage) {
#pragma unroll
          for (int q = 0; q < 4; ++q) wl[cur ^ 1][tid + 512 * q] = wreg[q];
        }
      }
      if constexpr (!((F & 65536) && (F & 2048))) __syncthreads();
      else asm volatile("" ::: "memory");   // no s_barrier, but the LDS reads must stay inside the loop
    }
  }

  if constexpr (F & 16) {
    float s = side;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[t][r];
    const long long pos = posl[w][i];
    if (active && pos >= 0 && half == 0) p.C[pos * p.ldc] = s;
    return;
  }
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
        stg[m * 64 + t * 32 + i] = acc[nt][r] + bv + side;
      }
    }
    __syncthreads();
    const int c4 = lane & 15, r0 = lane >> 4;
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
    __syncthreads();
  }
}

template <int F>
__global__ __launch_bounds__(512, 2) void clone2(const ws_gemm_b2p_args p) {
  clone_body<F>(p);
}
template <int F>
__global__ __launch_bounds__(512, 1) void clone1(const ws_gemm_b2p_args p) {
  clone_body<F>(p);
}

// variants built (flag sets without bit 7; bit 7 selects the launch bounds)
#define CLONE_VARIANTS(X) \
  X(0) X(1) X(2) X(4) X(8) X(16) X(32) X(64) X(256) X(12) X(3) X(17) X(18) X(20) X(24) X(28) X(30) X(31) X(48) X(80) X(272) X(19) \
  X(6) X(10) X(14) X(22) X(26) X(92) X(124) X(380) X(348) X(316) \
  X(540) X(1052) X(2076) X(4124) X(6172) X(4096) X(4112) X(2064) X(1040) X(528) X(7196) X(5148) X(3100) X(4126) X(8220) X(12316) X(8284) X(12380) X(24604) X(28700) X(32796) X(36892) X(67612) X(71708) X(135196) X(133148) X(202780) X(198684)

extern "C" int b2p_clone_launch(int flags, const ws_gemm_b2p_args* a, hipStream_t s) {
  const int nblk = ((a->sm.nseq + 31) / 32) * a->sm.L;
  int cus = 256;
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  const dim3 grid((flags & 4096) ? 2 * cus : (nblk + 7) / 8), block(512);
  const int f = flags & ~128;
#define X(F)                                                                   \
  if (f == F) {                                                                \
    if (flags & 128) hipLaunchKernelGGL(clone1<F>, grid, block, 0, s, *a);     \
    else hipLaunchKernelGGL(clone2<F>, grid, block, 0, s, *a);                 \
    return 0;                                                                  \
  }
  CLONE_VARIANTS(X)
#undef X
  return 1;  // variant not built
}
