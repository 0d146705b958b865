// Band-view forward recurrence with the INPUT PROJECTION FUSED IN (blocked layout, 32 sequences per
// workgroup).  The plain pipeline writes x W_ih^T + b to the 16E-byte `gates` buffer (gemm_p2b, 4.2 GB
// at R = 32) and the recurrence reads it back; both kernels are HBM-bound.  Here the recurrence takes
// the normalised input itself (BL(128), 1/16 of the bytes) and streams [W_ih | W_hh] as one K = 384
// weight stream: the x part (8 of 24 k-steps) does not depend on h_{t-1}, the matrix cores were at
// ~23 % in this view, and the per-step weight stream (1.5 MB at ~64 B/clk/CU = 10 us) stays under the
// HBM time of the step's 288 KB of stores at 256 workgroups.  Everything else (transposed product,
// lane-local cell update, 2-slot weight ring that never drains, single-basic-block step body,
// non-temporal activation streams) is lstm_fwd_bf16_kernel<BLK = true> of lstm_bf16.hip.
//   nn.LSTM forward inside ResRNN (bsrnn.py:27-33,40); replaces gemm_p2b(x-proj) + ws_lstm_fwd there.
#include <stdlib.h>

#include "lstm_bf16_common.h"

#define XROW 136  // bf16 per LDS row of x (128 + 8: 272 B = 4 banks mod 64)
#define FKS 24    // k-steps of the fused stream: 8 of W_ih (K = 128), 16 of W_hh (K = 256)
#define H8ROW 272 // bytes per LDS row of e4m3 h (256 + 16: 4 banks mod 64)

// unit ((((d*8 + w)*24 + ks)*4 + g)*2 + part)*64 + lane, element j =
//   part( ks < 8 ? W_ih[d][g*256 + 32w + (lane&31)][16ks + 8(lane>>5) + j]
//                : W_hh[d][g*256 + 32w + (lane&31)][16(ks-8) + 8(lane>>5) + j] )
// H16 (ws_lstm_fused_args.hfmt = 1, ABI v19): both parts carry 256 w -- the W_ih part (ks < 8) as bf16 hi / lo (x arrives as bf16
// split pairs), the W_hh part as fp16 hi / lo: the A operand of v_mfma_f32_32x32x16_f16 against h as ONE fp16 value
template <bool H16>
__global__ void lstm_pack_fused_kernel(const float* __restrict__ wih_f, const float* __restrict__ wih_r,
                                       const float* __restrict__ whh_f, const float* __restrict__ whh_r,
                                       __bf16* __restrict__ pf) {
  const int total = 2 * 8 * FKS * 4 * 64 * 8;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    int r = idx;
    const int j = r & 7; r >>= 3;
    const int lane = r & 63; r >>= 6;
    const int g = r & 3; r >>= 2;
    const int ks = r % FKS; r /= FKS;
    const int w = r & 7; r >>= 3;
    const int d = r;
    const int row = g * 256 + 32 * w + (lane & 31);
    float v;
    if (ks < 8) {
      const float* W = d ? wih_r : wih_f;
      v = W[row * 128 + 16 * ks + 8 * (lane >> 5) + j];
    } else {
      const float* W = d ? whh_r : whh_f;
      v = W[row * LH + 16 * (ks - 8) + 8 * (lane >> 5) + j];
    }
    const long long unit = ((((long long)(d * 8 + w) * FKS + ks) * 4 + g) * 2) * 64 + lane;
    if (H16 && ks >= 8) {
      _Float16* ph = reinterpret_cast<_Float16*>(pf);
      const float s = 256.f * v;
      const _Float16 hi = (_Float16)s;
      ph[unit * 8 + j] = hi;
      ph[(unit + 64) * 8 + j] = (_Float16)(s - (float)hi);
    } else {
      const float s = H16 ? 256.f * v : v;
      const __bf16 hi = (__bf16)s;
      pf[unit * 8 + j] = hi;
      pf[(unit + 64) * 8 + j] = (__bf16)(s - (float)hi);
    }
  }
}

extern "C" int ws_lstm_pack_fused(const float* wih_f, const float* wih_r, const float* whh_f, const float* whh_r,
                                  float* pack, void* stream) {
  WS_REQUIRE(wih_f && wih_r && whh_f && whh_r && pack, "ws_lstm_pack_fused: null pointer");
  hipLaunchKernelGGL(lstm_pack_fused_kernel<false>, dim3(512), dim3(256), 0, (hipStream_t)stream, wih_f, wih_r, whh_f, whh_r,
                     reinterpret_cast<__bf16*>(pack));
  return ws_check_launch("ws_lstm_pack_fused");
}

extern "C" int ws_lstm_pack_fused_h16(const float* wih_f, const float* wih_r, const float* whh_f, const float* whh_r,
                                      float* pack, void* stream) {
  WS_REQUIRE(wih_f && wih_r && whh_f && whh_r && pack, "ws_lstm_pack_fused_h16: null pointer");
  hipLaunchKernelGGL(lstm_pack_fused_kernel<true>, dim3(512), dim3(256), 0, (hipStream_t)stream, wih_f, wih_r, whh_f, whh_r,
                     reinterpret_cast<__bf16*>(pack));
  return ws_check_launch("ws_lstm_pack_fused_h16");
}

// ---- hfmt = 5 (ABI v20): the lo plane of W_hh as FP8 (e4m3) operands of v_mfma_scale_f32_32x32x64_f8f6f4 -----------------
// Region of (d, w): F8_REGION bytes.  [0, 64 KB): the W_ih k-steps exactly as in the H16 pack.  Then 16 recurrent k-steps q of
// 6 KB: four fp16 hi fragments (gate g at + 1 KB g; lane: 8 values k = 16 q + 8 (lane >> 5) + j of row g*256 + 32 w + (lane & 31))
// and, at + 4 KB, ONE FP8 fragment of the residuals 256 w - hi: gate q & 3, k block q >> 2 (k = 64 (q >> 2) .. + 63), a lane's 32
// bytes in the operand order of the instruction (profiles/r06_c20_f8_probe.txt): piece 0 (+ 0, lane * 16) = k 16 (lane >> 5) + j,
// piece 1 (+ 1 KB) = k 32 + 16 (lane >> 5) + j.  The fragment's codes are res * 2^-E with ONE exponent E per fragment (the
// largest residual lands in [128, 256)); the E8M0 byte 127 + E of fragment q is byte q of the 16 at F8_SCALES.
#define F8_HPART 65536
#define F8_KSTEP 6144
#define F8_SCALES (F8_HPART + 16 * F8_KSTEP)
#define F8_REGION (F8_SCALES + 64)
__global__ void lstm_pack_fused_h8_hi_kernel(const float* __restrict__ wih_f, const float* __restrict__ wih_r,
                                             const float* __restrict__ whh_f, const float* __restrict__ whh_r,
                                             unsigned char* __restrict__ pk) {
  const int total = 2 * 8 * FKS * 4 * 64 * 8;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    int r = idx;
    const int j = r & 7; r >>= 3;
    const int lane = r & 63; r >>= 6;
    const int g = r & 3; r >>= 2;
    const int ks = r % FKS; r /= FKS;
    const int w = r & 7; r >>= 3;
    const int d = r;
    const int row = g * 256 + 32 * w + (lane & 31);
    unsigned char* reg = pk + (long long)(d * 8 + w) * F8_REGION;
    if (ks < 8) {
      const float s = 256.f * (d ? wih_r : wih_f)[row * 128 + 16 * ks + 8 * (lane >> 5) + j];
      const __bf16 hi = (__bf16)s;
      __bf16* o = reinterpret_cast<__bf16*>(reg + ks * 8192 + g * 2048 + lane * 16);
      o[j] = hi;
      o[512 + j] = (__bf16)(s - (float)hi);     // + 1 KB: the lo fragment
    } else {
      const float s = 256.f * (d ? whh_r : whh_f)[row * LH + 16 * (ks - 8) + 8 * (lane >> 5) + j];
      reinterpret_cast<_Float16*>(reg + F8_HPART + (ks - 8) * F8_KSTEP + g * 1024 + lane * 16)[j] = (_Float16)s;
    }
  }
}
// one wave per FP8 fragment: (d, w, q)
__global__ __launch_bounds__(64) void lstm_pack_fused_h8_lo_kernel(const float* __restrict__ whh_f, const float* __restrict__ whh_r,
                                                                   unsigned char* __restrict__ pk) {
  const int lane = threadIdx.x, q = blockIdx.x & 15, w = (blockIdx.x >> 4) & 7, d = blockIdx.x >> 7;
  const int g = q & 3, kb = q >> 2;
  const float* W = (d ? whh_r : whh_f) + (long long)(g * 256 + 32 * w + (lane & 31)) * LH + 64 * kb + 16 * (lane >> 5);
  float res[32];
  float mx = 0.f;
#pragma unroll
  for (int j = 0; j < 32; ++j) {
    const float s = 256.f * W[(j >> 4) * 32 + (j & 15)];
    res[j] = s - (float)(_Float16)s;
    mx = fmaxf(mx, fabsf(res[j]));
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  int E = 0;
  if (mx > 0.f) E = ((__float_as_int(mx) >> 23) & 255) - 127 - 7;       // floor(log2 mx) - 7: the largest code in [128, 256)
  E = max(E, -126);                                                     // (residuals of denormal size: codes of zero)
  const float inv = __int_as_float((127 - E) << 23);
  unsigned char* reg = pk + (long long)(d * 8 + w) * F8_REGION;
  unsigned int* o = reinterpret_cast<unsigned int*>(reg + F8_HPART + q * F8_KSTEP + 4096 + lane * 16);
#pragma unroll
  for (int j = 0; j < 32; j += 4) {
    int v = 0;
    v = __builtin_amdgcn_cvt_pk_fp8_f32(res[j] * inv, res[j + 1] * inv, v, false);
    v = __builtin_amdgcn_cvt_pk_fp8_f32(res[j + 2] * inv, res[j + 3] * inv, v, true);
    o[(j >> 4) * 256 + ((j & 15) >> 2)] = (unsigned int)v;
  }
  if (lane == 0) reg[F8_SCALES + q] = (unsigned char)(127 + E);
}

extern "C" int ws_lstm_pack_fused_h8(const float* wih_f, const float* wih_r, const float* whh_f, const float* whh_r,
                                     float* pack, void* stream) {
  WS_REQUIRE(wih_f && wih_r && whh_f && whh_r && pack, "ws_lstm_pack_fused_h8: null pointer");
  static_assert(2 * 8 * F8_REGION <= WS_LSTM_FUSED_PACK_FLOATS * 4, "the F8 stream fits the pack of the other formats");
  unsigned char* pk = reinterpret_cast<unsigned char*>(pack);
  hipLaunchKernelGGL(lstm_pack_fused_h8_hi_kernel, dim3(512), dim3(256), 0, (hipStream_t)stream, wih_f, wih_r, whh_f, whh_r, pk);
  hipLaunchKernelGGL(lstm_pack_fused_h8_lo_kernel, dim3(2 * 8 * 16), dim3(64), 0, (hipStream_t)stream, whh_f, whh_r, pk);
  return ws_check_launch("ws_lstm_pack_fused_h8");
}

// H16: ws_lstm_fused_args.hfmt = 1 -- see lstm_fwd_fused64_body (fp16 h, one LDS plane per buffer, two recurrent terms)
// F8 (hfmt 5, with H16): the lo term on the FP8 matrix instruction, as in lstm_fwd_fused64_body; the e4m3 image of h lives in the
// part-1 plane of each buffer (unused by the fp16 format)
template <int GF, bool H16 = false, bool F8 = false>  // WS_GATES_*: != 0 -> activated gates leave as unorm16 (BLH), lstm_bf16_common.h
__global__ __launch_bounds__(512, 1) void lstm_fwd_fused_kernel(const ws_lstm_fused_args p) {
  static_assert(!F8 || H16, "the FP8 lo term belongs to the fp16-h kernel");
  __shared__ __attribute__((aligned(16))) __bf16 hl[2][2][SQ * HROW];  // [buf][part][seq][k]  66 KB
  __shared__ __attribute__((aligned(16))) __bf16 xl[2][2][SQ * XROW];  // [buf][part][seq][k]  34 KB
  __shared__ __attribute__((aligned(16))) float cl[SQ * (LH + 4)];     // cell state [seq][unit] 33 KB
  const int d = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int L = p.L;

  {  // h_{-1} = 0
    uint32_t* z = reinterpret_cast<uint32_t*>(&hl[0][0][0]);
    for (int i = tid; i < 2 * SQ * HROW / 2; i += 512) z[i] = 0u;
  }
  const int ubase = 32 * w + 4 * half;  // unit of register 4j + r: ubase + 8j + r
  const int glane = ((d * 256 + 8 * w + half) * 32 + l31) * 16;  // bytes; + (g*64 + 2j)*512
  const int clane = ((d * 64 + 8 * w + half) * 32 + l31) * 16;   // bytes; + 2j*512
  auto grs = [&](int t) { return mkrsrc(p.gates + (long long)(blockIdx.x * L + t) * (SQ * 2 * LG), SQ * 2 * LG * 4); };
  auto crs = [&](float* b, int t) { return mkrsrc(b + (long long)(blockIdx.x * L + t) * (SQ * 2 * LH), SQ * 2 * LH * 4); };
  auto xrs = [&](int t) { return mkrsrc(p.xn + (long long)(blockIdx.x * L + t) * (SQ * 128), SQ * 128 * 4); };
  auto hrs = [&](int t) { return mkrsrc(p.gates + (long long)(blockIdx.x * L + t) * (SQ * LG), SQ * 2 * LG * 2); };  // BLH
  auto st_gate = [&](const f32x4& v, int t, int g, int j) {
    if constexpr (GF != 0) bst8(g == 2 ? enc_u16x4<true>(v) : enc_u16x4<false>(v), hrs(t), glane >> 1, (g * 64 + 2 * j) * 256);
    else bst(v, grs(t), glane, (g * 64 + 2 * j) * 512);
  };
  auto st_ch = [&](const f32x4& v, float* b, int t, int j) { bst(v, crs(b, t), clane, 2 * j * 512); };
  // x tile of one step = one BL(128) block: 1024 cells of 16 B, cell u = quad * 32 + slot; thread: u = tid, tid + 512
  auto ld_x = [&](int t, int q) -> f32x4 { return bld(xrs(t), tid * 16, q * 8192); };
  auto st_x = [&](const f32x4& v, int buf, int q) {  // cell u -> row slot, columns 4*quad .. +3
    const int u = tid + 512 * q, quad = u >> 5, slot = u & 31;
    bf16x4 hi, lo;
    unpack_hl4(v, hi, lo);  // xn arrives as split pairs (BLS, written by gemm_p2b)
    *reinterpret_cast<bf16x4*>(&xl[buf][0][slot * XROW + 4 * quad]) = hi;
    *reinterpret_cast<bf16x4*>(&xl[buf][1][slot * XROW + 4 * quad]) = lo;
  };

  float* cme = &cl[l31 * (LH + 4) + ubase];  // this lane's 4 runs of 4 units: cme + 8j (private)
#pragma unroll
  for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(cme + 8 * j) = f32x4{0.f, 0.f, 0.f, 0.f};

  // biases b_ih + b_hh of this lane's units: the accumulators start from them
  f32x4 bias[4][4];  // [gate][run]
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      bias[g][j] = *reinterpret_cast<const f32x4*>(p.bias + d * LG + g * 256 + ubase + 8 * j) * (H16 ? 256.f : 1.f);

  // weight stream: per k-step 8 fragments (4 gates x {hi, lo}), 1 KB each per wave; 24 k-steps per step
  const __amdgpu_buffer_rsrc_t wrs =
      F8 ? __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<unsigned char*>(const_cast<float*>(p.wpack)) +
                                                 (long long)(d * 8 + w) * F8_REGION, 0, F8_REGION, 0x00020000)
         : __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wpack) + (long long)(d * 8 + w) * (FKS * 8 * 64 * 4), 0,
                                             FKS * 8 * 1024, 0x00020000);
  const int wlane = lane * 16;
  bf16x8 wr[2][8];
  auto refill = [&](int s, int kn, int zo) {   // k-step kn of the stream -> ring slot s (F8: six fragments per recurrent k-step)
#pragma unroll
    for (int f = 0; f < 8; ++f) {
      if (F8 && kn >= 8) {
        if (f < 6) wr[s][f] = wload(wrs, wlane + f * 1024, zo + F8_HPART + (kn - 8) * F8_KSTEP);
      } else {
        wr[s][f] = wload(wrs, wlane + (f & 3) * 1024, zo + kn * 8192 + (f >> 2) * 4096);
      }
    }
  };
  refill(0, 0, 0);
  refill(1, 1, 0);
  int ssc[4] = {0, 0, 0, 0};   // F8: the sixteen fragment exponents of this wave's stream
  if constexpr (F8) {
    const int* sp = reinterpret_cast<const int*>(reinterpret_cast<const unsigned char*>(p.wpack) +
                                                 (long long)(d * 8 + w) * F8_REGION + F8_SCALES);
#pragma unroll
    for (int i = 0; i < 4; ++i) ssc[i] = __builtin_amdgcn_readfirstlane(sp[i]);
  }

  // inputs: x of the first step goes to LDS now, x of the second step waits in registers
  f32x4 xr[2];
  {
    const int t0 = d == 0 ? 0 : L - 1;
    const int t1 = d == 0 ? min(1, L - 1) : max(L - 2, 0);
#pragma unroll
    for (int q = 0; q < 2; ++q) st_x(ld_x(t0, q), 0, q);
#pragma unroll
    for (int q = 0; q < 2; ++q) xr[q] = ld_x(t1, q);
  }
  __syncthreads();

  for (int step = 0; step < L; ++step) {
    const int t = d == 0 ? step : L - 1 - step;
    const int cur = step & 1;
    int zo = 0;
    asm volatile("" : "+s"(zo));
    const __bf16* xhi = &xl[cur][0][l31 * XROW + 8 * half];
    const __bf16* xlo = &xl[cur][1][l31 * XROW + 8 * half];
    const __bf16* hhi = &hl[cur][0][l31 * HROW + 8 * half];
    const __bf16* hlo = &hl[cur][1][l31 * HROW + 8 * half];
    f32x16 acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[g][4 * j + r] = bias[g][j][r];
    const unsigned char* h8r = reinterpret_cast<const unsigned char*>(&hl[cur][1][0]) + l31 * H8ROW + 16 * half;
    v8i b8;
    int scv = 0;
#pragma unroll
    for (int ks = 0; ks < FKS; ++ks) {
      const int s = ks & 1;
      const bf16x8 bh = *reinterpret_cast<const bf16x8*>(ks < 8 ? xhi + 16 * ks : hhi + 16 * (ks - 8));
      if (F8 && ks >= 8) {    // fp16 hi of 256 W_hh x fp16 h + ONE FP8 MFMA per k-step (gate q & 3 of k block q >> 2)
        const int q = ks - 8;
        if ((q & 3) == 0) {
          const i32x4 p0 = *reinterpret_cast<const i32x4*>(h8r + 64 * (q >> 2)), p1 = *reinterpret_cast<const i32x4*>(h8r + 64 * (q >> 2) + 32);
          b8 = __builtin_shufflevector(p0, p1, 0, 1, 2, 3, 4, 5, 6, 7);
          scv = ssc[q >> 2];
        }
        const f16x8 b16 = __builtin_bit_cast(f16x8, bh);
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = mfma16h(__builtin_bit_cast(f16x8, wr[s][g]), b16, acc[g]);
        const v8i a8 = __builtin_shufflevector(__builtin_bit_cast(i32x4, wr[s][4]), __builtin_bit_cast(i32x4, wr[s][5]), 0, 1, 2, 3, 4, 5, 6, 7);
        if ((q & 3) == 0) acc[0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[0], 0, 0, 0, scv, 0, 127);
        if ((q & 3) == 1) acc[1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[1], 0, 0, 1, scv, 0, 127);
        if ((q & 3) == 2) acc[2] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[2], 0, 0, 2, scv, 0, 127);
        if ((q & 3) == 3) acc[3] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[3], 0, 0, 3, scv, 0, 127);
      } else if (H16 && ks >= 8) {   // fp16 W_hh (hi, lo of 256 w) x fp16 h: two terms
        const f16x8 b16 = __builtin_bit_cast(f16x8, bh);
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = mfma16h(__builtin_bit_cast(f16x8, wr[s][2 * g]), b16, acc[g]);
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = mfma16h(__builtin_bit_cast(f16x8, wr[s][2 * g + 1]), b16, acc[g]);
      } else {
        const bf16x8 bl = *reinterpret_cast<const bf16x8*>(ks < 8 ? xlo + 16 * ks : hlo + 16 * (ks - 8));
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = mfma32(wr[s][2 * g], bh, acc[g]);
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = mfma32(wr[s][2 * g + 1], bh, acc[g]);
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = mfma32(wr[s][2 * g], bl, acc[g]);
      }
      // refill this slot with k-step ks+2 (wraps into the next step: the stream never drains)
      refill(s, (ks + 2) % FKS, zo);
      __builtin_amdgcn_sched_barrier(0);  // keep the k-steps in program order: loads stay 2 k-steps ahead
    }

    // next step's x (loaded one step ago) -> LDS; x of the step after it -> registers (two steps of HBM latency
    // budget, and only 2 loads per thread sit in front of the weight stream of the next step)
    {
      const int s2 = min(step + 2, L - 1);
      const int t2 = d == 0 ? s2 : L - 1 - s2;
#pragma unroll
      for (int q = 0; q < 2; ++q) st_x(xr[q], cur ^ 1, q);
#pragma unroll
      for (int q = 0; q < 2; ++q) xr[q] = ld_x(t2, q);
    }
    __builtin_amdgcn_sched_barrier(0);

    // cell update, lane-local; global traffic as 16-byte vectors
    __bf16* nhi = &hl[cur ^ 1][0][l31 * HROW + ubase];
    __bf16* nlo = &hl[cur ^ 1][1][l31 * HROW + ubase];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x4 vi, vf, vg, vo, vc, vh;
      const f32x4 cold = *reinterpret_cast<const f32x4*>(cme + 8 * j);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float ig = H16 ? c2_sig256(acc[0][4 * j + r]) : fsig(acc[0][4 * j + r]);
        const float fg = H16 ? c2_sig256(acc[1][4 * j + r]) : fsig(acc[1][4 * j + r]);
        const float gg = H16 ? c2_tanh256(acc[2][4 * j + r]) : ftanh(acc[2][4 * j + r]);
        const float og = H16 ? c2_sig256(acc[3][4 * j + r]) : fsig(acc[3][4 * j + r]);
        const float cn = fg * cold[r] + ig * gg;
        vi[r] = ig;
        vf[r] = fg;
        vg[r] = gg;
        vo[r] = og;
        vc[r] = cn;
        vh[r] = og * ftanh(cn);
      }
      bf16x4 h_hi, h_lo;
      split4(vh, h_hi, h_lo);
      if constexpr (H16) {   // the recurrent operand: fp16(h), the hi plane's storage
        *reinterpret_cast<u32x2*>(nhi + 8 * j) = enc_f16x4(vh);
        if constexpr (F8) {  // + its e4m3 image in the buffer's other plane
          int c8 = 0;
          c8 = __builtin_amdgcn_cvt_pk_fp8_f32(vh[0], vh[1], c8, false);
          c8 = __builtin_amdgcn_cvt_pk_fp8_f32(vh[2], vh[3], c8, true);
          *reinterpret_cast<int*>(reinterpret_cast<unsigned char*>(&hl[cur ^ 1][1][0]) + l31 * H8ROW + ubase + 8 * j) = c8;
        }
      } else {
        *reinterpret_cast<bf16x4*>(nhi + 8 * j) = h_hi;
        *reinterpret_cast<bf16x4*>(nlo + 8 * j) = h_lo;
      }
      *reinterpret_cast<f32x4*>(cme + 8 * j) = vc;
      st_gate(vi, t, 0, j);
      st_gate(vf, t, 1, j);
      st_gate(vg, t, 2, j);
      st_gate(vo, t, 3, j);
      st_ch(vc, p.cbuf, t, j);
      st_ch(pack_hl4(h_hi, h_lo), p.hcat, t, j);  // BLS
      __builtin_amdgcn_sched_barrier(0);  // one run at a time: bounds the live temporaries
    }
    __syncthreads();
  }
}


// ---------------------------------------------------------------------------------------------
// 64 sequences per workgroup (round 3): two 32-sequence tiles share every weight fragment of the K = 384 stream.
// The 32-sequence kernel above is bound by that stream (1.5 MB per step through the CU's L1, ~10 us, MFMA 7.7 us
// under it, then the cell update and the stores: 21.6 us per step, nothing overlaps with one workgroup per CU); with
// two tiles per fragment the stream is paid once per 64 sequences and the step becomes MFMA-bound.
// Measured (band view, R = 32): 2.75 -> 2.48 ms per launch, not the 1.6x the serial model promised: the pass of the
// weight stream itself takes ~21 us whatever sits around it (a software pipeline that hid tile 1's cell update under
// tile 0's MFMAs, streaming the weights once per tile, measured 2.91 ms), i.e. the stream is bound by the L2 -> CU path
// at the ring depth the registers allow (16 KB in flight per wave), and sharing it between two tiles is what pays.
// Budget (8 waves x 256 registers, 160 KB of LDS): accumulators of both tiles 128 registers, weight ring 64;
// h single-buffered (two barriers per step instead of one), the x tiles arrive RAW (BLS) by LDS DMA
// (buffer_load ... lds: no staging registers) and are unpacked into MFMA fragments by the reading wave, the cell state
// of tile 0 lives in LDS (lane-private, unpadded), of tile 1 in registers, the bias in LDS.
// Per sequence the arithmetic and its order are those of the 32-sequence kernel: results are bit-identical.
// ---------------------------------------------------------------------------------------------
// one object, members in this order: the DMA target sits at LDS address 0 (its base travels in M0)
// (a named type: with an unnamed struct inside the kernel TEMPLATE the host stub of the instantiations is not emitted)
struct fused64_lds {
  f32x4 xw[2][1024];            // [tile][quad*32 + slot] raw BLS      32 KB
  __bf16 hl[2][2 * SQ * HROW];  // [part][tile*32 + seq][k]            67.6 KB
  f32x4 c0l[8 * 2 * 4 * 32];    // tile 0 cells [w][half][run][seq]    32 KB
  float bs[LG];                 // b_ih + b_hh of this direction        4 KB
};
// (the body is a device function template behind two plain kernels: hipcc 7.2 does not emit the host stubs of a
//  __global__ TEMPLATE with this body -- "substitution failure" without a diagnostic)
// W1 (MEASUREMENT ONLY, WS_FUSED_W1=1): one weight plane -- the W_lo x_hi term and the lo
// fragments of the weight stream are dropped (two MFMAs per product, half the L2 -> CU stream): weights at bf16 precision.
// H16 (ws_lstm_fused_args.hfmt = 1, ABI v19; round 6): the recurrent part of the stream on v_mfma_f32_32x32x16_f16 -- h in (-1, 1) as
// ONE fp16 operand (one LDS plane), W_hh as fp16 hi / lo of 256 w: two MFMAs per product instead of three (448 instead of 576 per
// wave and step); the x part keeps the full three-term split product on 256 W_ih, the accumulators carry 256 x the
// pre-activation and the 2^-8 leaves in the activations' exponent scale.  The arithmetic ws_lstm_fwd_cluster2 runs in the time
// view since round 5 (there the 60-step trajectory did not move with it; the fp16 INPUT did, which is why x keeps its pairs).
// F8 (hfmt = 5, ABI v20; with H16): the lo term of the recurrent product on the block-scaled FP8 matrix instruction --
// 256 w = hi (fp16) + res; res x h is 2^-12 of the product, so e4m3 operands (res as codes with one exponent per fragment, h as
// e4m3 of the same h the fp16 plane holds) leave it good to 2^-16 of the whole, and ONE v_mfma_scale_f32_32x32x64_f8f6f4 (K = 64 at
// twice the fp16 rate: 4 776 against 2 184 TFLOP/s measured, profiles/r06_c20_f8_probe.txt) replaces four fp16 MFMAs of K = 16:
// per recurrent k-step four hi MFMAs + one FP8 MFMA (gate q & 3 of k block q >> 2) instead of eight, 6 KB of stream instead of 8.
template <int GF, bool W1 = false, bool H16 = false, bool F8 = false>
__device__ __forceinline__ void lstm_fwd_fused64_body(const ws_lstm_fused_args& p) {
  static_assert(!(W1 && H16), "W1 is a measurement build of the three-term kernel");
  static_assert(!F8 || H16, "the FP8 lo term belongs to the fp16-h kernel");
  __shared__ __attribute__((aligned(16))) fused64_lds sm;
  auto& xw = sm.xw;
  auto& hl = sm.hl;
  auto& c0l = sm.c0l;
  auto& bs = sm.bs;
  const int d = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int L = p.L;
  const int ntile = (p.nseq + SQ - 1) / SQ;
  const int tile0 = 2 * blockIdx.x;
  // an odd tile count leaves the last workgroup's second tile empty: zero-sized descriptors (loads 0, stores dropped)
  const unsigned live1 = tile0 + 1 < ntile ? 1u : 0u;

  {
    uint32_t* z = reinterpret_cast<uint32_t*>(&hl[0][0]);
    for (int i = tid; i < 2 * 2 * SQ * HROW / 2; i += 512) z[i] = 0u;  // h_{-1} = 0
    for (int i = tid; i < LG; i += 512) bs[i] = (H16 ? 256.f : 1.f) * p.bias[d * LG + i];
  }
  const int ubase = 32 * w + 4 * half;  // unit of register 4j + r: ubase + 8j + r
  const int glane = ((d * 256 + 8 * w + half) * 32 + l31) * 16;  // bytes; + (g*64 + 2j)*512
  const int clane = ((d * 64 + 8 * w + half) * 32 + l31) * 16;   // bytes; + 2j*512
  auto blk = [&](int e, int t) { return (long long)((tile0 + e) * L + t); };
  auto lim = [&](int e, unsigned bytes) { return e == 0 ? bytes : bytes * live1; };
  auto grs = [&](int e, int t) { return mkrsrc(p.gates + blk(e, t) * (SQ * 2 * LG), lim(e, SQ * 2 * LG * 4)); };
  auto hrs = [&](int e, int t) { return mkrsrc(p.gates + blk(e, t) * (SQ * LG), lim(e, SQ * 2 * LG * 2)); };  // BLH
  auto st_gate = [&](const f32x4& v, int e, int t, int g, int j) {
    if constexpr (GF != 0) bst8(g == 2 ? enc_u16x4<true>(v) : enc_u16x4<false>(v), hrs(e, t), glane >> 1, (g * 64 + 2 * j) * 256);
    else bst(v, grs(e, t), glane, (g * 64 + 2 * j) * 512);
  };
  auto crs = [&](float* b, int e, int t) { return mkrsrc(b + blk(e, t) * (SQ * 2 * LH), lim(e, SQ * 2 * LH * 4)); };
  auto xrs = [&](int e, int t) { return mkrsrc(p.xn + blk(e, t) * (SQ * 128), lim(e, SQ * 128 * 4)); };
  // x tile of one step = one BL(128) block of 16 KB: wave w copies cells 64w .. 64w+63 and 512 + 64w .. of each tile
  auto dma_x = [&](int e, int t) {
#pragma unroll
    for (int q = 0; q < 2; ++q)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs(e, t), (__attribute__((address_space(3))) void*)&xw[e][512 * q + 64 * w], 16,
                                               lane * 16, (512 * q + 64 * w) * 16, 0, WS_STREAM_AUX);
  };

  f32x4* c0 = &c0l[((w * 2 + half) * 4) * 32 + l31];  // + 32 * j
#pragma unroll
  for (int j = 0; j < 4; ++j) c0[32 * j] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 c1[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) c1[j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const __amdgpu_buffer_rsrc_t wrs =
      F8 ? __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<unsigned char*>(const_cast<float*>(p.wpack)) +
                                                 (long long)(d * 8 + w) * F8_REGION, 0, F8_REGION, 0x00020000)
         : __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wpack) + (long long)(d * 8 + w) * (FKS * 8 * 64 * 4), 0,
                                             FKS * 8 * 1024, 0x00020000);
  const int wlane = lane * 16;
  bf16x8 wr[2][8];
  // k-step kn of the stream -> ring slot s (F8: the recurrent k-steps are 6 fragments -- four hi, two halves of the FP8 one)
  auto refill = [&](int s, int kn, int zo) {
#pragma unroll
    for (int f = 0; f < 8; ++f) {
      if (F8 && kn >= 8) {
        if (f < 6) wr[s][f] = wload(wrs, wlane + f * 1024, zo + F8_HPART + (kn - 8) * F8_KSTEP);
      } else if (!W1 || !(f & 1)) {
        wr[s][f] = wload(wrs, wlane + (f & 3) * 1024, zo + kn * 8192 + (f >> 2) * 4096);
      }
    }
  };
  refill(0, 0, 0);
  refill(1, 1, 0);
  // F8: the sixteen fragment exponents (E8M0 bytes) of this wave's stream, four per k block, in scalar registers
  int ssc[4] = {0, 0, 0, 0};
  if constexpr (F8) {
    const int* sp = reinterpret_cast<const int*>(reinterpret_cast<const unsigned char*>(p.wpack) +
                                                 (long long)(d * 8 + w) * F8_REGION + F8_SCALES);
#pragma unroll
    for (int i = 0; i < 4; ++i) ssc[i] = __builtin_amdgcn_readfirstlane(sp[i]);
  }
  // e4m3 h_{t-1}: rows of 272 B in the LDS plane the fp16 format leaves unused ([tile*32 + seq][k], 17 KB)
  unsigned char* h8 = reinterpret_cast<unsigned char*>(&hl[1][0]);
  const unsigned char* h8row[2] = {h8 + l31 * H8ROW + 16 * half, h8 + (32 + l31) * H8ROW + 16 * half};

  const __bf16* hrow[2] = {&hl[0][l31 * HROW + 8 * half], &hl[0][(32 + l31) * HROW + 8 * half]};  // h_{t-1} rows of this lane
  auto tof = [&](int n) { const int m = min(n, L - 1); return d == 0 ? m : L - 1 - m; };
  dma_x(0, tof(0));
  dma_x(1, tof(0));
  __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0) lgkmcnt(0): this wave's share of x(0) has landed (and the ring's first fragments)
  __syncthreads();

  for (int step = 0; step < L; ++step) {
    const int t = tof(step);
    int zo = 0;
    asm volatile("" : "+s"(zo));
    f32x16 acc[2][4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(&bs[g * 256 + ubase + 8 * j]);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[0][g][4 * j + r] = acc[1][g][4 * j + r] = b[r];
      }
    v8i b8[2];      // F8: e4m3 h of the current k block (64 values of k), both tiles
    int scv = 0;    //     its four fragment exponents
#pragma unroll
    for (int ks = 0; ks < FKS; ++ks) {
      const int s = ks & 1;
      bf16x8 bh[2], bl[2];
      if (F8 && ks >= 8 && ((ks - 8) & 3) == 0) {
        const int kb = (ks - 8) >> 2;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const i32x4 p0 = *reinterpret_cast<const i32x4*>(h8row[e] + 64 * kb), p1 = *reinterpret_cast<const i32x4*>(h8row[e] + 64 * kb + 32);
          b8[e] = __builtin_shufflevector(p0, p1, 0, 1, 2, 3, 4, 5, 6, 7);
        }
        scv = ssc[kb];
      }
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        if (ks < 8) {
          const f32x4 v0 = xw[e][(4 * ks + 2 * half) * 32 + l31], v1 = xw[e][(4 * ks + 2 * half + 1) * 32 + l31];
          bf16x4 h0, l0, h1, l1;
          unpack_hl4(v0, h0, l0);
          unpack_hl4(v1, h1, l1);
          bh[e] = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
          bl[e] = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
        } else {
          // (ONE base address per tile, the k-step in the instruction's immediate offset: written as an index expression
          //  8 * half + 16 * (ks - 8), the compiler merges the two disjoint-bit terms with v_or and then keeps sixteen
          //  separate address registers alive across the step -- the 64 B / lane of scratch of round 3)
          bh[e] = *reinterpret_cast<const bf16x8*>(hrow[e] + 16 * (ks - 8));
          if (!H16) bl[e] = *reinterpret_cast<const bf16x8*>(hrow[e] + 2 * SQ * HROW + 16 * (ks - 8));
        }
      }
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        if (F8 && ks >= 8) {    // fp16 hi of 256 W_hh x fp16 h; the FP8 term of this k-step's gate follows both tiles' hi terms
          const f16x8 b16 = __builtin_bit_cast(f16x8, bh[e]);
#pragma unroll
          for (int g = 0; g < 4; ++g) acc[e][g] = mfma16h(__builtin_bit_cast(f16x8, wr[s][g]), b16, acc[e][g]);
          continue;
        }
        if (H16 && ks >= 8) {   // fp16 W_hh (hi, lo of 256 w) x fp16 h: two terms
          const f16x8 b16 = __builtin_bit_cast(f16x8, bh[e]);
#pragma unroll
          for (int g = 0; g < 4; ++g) acc[e][g] = mfma16h(__builtin_bit_cast(f16x8, wr[s][2 * g]), b16, acc[e][g]);
#pragma unroll
          for (int g = 0; g < 4; ++g) acc[e][g] = mfma16h(__builtin_bit_cast(f16x8, wr[s][2 * g + 1]), b16, acc[e][g]);
          continue;
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[e][g] = mfma32(wr[s][2 * g], bh[e], acc[e][g]);
#pragma unroll
        for (int g = 0; g < 4; ++g)
          if (!W1) acc[e][g] = mfma32(wr[s][2 * g + 1], bh[e], acc[e][g]);
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[e][g] = mfma32(wr[s][2 * g], bl[e], acc[e][g]);
      }
      if constexpr (F8) {
        if (ks >= 8) {
          constexpr int ONE = 127;     // h is stored unscaled
          const int q = ks - 8, g8 = q & 3;
          const v8i a8 = __builtin_shufflevector(__builtin_bit_cast(i32x4, wr[s][4]), __builtin_bit_cast(i32x4, wr[s][5]), 0, 1, 2, 3,
                                                 4, 5, 6, 7);
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            if (g8 == 0) acc[e][0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8[e], acc[e][0], 0, 0, 0, scv, 0, ONE);
            if (g8 == 1) acc[e][1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8[e], acc[e][1], 0, 0, 1, scv, 0, ONE);
            if (g8 == 2) acc[e][2] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8[e], acc[e][2], 0, 0, 2, scv, 0, ONE);
            if (g8 == 3) acc[e][3] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8[e], acc[e][3], 0, 0, 3, scv, 0, ONE);
          }
        }
      }
      refill(s, (ks + 2) % FKS, zo);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();  // every wave has read h_{t-1} and x_t: both may be overwritten now
    dma_x(0, tof(step + 1));
    dma_x(1, tof(step + 1));
    __builtin_amdgcn_sched_barrier(0);

#pragma unroll
    for (int e = 0; e < 2; ++e) {
      __bf16* nhi = &hl[0][(32 * e + l31) * HROW + ubase];
      __bf16* nlo = &hl[1][(32 * e + l31) * HROW + ubase];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4 vi, vf, vg, vo, vc, vh;
        const f32x4 cold = e == 0 ? c0[32 * j] : c1[j];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float ig = H16 ? c2_sig256(acc[e][0][4 * j + r]) : fsig(acc[e][0][4 * j + r]);
          const float fg = H16 ? c2_sig256(acc[e][1][4 * j + r]) : fsig(acc[e][1][4 * j + r]);
          const float gg = H16 ? c2_tanh256(acc[e][2][4 * j + r]) : ftanh(acc[e][2][4 * j + r]);
          const float og = H16 ? c2_sig256(acc[e][3][4 * j + r]) : fsig(acc[e][3][4 * j + r]);
          const float cn = fg * cold[r] + ig * gg;
          vi[r] = ig;
          vf[r] = fg;
          vg[r] = gg;
          vo[r] = og;
          vc[r] = cn;
          vh[r] = og * ftanh(cn);
        }
        bf16x4 h_hi, h_lo;
        split4(vh, h_hi, h_lo);
        if constexpr (H16) {   // the recurrent operand: fp16(h), one plane
          *reinterpret_cast<u32x2*>(nhi + 8 * j) = enc_f16x4(vh);
          if constexpr (F8) {  // + its e4m3 image, the operand of the lo term
            int c8 = 0;
            c8 = __builtin_amdgcn_cvt_pk_fp8_f32(vh[0], vh[1], c8, false);
            c8 = __builtin_amdgcn_cvt_pk_fp8_f32(vh[2], vh[3], c8, true);
            *reinterpret_cast<int*>(h8 + (32 * e + l31) * H8ROW + ubase + 8 * j) = c8;
          }
        } else {
          *reinterpret_cast<bf16x4*>(nhi + 8 * j) = h_hi;
          *reinterpret_cast<bf16x4*>(nlo + 8 * j) = h_lo;
        }
        if (e == 0)
          c0[32 * j] = vc;
        else
          c1[j] = vc;
        st_gate(vi, e, t, 0, j);
        st_gate(vf, e, t, 1, j);
        st_gate(vg, e, t, 2, j);
        st_gate(vo, e, t, 3, j);
        bst(vc, crs(p.cbuf, e, t), clane, 2 * j * 512);
        bst(pack_hl4(h_hi, h_lo), crs(p.hcat, e, t), clane, 2 * j * 512);  // BLS
        __builtin_amdgcn_sched_barrier(0);  // one run at a time: bounds the live temporaries
      }
    }
    // x_{t+1} was requested before the cell update; every wave waits for its own share before the barrier publishes it.  vmcnt
    // counts in order and exactly 2 tiles x 4 runs x (4 gates + c + h) = 48 stores were issued behind the four DMA loads, so
    // vmcnt(48) is "the DMA has landed" -- rounds 3-5 waited for vmcnt(0): every step then also sat out the acknowledgement of
    // its 384 KB of stores (the HBM phase of the step, ~14 us chip-wide) before the next step's MFMAs could start, where now they
    // drain under them.  (ws_lstm_fused_args.hfmt bit 1, WESEP_FUSED_DRAIN=1: the old wait, for A/B.)
    if (p.hfmt & 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
    __syncthreads();
  }
}

__global__ __launch_bounds__(512, 1) void lstm_fwd_fused64_kernel(const ws_lstm_fused_args p) {
  lstm_fwd_fused64_body<0>(p);
}
__global__ __launch_bounds__(512, 1) void lstm_fwd_fused64h_kernel(const ws_lstm_fused_args p) {
  lstm_fwd_fused64_body<WS_GATES_H2>(p);
}
__global__ __launch_bounds__(512, 1) void lstm_fwd_fused64h_w1_kernel(const ws_lstm_fused_args p) {   // measurement only
  lstm_fwd_fused64_body<WS_GATES_H2, true>(p);
}
__global__ __launch_bounds__(512, 1) void lstm_fwd_fused64h16_kernel(const ws_lstm_fused_args p) {   // hfmt 1: fp16 h, two terms
  lstm_fwd_fused64_body<WS_GATES_H2, false, true>(p);
}
__global__ __launch_bounds__(512, 1) void lstm_fwd_fused64h8_kernel(const ws_lstm_fused_args p) {   // hfmt 5: + the lo term in FP8
  lstm_fwd_fused64_body<WS_GATES_H2, false, true, true>(p);
}

extern "C" int ws_lstm_fwd_fused(const ws_lstm_fused_args* a, void* stream) {
  WS_REQUIRE(a && a->gates && a->cbuf && a->hcat && a->xn && a->wpack && a->bias, "ws_lstm_fwd_fused: null pointer");
  WS_REQUIRE(a->nseq > 0 && a->L > 0, "ws_lstm_fwd_fused: bad nseq/L");
  WS_REQUIRE(a->gfmt >= WS_GATES_F32 && a->gfmt <= WS_GATES_H2F, "ws_lstm_fwd_fused: gfmt %d", a->gfmt);
  WS_REQUIRE((a->hfmt & ~7) == 0 && (!(a->hfmt & 1) || a->gfmt != WS_GATES_F32) && (!(a->hfmt & 4) || (a->hfmt & 1)),
             "ws_lstm_fwd_fused: hfmt %d (1 = fp16 h, pack from ws_lstm_pack_fused_h16; 5 = + FP8 lo term, pack from "
             "ws_lstm_pack_fused_h8; 2-byte gate formats only)", a->hfmt);
  const int ntile = (a->nseq + SQ - 1) / SQ;
  dim3 grid(ntile, 2), block(512);
  hipStream_t s = (hipStream_t)stream;
  // 64 sequences per workgroup when that saves time after rounding both grids up to whole rounds of the chip (one
  // workgroup per CU either way; a 64-sequence step costs 1.8x a 32-sequence step).  Measured: pBSRNN band view,
  // 1002 vs 502 workgroups (4 vs 2 rounds): 2.70 -> 2.43 ms; TF-GridNet intra path, 752 vs 376 (3 vs 2 rounds): 4.08 -> 4.64.
  // WS_FUSED_SEQS=32|64 overrides (diagnostics; both kernels give the same bits)
  const char* env = getenv("WS_FUSED_SEQS");
  static int cus = 0;      // same part on every device of a node; queried once
  if (!cus && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0) != hipSuccess) cus = 256;
  const int rounds32 = (2 * ntile + cus - 1) / cus, rounds64 = (2 * ((ntile + 1) / 2) + cus - 1) / cus;
  const bool wide = env ? atoi(env) == 64 : 9 * rounds64 < 5 * rounds32;
  ws_prof_begin(WS_PROF_LSTM_FWD, s);
  const char* w1 = getenv("WS_FUSED_W1");   // measurement only: one weight plane (bf16 weights), see lstm_fwd_fused64_body
  if ((a->hfmt & 4) && wide)
    hipLaunchKernelGGL(lstm_fwd_fused64h8_kernel, dim3((ntile + 1) / 2, 2), block, 0, s, *a);
  else if (a->hfmt & 4)
    hipLaunchKernelGGL((lstm_fwd_fused_kernel<WS_GATES_H2, true, true>), grid, block, 0, s, *a);
  else if ((a->hfmt & 1) && wide)
    hipLaunchKernelGGL(lstm_fwd_fused64h16_kernel, dim3((ntile + 1) / 2, 2), block, 0, s, *a);
  else if (a->hfmt & 1)
    hipLaunchKernelGGL((lstm_fwd_fused_kernel<WS_GATES_H2, true>), grid, block, 0, s, *a);
  else if (wide && a->gfmt && w1 && atoi(w1) == 1)
    hipLaunchKernelGGL(lstm_fwd_fused64h_w1_kernel, dim3((ntile + 1) / 2, 2), block, 0, s, *a);
  else if (wide && a->gfmt)
    hipLaunchKernelGGL(lstm_fwd_fused64h_kernel, dim3((ntile + 1) / 2, 2), block, 0, s, *a);
  else if (wide)
    hipLaunchKernelGGL(lstm_fwd_fused64_kernel, dim3((ntile + 1) / 2, 2), block, 0, s, *a);
  else if (a->gfmt)
    hipLaunchKernelGGL(lstm_fwd_fused_kernel<WS_GATES_H2>, grid, block, 0, s, *a);
  else
    hipLaunchKernelGGL(lstm_fwd_fused_kernel<0>, grid, block, 0, s, *a);
  ws_prof_end(WS_PROF_LSTM_FWD, s);
  return ws_check_launch("ws_lstm_fwd_fused");
}
