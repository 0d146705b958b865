// Time-view BPTT of the split-bf16 BLSTM over PAIRS of co-operating workgroups (blocked layout BL).
//
// What bounds the streaming BPTT kernels (lstm_bf16*.hip) in the time view of pBSRNN (1024 sequences x 501
// latency-bound steps) is the per-step stream of BOTH W_hh planes from L2 through one CU's L1: 1 MB per workgroup and
// step.  The cluster BPTT (lstm_cluster.hip) removes the stream but needs all 256 CUs and a 56 KB + 56 KB reduce-scatter
// per step.  This kernel sits between the two: a (32-sequence tile, direction) is owned by TWO workgroups on two CUs
// (128 CUs at R = 32: the other half of the chip stays free for the side stream's weight-gradient GEMMs), split by
// GATE ROWS:
//   workgroup hs owns the hidden units U = [128 hs, 128 hs + 128): their cell backward, their 512 d(gates) columns
//   (the kernel's output) and the matching 512 rows of W_hh, whose hi (bf16) plane lives in the REGISTERS of its 8
//   waves for the whole launch (wave w: the 32 output units of m-tile w x all 512 local k = 128 VGPRs); only the lo
//   plane is streamed (256 KB per step: a quarter of the streaming kernels' traffic through L1).
//   dh_{t-1}[seq][u'] = sum_k dgates_t[seq][k] W_hh[k][u']  splits over k into the two workgroups' partial sums:
//   each multiplies its own d(gates) (B operand: a bf16 hi/lo image in LDS) into a partial for ALL 256 units and hands
//   the partner the half that belongs to the partner's units -- 16 KB fp32 out and 16 KB in per step (the cluster
//   kernel: 56 + 56; a split by hidden slice would gather 64 KB of d(gates) instead).
// The hand-off is hidden behind the other half of the MFMAs by wave roles:
//   X-waves (0..3): the m-tiles of the PARTNER's units at s_setprio 1 (the SIMD's matrix pipe serves them first),
//                   then publish (write-through sc1 16-byte stores, recipe R1 of cdna_hip_programming.md Guideline 16:
//                   every storing wave drains vmcnt, one relaxed agent-scope flag per wave), poll the partner's flag for
//                   the same m-tile and gather its partial (sc1 loads) into LDS;
//   O-waves (4..7): the m-tiles of the workgroup's OWN units, running under and behind the X-waves' MFMAs, into LDS.
// The exchange is per WAVE (four flags per workgroup and step): no workgroup barrier sits between a wave's last MFMA
// and its publish.  Exchange slots are double-buffered by step parity (a workgroup can publish step s + 2 only after it
// gathered the partner's step s + 1, which the partner published after it had consumed step s).  All spins are bounded;
// a timeout poisons d(gates) with NaN and sets the launch's timeout word and *status (the kernel works in place: no
// device-side repair, callers check the word -- dev.poll_cluster_status).  Sums are taken in a fixed order: results are
// bit-identical run to run.
//
// Round 5 (RF template parameter, ws_lstm_pair_args.rfmt): 1 = the product on the fp16 MFMA instruction against the STORED
// scaled-fp16 d(gates) (two terms, one LDS image plane, data-tagged hand-off); 2 (the default of the Python layer) = the same
// with W_hh's lo plane as block-scaled FP8 -- the WHOLE of W_hh then stays on the compute unit (hi plane in registers, 27 of 32
// lo k-steps in LDS, 5 in registers) and the step loop loads nothing of it.  The paragraphs above describe RF = 0.
//
// Residency: both members of a pair must be resident at the same time: the launcher requires 2 workgroups per
// (tile, direction) <= CUs (one workgroup per CU: 145 - 157 KB of LDS, 8 waves x <= 256 VGPRs).  Members of a pair are 8
// apart in dispatch order, i.e. on the same XCD (blockIdx % 8) and behind the same L2 -- for speed; the protocol does
// not depend on it.
#include "lstm_bf16_common.h"

typedef __attribute__((address_space(1))) unsigned gu32;
#define SC1 16          // aux bit of raw buffer ops: sc1 (write-through store / L1-bypassing load)
#define PR_ROW 520      // bf16 per LDS row of the local d(gates) image (512 + 8: 1040 B = 4 banks mod 64)
#define PR_SPIN_LIMIT (1u << 22)
#define PR_XSLOT 16384  // bytes of one exchange slot: 1024 cells (32 unit quads x 32 sequences) x 16 B

// ---------------------------------------------------------------------------------------------
// weight packing: 16-byte units, `lane` = the MFMA lane that loads the unit
//   unit (((d*2 + hs)*8 + w)*2 + part)*32*64 + ks*64 + lane, element j
//     = part( W_hh[d][ g*256 + 128 hs + ul ][ 32 mt(w, hs) + (lane & 31) ] ),  local k = 16 ks + 8 (lane >> 5) + j
//       = g*128 + ul;  mt = m-tile of wave w: partner's units for w < 4 (4 (1 - hs) + w), own units else (4 hs + w - 4)
// ---------------------------------------------------------------------------------------------
__global__ void lstm_pack_pair_kernel(const float* __restrict__ whh_f, const float* __restrict__ whh_r,
                                      __bf16* __restrict__ out) {
  const int total = 2 * LG * LH;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    int r = idx;
    const int j = r & 7; r >>= 3;
    const int lane = r & 63; r >>= 6;
    const int ks = r & 31; r >>= 5;
    const int w = r & 7; r >>= 3;
    const int hs = r & 1; r >>= 1;
    const int d = r;
    const float* W = d ? whh_r : whh_f;
    const int mt = w < 4 ? 4 * (1 - hs) + w : 4 * hs + (w - 4);
    const int u = 32 * mt + (lane & 31);
    const int kl = 16 * ks + 8 * (lane >> 5) + j;
    const int row = (kl >> 7) * LH + 128 * hs + (kl & 127);
    const float v = W[row * LH + u];
    const __bf16 hi = (__bf16)v;
    const long long unit = ((long long)((d * 2 + hs) * 8 + w) * 2) * (32 * 64) + ks * 64 + lane;
    out[unit * 8 + j] = hi;
    out[(unit + 32 * 64) * 8 + j] = (__bf16)(v - (float)hi);
  }
}

// The same unit order with fp16 elements: hi = fp16(256 w), lo = fp16(256 w - hi) -- the A operand of
// v_mfma_f32_32x32x16_f16 in the RF = 1 kernel below (the factor 2^8 keeps the lo terms of |w| >= 5e-4 out of the fp16
// denormals, as in ws_pack_w_f16; the kernel undoes it exactly).  |w| < 255.
__global__ void lstm_pack_pair_f16_kernel(const float* __restrict__ whh_f, const float* __restrict__ whh_r,
                                          _Float16* __restrict__ out) {
  const int total = 2 * LG * LH;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    int r = idx;
    const int j = r & 7; r >>= 3;
    const int lane = r & 63; r >>= 6;
    const int ks = r & 31; r >>= 5;
    const int w = r & 7; r >>= 3;
    const int hs = r & 1; r >>= 1;
    const int d = r;
    const float* W = d ? whh_r : whh_f;
    const int mt = w < 4 ? 4 * (1 - hs) + w : 4 * hs + (w - 4);
    const int u = 32 * mt + (lane & 31);
    const int kl = 16 * ks + 8 * (lane >> 5) + j;
    const int row = (kl >> 7) * LH + 128 * hs + (kl & 127);
    const float v = 256.f * W[row * LH + u];
    const _Float16 hi = (_Float16)v;
    const long long unit = ((long long)((d * 2 + hs) * 8 + w) * 2) * (32 * 64) + ks * 64 + lane;
    out[unit * 8 + j] = hi;
    out[(unit + 32 * 64) * 8 + j] = (_Float16)(v - (float)hi);
  }
}

// RF = 2 (ABI v18): the lo plane as FP8 (OCP e4m3) codes of (256 w - hi) / S, S a power of two per (d, hs, w) block -- the
// 32 k-steps of one wave -- chosen from the block's max |256 w| so that the largest possible lo (half an fp16 ulp of the
// largest hi) maps to 256 (e4m3 overflows to NaN above 448, it does not saturate: profiles/r05_fp8_probe.txt).  Block of
// 64 KB as before: fp16 hi plane at 0 (32 KB, the RF = 1 order), codes at 32 KB (unit = 8 bytes per lane and k-step, 16 KB),
// S as one float at 48 KB.  hi + lo carries ~16 significant bits of every weight (the bf16 pair of round 4: 16); the
// 48 KB per wave that remain ALL stay on the CU for the whole launch (hi in registers, lo in LDS + 16 registers): nothing of
// W_hh is streamed any more.  One workgroup per block (it needs the block's max first).
__global__ __launch_bounds__(512) void lstm_pack_pair_f8_kernel(const float* __restrict__ whh_f, const float* __restrict__ whh_r,
                                                                 char* __restrict__ out) {
  __shared__ float red[8];
  const int blk = blockIdx.x, w = blk & 7, hs = (blk >> 3) & 1, d = blk >> 4;
  const float* W = d ? whh_r : whh_f;
  const int mt = w < 4 ? 4 * (1 - hs) + w : 4 * hs + (w - 4);
  const int tid = threadIdx.x;
  float v0[16], v1[16], m = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int pi = tid + 512 * i, j2 = pi & 3, lane = (pi >> 2) & 63, ks = pi >> 8;
    const int u = 32 * mt + (lane & 31), kl = 16 * ks + 8 * (lane >> 5) + 2 * j2;
    const int row = (kl >> 7) * LH + 128 * hs + (kl & 127);   // (kl even: kl + 1 stays inside the same gate's 128 rows)
    v0[i] = 256.f * W[row * LH + u];
    v1[i] = 256.f * W[(row + 1) * LH + u];
    m = fmaxf(m, fmaxf(fabsf(v0[i]), fabsf(v1[i])));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  m = red[0];
#pragma unroll
  for (int i = 1; i < 8; ++i) m = fmaxf(m, red[i]);
  // m in [2^(e-1), 2^e): |lo| <= 2^(e-12); S = 2^(e-20) puts that at 256.  Biased exponent of S = that of m - 19.
  const int eb = (int)((__builtin_bit_cast(unsigned, m) >> 23) & 0xffu);
  const float S = (eb > 19 && eb < 255) ? __builtin_bit_cast(float, (unsigned)(eb - 19) << 23) : 1.f;
  char* ob = out + (long long)blk * (64 * 1024);
  if (tid == 0) *reinterpret_cast<float*>(ob + 48 * 1024) = S;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int pi = tid + 512 * i, j2 = pi & 3, lane = (pi >> 2) & 63, ks = pi >> 8;
    const _Float16 h0 = (_Float16)v0[i], h1 = (_Float16)v1[i];
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    typedef short s16x2 __attribute__((ext_vector_type(2)));
    *reinterpret_cast<f16x2*>(ob + ks * 1024 + lane * 16 + j2 * 4) = f16x2{h0, h1};
    const s16x2 c = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(s16x2{0, 0}, v0[i] - (float)h0, v1[i] - (float)h1, S, false);
    *reinterpret_cast<short*>(ob + 32 * 1024 + ks * 512 + lane * 8 + j2 * 2) = c[0];
  }
}

// RF = 3 (ABI v20): the codes of RF = 2 -- same values, same block exponent S -- laid out as A operands of
// v_mfma_scale_f32_32x32x64_f8f6f4: eight fragments kb of K = 64 per wave, a lane's 32 bytes as two 16-byte pieces at
// 32 KB + 2 KB kb + 1 KB piece + lane * 16.  Which of the 64 columns a byte holds is free as long as both operands agree (the
// instruction only pairs byte b of lane (row, half) of A with byte b of lane (column, half) of B,
// profiles/r06_c20_f8_probe.txt), and the B operand is built IN REGISTERS from the four fp16 fragments of the chunk's k-steps:
// bytes 8 i .. 8 i + 7 of a lane = the eight columns 16 (4 kb + i) + 8 (lane >> 5) + j of k-step 4 kb + i -- a lane's 32 bytes
// are its four 8-byte units of the RF = 2 pack, in order.  S at 48 KB as before; its E8M0 byte (the scale operand of the
// instruction) as an int behind it.
__global__ __launch_bounds__(512) void lstm_pack_pair_f8mx_kernel(const float* __restrict__ whh_f, const float* __restrict__ whh_r,
                                                                   char* __restrict__ out) {
  __shared__ float red[8];
  const int blk = blockIdx.x, w = blk & 7, hs = (blk >> 3) & 1, d = blk >> 4;
  const float* W = d ? whh_r : whh_f;
  const int mt = w < 4 ? 4 * (1 - hs) + w : 4 * hs + (w - 4);
  const int tid = threadIdx.x;
  float v0[16], v1[16], m = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int pi = tid + 512 * i, j2 = pi & 3, lane = (pi >> 2) & 63, ks = pi >> 8;
    const int u = 32 * mt + (lane & 31), kl = 16 * ks + 8 * (lane >> 5) + 2 * j2;
    const int row = (kl >> 7) * LH + 128 * hs + (kl & 127);
    v0[i] = 256.f * W[row * LH + u];
    v1[i] = 256.f * W[(row + 1) * LH + u];
    m = fmaxf(m, fmaxf(fabsf(v0[i]), fabsf(v1[i])));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  m = red[0];
#pragma unroll
  for (int i = 1; i < 8; ++i) m = fmaxf(m, red[i]);
  const int eb = (int)((__builtin_bit_cast(unsigned, m) >> 23) & 0xffu);
  const bool ok = eb > 19 && eb < 255;
  const float S = ok ? __builtin_bit_cast(float, (unsigned)(eb - 19) << 23) : 1.f;
  char* ob = out + (long long)blk * (64 * 1024);
  if (tid == 0) {
    *reinterpret_cast<float*>(ob + 48 * 1024) = S;
    *reinterpret_cast<int*>(ob + 48 * 1024 + 4) = ok ? eb - 19 : 127;
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int pi = tid + 512 * i, j2 = pi & 3, lane = (pi >> 2) & 63, ks = pi >> 8;
    const _Float16 h0 = (_Float16)v0[i], h1 = (_Float16)v1[i];
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    typedef short s16x2 __attribute__((ext_vector_type(2)));
    *reinterpret_cast<f16x2*>(ob + ks * 1024 + lane * 16 + j2 * 4) = f16x2{h0, h1};
    const s16x2 c = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(s16x2{0, 0}, v0[i] - (float)h0, v1[i] - (float)h1, S, false);
    *reinterpret_cast<short*>(ob + 32 * 1024 + (ks >> 2) * 2048 + ((ks >> 1) & 1) * 1024 + lane * 16 + (ks & 1) * 8 + 2 * j2) = c[0];
  }
}

extern "C" int ws_lstm_pack_pair_f8mx(const float* whh_f, const float* whh_r, float* pack, void* stream) {
  WS_REQUIRE(whh_f && whh_r && pack, "ws_lstm_pack_pair_f8mx: null pointer");
  hipLaunchKernelGGL(lstm_pack_pair_f8mx_kernel, dim3(32), dim3(512), 0, (hipStream_t)stream, whh_f, whh_r,
                     reinterpret_cast<char*>(pack));
  return ws_check_launch("ws_lstm_pack_pair_f8mx");
}

extern "C" int ws_lstm_pack_pair_f8(const float* whh_f, const float* whh_r, float* pack, void* stream) {
  WS_REQUIRE(whh_f && whh_r && pack, "ws_lstm_pack_pair_f8: null pointer");
  hipLaunchKernelGGL(lstm_pack_pair_f8_kernel, dim3(32), dim3(512), 0, (hipStream_t)stream, whh_f, whh_r,
                     reinterpret_cast<char*>(pack));
  return ws_check_launch("ws_lstm_pack_pair_f8");
}

extern "C" int ws_lstm_pack_pair_f16(const float* whh_f, const float* whh_r, float* pack, void* stream) {
  WS_REQUIRE(whh_f && whh_r && pack, "ws_lstm_pack_pair_f16: null pointer");
  hipLaunchKernelGGL(lstm_pack_pair_f16_kernel, dim3(512), dim3(256), 0, (hipStream_t)stream, whh_f, whh_r,
                     reinterpret_cast<_Float16*>(pack));
  return ws_check_launch("ws_lstm_pack_pair_f16");
}

extern "C" int ws_lstm_pack_pair(const float* whh_f, const float* whh_r, float* pack, void* stream) {
  WS_REQUIRE(whh_f && whh_r && pack, "ws_lstm_pack_pair: null pointer");
  hipLaunchKernelGGL(lstm_pack_pair_kernel, dim3(512), dim3(256), 0, (hipStream_t)stream, whh_f, whh_r,
                     reinterpret_cast<__bf16*>(pack));
  return ws_check_launch("ws_lstm_pack_pair");
}

// cold path of a bounded wait: set this launch's timeout word and the caller's sticky status word
__device__ __noinline__ void pair_timed_out(unsigned* tword, unsigned* status) {
  __hip_atomic_store((gu32*)tword, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (status) __hip_atomic_store((gu32*)status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

#define PAIR_RING 4   // lo-plane fragments per ring slot (two slots)
// k-steps whose lo fragments stay in LDS (the rest is streamed): 8 beside the two-plane bf16 image of d(gates), 12 beside the
// one-plane fp16 image of RF = 1 (33 instead of 65 KB)
template <int RF> struct pair_ldsk { static constexpr int value = RF ? 12 : 8; };

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// one product term of the recurrent GEMM: RF = 0 bf16 x bf16, RF = 1 fp16 x fp16 (operands travel as bf16x8 bit patterns)
template <int RF>
__device__ __forceinline__ f32x16 pair_mfma(const bf16x8& a, const bf16x8& b, const f32x16& c) {
  if constexpr (RF != 0)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return mfma32(a, b, c);
}

// V: compile-time variant bits (the step body stays free of run-time branches): 8 = test build that forces a timeout
// in pair 0 at step 2 (the bisect builds of round 3 -- no reloads, no priorities, cell backward only, no hand-off, no
// MFMA loop -- are gone from the library; profiles/r03_store_hazard.md records what they found)
// variant 2048 (diagnosis): s_memtime stamps of pair 0 / member 0, waves 0 (X) and 4 (O), into dbg_buf:
//   dbg_buf[((step * 2 + role) * 8 + k)] as 64-bit ticks; k: 0 loop top, 1 cell backward done, 2 past S1, 3 MFMA loop done,
//   4 X: published + drained + flagged / O: partial in LDS, 5 X: next step's loads requested, 6 X: partner's flag seen,
//   7 X: gather arrived; the next k = 0 closes the step (past S2)
#define TS(k)                                                                                                      \
  if constexpr (V & 2048) {                                                                                        \
    if (pr == 0 && hs == 0 && wx == 0 && lane == 0 && p.dbg_buf)                                                   \
      reinterpret_cast<unsigned long long*>(p.dbg_buf)[(step * 2 + role) * 8 + (k)] = __builtin_readcyclecounter(); \
  }

// GF: WS_GATES_* (lstm_bf16_common.h): H2 = unorm16 gates in, bf16 d(gates) out, in place on the BLH buffer; H2S = unorm16
// gates in, d(gates) as BLS pairs to p.dgates
// RF (ABI v17, ws_lstm_pair_args.rfmt; GF = WS_GATES_H2F only): 1 = the recurrent product on v_mfma_f32_32x32x16_f16 with the
// STORED scaled-fp16 d(gates) as its one B operand -- exactly the value the weight-gradient GEMMs and d(xn) consume -- and
// W_hh as fp16 hi / lo of 256 w (ws_lstm_pack_pair_f16): TWO MFMAs per product instead of three, one LDS image plane
// instead of two, and the LDS that frees holds four more k-steps of the lo plane (160 instead of 192 KB per step from L2).
// Settled on the CPU emulation first (tools/r04_h2_numerics.py, probe bit 4096: no measurable change of any gradient at
// the fixture size or at 501 frames).
// RF = 2 (ABI v18, ws_lstm_pack_pair_f8): RF = 1 with the lo plane as FP8 codes and a block scale -- k-steps 0..26 of the lo
// plane in 108 KB of LDS, 27..31 in 10 registers (the ring's 32 are gone), converted to fp16 fragments on the way
// into the MFMA (v_cvt_scalef32_pk_f16_fp8 at scale 1: codes are exact in fp16; the block scale multiplies the lo
// accumulator once per step).  No load of W_hh inside the step loop: what the side stream's GEMMs do to L2 no longer
// reaches the MFMA phase.  Emulation first (probe 32768: no gradient moves, fixture size and 501 frames).
template <int V, int GF = 0, int RF = 0>
__global__ __launch_bounds__(512, 2) void lstm_bwd_pair_kernel(const ws_lstm_pair_args p) {
  static_assert(RF == 0 || GF == WS_GATES_H2F, "the fp16 recurrence takes the scaled-fp16 d(gates) of WS_GATES_H2F");
  constexpr int PAIR_LDSK = pair_ldsk<RF>::value;
  __shared__ __attribute__((aligned(16))) __bf16 bimg[RF ? 1 : 2][SQ * PR_ROW];  // [part][seq][local gate col] 65 / 33 KB
  constexpr int LDSK8 = 27;   // RF = 2: k-steps of the 8-byte lo fragments in LDS (108 KB: what 160 KB leave); 5 in registers
  constexpr int LDSB = 6;     // RF = 3: FP8 fragments (K = 64, 2 KB per wave) in LDS (96 KB); 2 in registers (16)
  constexpr bool RES = RF >= 2;   // all of W_hh resident on the CU: no ring, the next step's state touched early
  __shared__ __attribute__((aligned(16))) bf16x8 whl[RF == 3 ? 8 * LDSB * 128 : RF == 2 ? 4 * LDSK8 * 64 : 8 * PAIR_LDSK * 64];   // lo fragments, 64 / 96 / 108 / 80 KB
  __shared__ __attribute__((aligned(16))) f32x4 rec[2][512];                // the other role's partial dh, 16 KB
  const int ntile = (p.nseq + SQ - 1) / SQ, npair = 2 * ntile;
  // block -> (pair, member): members of a pair are 8 blocks apart (same XCD under round-robin dispatch)
  const int pr = ((int)blockIdx.x >> 4) * 8 + ((int)blockIdx.x & 7), hs = ((int)blockIdx.x >> 3) & 1;
  if (pr >= npair) return;
  // variant 2048, round 6: every workgroup's wall-clock times (100 MHz, one counter for the chip) of entry, first loop top and loop
  // end, behind the step stamps: dbg_buf as u64 [L * 16 + (pr * 2 + hs) * 4 + {0, 1, 2}] (tools/r06_instep_stamps.py)
  long long wall_in = 0;
  if constexpr (V & 2048) wall_in = wall_clock64();
  const int d = pr & 1, tile = pr >> 1;
  const int tid = threadIdx.x, lane = tid & 63, n = lane & 31, half = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool xrole = w < 4;
  const int wx = w & 3, role = w >> 2;
  const int L = p.L;

  // ---- cells.  The D fragment of m-tile wx holds, per lane (n = sequence slot, half), four cells of 4 units:
  //      local unit quad 8 wx + 2 q4 + half, q4 = 0..3.  The cell backward of quads q4 = 0, 1 runs on the X-wave wx
  //      (which RECEIVES the partner's partial of exactly these cells into registers), that of q4 = 2, 3 on the O-wave
  //      4 + wx (which COMPUTES the own partial of exactly these cells): only the other half of each sum crosses LDS.
  //      BL cell of (gate g, local quad q): ((d*256 + g*64 + 32 hs + q)*32 + slot)*16 bytes inside the block.
  const int q0 = 8 * wx + 4 * role + half;                   // this thread's cells: quads q0 (e = 0) and q0 + 2 (e = 1)
  const int gvo = ((d * 256 + 32 * hs + q0) * 32 + n) * 16;  // bytes; + g*64*512; cell e: + 2e*512
  const int cvo = ((d * 64 + 32 * hs + q0) * 32 + n) * 16;   // bytes; cell e: + 2e*512
  float* gdst = GF == WS_GATES_H2S ? p.dgates : p.gates;
  auto grs = [&](int t) { return mkrsrc(gdst + (long long)(tile * L + t) * (SQ * 2 * LG), SQ * 2 * LG * 4); };
  auto hrs = [&](int t) { return mkrsrc(p.gates + (long long)(tile * L + t) * (SQ * LG), SQ * 2 * LG * 2); };  // BLH
  constexpr bool G2 = GF == WS_GATES_H2 || GF == WS_GATES_H2F;  // 2-byte d(gates): bf16, or fp16 scaled by dS
  float* hdst = (G2 && p.dgates) ? p.dgates : p.gates;  // in place, or to their own BLH buffer
  auto ors = [&](int t) { return mkrsrc(hdst + (long long)(tile * L + t) * (SQ * LG), SQ * 2 * LG * 2); };
  const float dS = GF == WS_GATES_H2F ? ws_dgates_scale(*p.amax) : 1.f;
  auto crs = [&](const float* b, int t) { return mkrsrc(b + (long long)(tile * L + t) * (SQ * 2 * LH), SQ * 2 * LH * 4); };
  typedef typename gate_cell<GF>::type gcell;
  auto ld_gate = [&](int t, int g, int e) -> gcell {
    if constexpr (GF != 0) return bld8(hrs(t), gvo >> 1, (g * 64 + 2 * e) * 256);
    else return bld(grs(t), gvo, (g * 64 + 2 * e) * 512);
  };
  // ---- exchange: X[pair][parity][destination member][cell] x 16 B; flags[pair][source member][wave]
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<char*>(p.xchg) + (long long)pr * (4 * PR_XSLOT), 0, 4 * PR_XSLOT, 0x00020000);
  gu32* flags = (gu32*)(p.flags) + pr * 8;
  const int xc0 = ((8 * wx + half) * 32 + n) * 16;  // exchange byte offset of this lane's cell q4 = 0; q4: + q4*2*512

  // ---- resident hi plane of this wave's m-tile: 32 k-steps x 16 B per lane; lo plane: k-steps 0..7 in LDS, the
  //      rest streamed every step through a two-slot register ring (no load is in flight across the cell backward)
  const char* wbase = reinterpret_cast<const char*>(p.wpack) + (long long)((d * 2 + hs) * 8 + w) * (2 * 32 * 1024);
  bf16x8 wh[32];
#pragma unroll
  for (int ks = 0; ks < 32; ++ks) wh[ks] = *reinterpret_cast<const bf16x8*>(wbase + ks * 1024 + lane * 16);
  bf16x8* wlds = &whl[w * PAIR_LDSK * 64 + lane];
  u32x2* wlds8 = reinterpret_cast<u32x2*>(whl) + w * LDSK8 * 64 + lane;   // RF = 2: 8-byte units
  u32x2 wq[RF == 2 ? 32 - LDSK8 : 1];
  v8i wq8[RF == 3 ? 8 - LDSB : 1];
  bf16x8* wlds3 = &whl[w * LDSB * 128 + lane];                           // RF = 3: piece pc of fragment kb at [(2 kb + pc) * 64]
  int sA = 127;                                                          //         E8M0 exponent of the block's codes
  float wS = 1.f;
  if constexpr (RF == 3) {
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
      const i32x4 p0 = *reinterpret_cast<const i32x4*>(wbase + 32 * 1024 + kb * 2048 + lane * 16);
      const i32x4 p1 = *reinterpret_cast<const i32x4*>(wbase + 32 * 1024 + kb * 2048 + 1024 + lane * 16);
      if (kb < LDSB) {
        wlds3[(2 * kb) * 64] = __builtin_bit_cast(bf16x8, p0);
        wlds3[(2 * kb + 1) * 64] = __builtin_bit_cast(bf16x8, p1);
      } else {
        wq8[kb >= LDSB ? kb - LDSB : 0] = __builtin_shufflevector(p0, p1, 0, 1, 2, 3, 4, 5, 6, 7);
      }
    }
    sA = __builtin_amdgcn_readfirstlane(*reinterpret_cast<const int*>(wbase + 48 * 1024 + 4));
  } else if constexpr (RF == 2) {
#pragma unroll
    for (int ks = 0; ks < LDSK8; ++ks)
      wlds8[ks * 64] = *reinterpret_cast<const u32x2*>(wbase + 32 * 1024 + ks * 512 + lane * 8);
#pragma unroll
    for (int ks = 0; ks < 32 - LDSK8; ++ks)
      wq[ks] = *reinterpret_cast<const u32x2*>(wbase + 32 * 1024 + (LDSK8 + ks) * 512 + lane * 8);
    wS = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(*reinterpret_cast<const int*>(wbase + 48 * 1024)));
  } else {
#pragma unroll
    for (int ks = 0; ks < PAIR_LDSK; ++ks)
      wlds[ks * 64] = *reinterpret_cast<const bf16x8*>(wbase + 32 * 1024 + ks * 1024 + lane * 16);
  }
  const __amdgpu_buffer_rsrc_t wrs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(wbase) + 32 * 1024, 0, 32 * 1024, 0x00020000);
  const int wlane = lane * 16;
  constexpr int NCH = 32 / PAIR_RING, CH0 = PAIR_LDSK / PAIR_RING;  // chunks per step; first streamed chunk

  rec[0][tid] = rec[1][tid] = f32x4{0.f, 0.f, 0.f, 0.f};
  const f32x4* oth = &rec[role][(wx * 2) * 64 + lane];  // X-waves read what the O-waves wrote (rec[0]) and vice versa
  f32x4* mineo = &rec[role ^ 1][(wx * 2) * 64 + lane];  // ... and write the half the other role's cells need

  gcell n_i[2], n_f[2], n_g[2], n_o[2];
  f32x4 n_dh[2], n_cp[2], c_cur[2], dc[2], mine[2];
  const f32x4 zero4v = {0.f, 0.f, 0.f, 0.f};
  auto load_step = [&](int t, int e) {
    n_i[e] = ld_gate(t, 0, e);
    n_f[e] = ld_gate(t, 1, e);
    n_g[e] = ld_gate(t, 2, e);
    n_o[e] = ld_gate(t, 3, e);
    n_dh[e] = bld(crs(p.dhcat, t), cvo, 2 * e * 512);
    const int tp = d == 0 ? max(t - 1, 0) : min(t + 1, L - 1);  // clamped; masked at its use
    n_cp[e] = bld(crs(p.cbuf, tp), cvo, 2 * e * 512);
  };
  // the same twelve addresses as load_step(t, 0 / 1), one dword per lane, normal cache policy (the loads proper are nt),
  // destination: one register nobody reads.  Inline asm: the compiler keeps no count of these loads (its own vmcnt waits
  // stay correct -- returns are in order and these are OLDER than every load it waits for) and `sink` stays allocated
  // until the asm at the end of the next cell backward, by when the loads proper -- younger -- have been consumed.
  unsigned sink = 0u;
  auto touch = [&](__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "+v"(sink) : "v"(voff), "s"(r), "s"(soff));
  };
  auto touch_step = [&](int t) {
    const int tp = d == 0 ? max(t - 1, 0) : min(t + 1, L - 1);
#pragma unroll
    for (int e = 0; e < 2; ++e) {
#pragma unroll
      for (int g = 0; g < 4; ++g) touch(hrs(t), gvo >> 1, (g * 64 + 2 * e) * 256);
      touch(crs(p.dhcat, t), cvo, 2 * e * 512);
      touch(crs(p.cbuf, tp), cvo, 2 * e * 512);
    }
  };
  {
    const int t0 = d == 0 ? L - 1 : 0;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      load_step(t0, e);
      c_cur[e] = bld(crs(p.cbuf, t0), cvo, 2 * e * 512);
      dc[e] = mine[e] = zero4v;
    }
  }
  int dead = 0;  // wave-uniform: a bounded wait of this wave timed out
  __syncthreads();
  if constexpr (V & 2048) {
    if (threadIdx.x == 0 && p.dbg_buf) {
      long long* wt = reinterpret_cast<long long*>(p.dbg_buf) + (long long)L * 16 + (pr * 2 + hs) * 4;
      wt[0] = wall_in;
      wt[1] = wall_clock64();
    }
  }

  for (int step = 0; step < L; ++step) {
    const int t = d == 0 ? L - 1 - step : step;
    const int sn = min(step + 1, L - 1);
    const int tn = d == 0 ? L - 1 - sn : sn;
    const bool has_prev = d == 0 ? (t > 0) : (t < L - 1);
    const int par = step & 1;
    int zo = 0;
    asm volatile("" : "+s"(zo));
    // ---- phase C: cell backward of this thread's two cells -> d(gates): LDS image (B operand) + HBM (BLS) -----------
    // The eight HBM stores are issued TOGETHER at the end, after every value they carry has been computed, with nothing
    // but the barrier behind them.  Found on the MI355X (profiles/r03_store_hazard.md): when the instruction after a
    // 16-byte buffer store with an SGPR soffset is a VALU write to the store's data registers, the store can pick up
    // the NEW contents in the dwords / lanes it reads last (hipcc pads that hazard only for stores WITHOUT a register
    // soffset) -- sparse, run-to-run varying wrong dwords.
    f32x4 pk[2][4];
    TS(0);
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int q = q0 + 2 * e;
      f32x4 dhr = mine[e] + oth[e * 64];
      if constexpr (RF != 0) dhr *= (1.f / 256.f);   // the fp16 weights are 256 w (exact to undo)
      f32x4 pi, pf, pg, po;
      const f32x4 vi = gate_val<false>(n_i[e]), vf = gate_val<false>(n_f[e]), vg = gate_val<true>(n_g[e]),
                  vo = gate_val<false>(n_o[e]);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float ig = vi[r], fg = vf[r], gg = vg[r], og = vo[r];
        const float dhv = dh_in<GF>(n_dh[e][r], dhr[r], dS);
        const float tc = ftanh(c_cur[e][r]);
        const float dov = dhv * tc;
        const float dcv = dc[e][r] + dhv * og * (1.f - tc * tc);
        dc[e][r] = dcv * fg;
        pi[r] = dcv * gg * ig * (1.f - ig);
        pf[r] = dcv * (has_prev ? n_cp[e][r] : 0.f) * fg * (1.f - fg);
        pg[r] = dcv * ig * (1.f - gg * gg);
        po[r] = dov * og * (1.f - og);
      }
      c_cur[e] = n_cp[e];
      auto emit = [&](const f32x4& v, int g) {
        if constexpr (RF != 0) {
          const u32x2 code = enc_f16x4(v);   // what goes to HBM IS the B operand
          *reinterpret_cast<u32x2*>(&bimg[0][n * PR_ROW + g * 128 + 4 * q]) = code;
          bst8(code, ors(t), gvo >> 1, (g * 64 + 2 * e) * 256);
          return;
        }
        bf16x4 hi, lo;
        split4(v, hi, lo);
        *reinterpret_cast<bf16x4*>(&bimg[0][n * PR_ROW + g * 128 + 4 * q]) = hi;
        *reinterpret_cast<bf16x4*>(&bimg[RF ? 0 : 1][n * PR_ROW + g * 128 + 4 * q]) = lo;
        if constexpr (G2) {
          bst8(enc_dgates<GF>(v, hi), ors(t), gvo >> 1, (g * 64 + 2 * e) * 256);
        } else {
          pk[e][g] = pack_hl4(hi, lo);
          bst(pk[e][g], grs(t), gvo, (g * 64 + 2 * e) * 512);  // (zero soffset inside: hipcc pads the data hazard)
        }
      };
      emit(pi, 0);
      emit(pf, 1);
      emit(pg, 2);
      emit(po, 3);
    }
    if constexpr (RES) asm volatile("" : "+v"(sink));   // (the touches of the previous step have returned: see touch)
    TS(1);
    __syncthreads();  // S1: the d(gates) image of this step is complete; rec is consumed
    TS(2);

    // RF = 2: the cache lines of the next step's saved state are TOUCHED here, a whole MFMA phase before the loads proper
    // (one dword per lane at the loads' own addresses, all into one throw-away register): an X-wave's poll for the partner's
    // partial sits behind those loads in the wave's in-order return queue and sat out their HBM latency (1.4 of 5.3 us per
    // step, profiles/r05_c13_recur_probe.txt); behind L2 hits it waits 0.6.  The twelve extra VMEM instructions per wave cost
    // the MFMA phase 0.3 - 0.8 us (profiles/r05_c17_recur_probe.txt), the step gains 0.6: 2.56 -> 2.39 ms per launch alone,
    // 0.5 - 1 ms per training step (r05_ab/r05_c1[67]_*).  Tried instead: three touches per chunk over the first half of the
    // MFMA loop (the same within noise); requesting the state ITSELF here (32 registers: no spills once the lo ring is gone,
    // but the B fragments lose their double buffer -- faster alone, 0.9 ms per step slower in the step, r05_c14).
    if constexpr (RES) touch_step(tn);
    // ---- partial dh^T [32 units of this wave's m-tile][32 sequences] = W_hh^T slice * dgates^T ----------------------
    if (xrole) __builtin_amdgcn_s_setprio(1);
    bf16x8 wl[2][PAIR_RING];
    if constexpr (!RES) {
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int f = 0; f < PAIR_RING; ++f)
          wl[s][f] = wload(wrs, wlane + f * 1024, zo + (CH0 + s) * (PAIR_RING * 1024));
    }
    const __bf16* bhi = &bimg[0][n * PR_ROW + 8 * half];
    const __bf16* blo = &bimg[RF ? 0 : 1][n * PR_ROW + 8 * half];
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
    v8i b8;   // RF = 3: e4m3 of the chunk's d(gates) / 256 (|fp16| / 256 < 256: inside e4m3's range, which has no infinity)
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const int s = ch & 1;
#pragma unroll
      for (int f = 0; f < PAIR_RING; ++f) {
        const int ks = PAIR_RING * ch + f;
        const bf16x8 bh = *reinterpret_cast<const bf16x8*>(bhi + 16 * ks);
        if constexpr (RF == 3) {
          // hi term on the fp16 MFMA; behind the fourth k-step of a chunk ONE FP8 MFMA takes the lo term of all 64 columns
          acc0 = pair_mfma<1>(wh[ks], bh, acc0);
          {
            typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
            typedef short s16x2 __attribute__((ext_vector_type(2)));
            const f16x8 b16 = __builtin_bit_cast(f16x8, bh);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              s16x2 c = {0, 0};
              c = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(c, f16x2{b16[4 * i], b16[4 * i + 1]}, 256.f, false);
              c = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(c, f16x2{b16[4 * i + 2], b16[4 * i + 3]}, 256.f, true);
              b8[2 * f + i] = __builtin_bit_cast(int, c);
            }
          }
          if (f == PAIR_RING - 1) {
            static_assert(PAIR_RING == 4, "a chunk is one K = 64 fragment");
            v8i a8;
            if (ch < LDSB) {
              a8 = __builtin_shufflevector(__builtin_bit_cast(i32x4, wlds3[(2 * ch) * 64]), __builtin_bit_cast(i32x4, wlds3[(2 * ch + 1) * 64]),
                                           0, 1, 2, 3, 4, 5, 6, 7);
            } else {
              a8 = wq8[ch >= LDSB ? ch - LDSB : 0];
            }
            acc1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc1, 0, 0, 0, sA, 0, 127 + 8);
          }
          continue;
        }
        if constexpr (RF == 2) {
          // (the compiler converts the five register-resident fragments ONCE, in front of the step loop: 20 registers instead
          //  of 10.  Pinning the codes with an empty asm so that they are converted every step measured 11 % slower alone.)
          const u32x2 c8 = ks < LDSK8 ? wlds8[ks * 64] : wq[ks >= LDSK8 ? ks - LDSK8 : 0];
          typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
          const f16x2 a0 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(c8[0], 1.f, false);
          const f16x2 a1 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(c8[0], 1.f, true);
          const f16x2 a2 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(c8[1], 1.f, false);
          const f16x2 a3 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(c8[1], 1.f, true);
          const f16x8 al8 = {a0[0], a0[1], a1[0], a1[1], a2[0], a2[1], a3[0], a3[1]};
          acc0 = pair_mfma<1>(wh[ks], bh, acc0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al8, __builtin_bit_cast(f16x8, bh), acc1, 0, 0, 0);
          continue;
        }
        const bf16x8 al = ch < CH0 ? wlds[ks * 64] : wl[s][f];
        if constexpr (RF != 0) {
          acc0 = pair_mfma<1>(wh[ks], bh, acc0);
          acc1 = pair_mfma<1>(al, bh, acc1);
          continue;
        }
        const bf16x8 bl = *reinterpret_cast<const bf16x8*>(blo + 16 * ks);
        if (ks & 1) {
          acc1 = mfma32(wh[ks], bh, acc1);
          acc0 = mfma32(al, bh, acc0);
          acc1 = mfma32(wh[ks], bl, acc1);
        } else {
          acc0 = mfma32(wh[ks], bh, acc0);
          acc1 = mfma32(al, bh, acc1);
          acc0 = mfma32(wh[ks], bl, acc0);
        }
      }
      if (!RES && ch >= CH0 && ch + 2 < NCH) {
#pragma unroll
        for (int f = 0; f < PAIR_RING; ++f) wl[s][f] = wload(wrs, wlane + f * 1024, zo + (ch + 2) * (PAIR_RING * 1024));
      }
      // (an explicitly double-buffered version of this loop -- B fragments of k-step ks + 1 requested before the MFMAs
      //  of ks, pinned with sched_group_barriers -- measured 4% SLOWER: the loop is not waiting for LDS but for the lo
      //  stream and the CU's memory pipe, tools/pair_diag.py --ts)
      __builtin_amdgcn_sched_barrier(0);
    }
    TS(3);
    f32x4 sum[4];
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4)
      sum[q4] = f32x4{acc0[4 * q4] + wS * acc1[4 * q4], acc0[4 * q4 + 1] + wS * acc1[4 * q4 + 1],
                      acc0[4 * q4 + 2] + wS * acc1[4 * q4 + 2], acc0[4 * q4 + 3] + wS * acc1[4 * q4 + 3]};
    if (xrole) {
      __builtin_amdgcn_s_setprio(0);
      u32x4 pv[4];
      if constexpr (RF != 0) {
        // ---- data-tagged hand-off (round 5): the LEAST significant mantissa bit of every fp32 partial carries the step's
        // tag ((step >> 1) & 1; the slots are double-buffered by step parity, so a slot's previous content -- two steps old
        // -- carries the other tag; the launcher fills the buffer with tag 1, steps 0 and 1 carry tag 0).  The producer
        // stores write-through and is done: no vmcnt drain, no flag; the consumer polls the DATA (sc1 loads), accepting it
        // when all sixteen dwords carry the tag.  2^-24 relative on a partial d(h) that is rounded to fp16 one step later.
        const unsigned tagb = (unsigned)(step >> 1) & 1u;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          u32x4 v = __builtin_bit_cast(u32x4, sum[q4]);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = (v[r] & ~1u) | tagb;
          __builtin_amdgcn_raw_buffer_store_b128(v, xrs, xc0 + (par * 2 + (1 - hs)) * PR_XSLOT + q4 * 1024, 0, SC1);
        }
        TS(4);
        // (requesting the next step's saved state BEFORE the MFMA phase instead -- so that no HBM load sits in front of the
        //  poll in this wave's in-order queue -- measured 13 % slower: 16 spilled registers and the lo stream's fragments
        //  queue behind those loads, profiles/r05_c4_recur_probe.txt)
        load_step(tn, 0);
        load_step(tn, 1);
        TS(5);
        unsigned spins = 0;
        const bool force = (V & 8) && step == 2 && pr == 0 && hs == 0 && wx == 0;  // test build: a timeout on demand
        while (true) {
          asm volatile("" ::: "memory");   // re-issue the loads every round
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4)
            pv[q4] = __builtin_amdgcn_raw_buffer_load_b128(xrs, xc0, (par * 2 + hs) * PR_XSLOT + q4 * 1024, SC1);
          unsigned bad = 0u;
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4)
#pragma unroll
            for (int r = 0; r < 4; ++r) bad |= pv[q4][r] ^ tagb;
          // wave-uniform exit: the wave's 64 lanes read one 4 KB piece the partner's wave wrote with four stores
          if ((__all((int)((bad & 1u) == 0u)) && !force) || dead || (p.dbg & 1)) break;
          if (force || ++spins > (PR_SPIN_LIMIT >> 2)) {
            if (lane == 0) pair_timed_out(p.flags + npair * 8, p.status);
            dead = 1;
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
        TS(6);
      } else {
      // publish the partial of the PARTNER's units: write-through 16-byte stores, drain, one flag per wave
      if (!(p.dbg & 2)) {
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, sum[q4]), xrs,
                                                 xc0 + (par * 2 + (1 - hs)) * PR_XSLOT + q4 * 1024, 0, SC1);
        if (p.dbg & 64) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // probe: full release instead of R1
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0)
          __hip_atomic_store(flags + hs * 4 + wx, (unsigned)(step + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      TS(4);
      // the next step's saved activations are requested BEFORE the wait for the partner: they fly while this wave
      // polls (the gather then returns behind them in the in-order queue, about when the flag arrives anyway:
      // 4.17 -> 3.93 ms per launch)
      load_step(tn, 0);
      load_step(tn, 1);
      TS(5);
      // the partner's partial of OUR units (its X-wave wx computed our m-tile wx)
      if (!dead && !(p.dbg & 3)) {
        unsigned spins = 0;
        const bool force = (V & 8) && step == 2 && pr == 0 && hs == 0 && wx == 0;  // test build: a timeout on demand
        while (true) {
          const unsigned v = __hip_atomic_load(flags + (1 - hs) * 4 + wx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (!force && v >= (unsigned)(step + 1)) break;
          if (force || ++spins > PR_SPIN_LIMIT) {
            if (lane == 0) pair_timed_out(p.flags + npair * 8, p.status);
            dead = 1;
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      TS(6);
      if (p.dbg & 64) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4)
        pv[q4] = (p.dbg & 2) ? u32x4{0u, 0u, 0u, 0u}
                             : __builtin_amdgcn_raw_buffer_load_b128(xrs, xc0, (par * 2 + hs) * PR_XSLOT + q4 * 1024, SC1);
      }
      if (dead) {
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) pv[q4] = u32x4{0x7fc00000u, 0x7fc00000u, 0x7fc00000u, 0x7fc00000u};
      }
      if (p.dbg_buf && !(V & 2048)) {  // diagnosis: what this wave sent and what it received, per step (tools/pair_diag.py)
        float* db = p.dbg_buf + ((((long long)pr * L + step) * 2 + hs) * 2) * 4096;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          *reinterpret_cast<f32x4*>(db + (xc0 >> 2) + q4 * 256) = sum[q4];
          *reinterpret_cast<u32x4*>(db + 4096 + (xc0 >> 2) + q4 * 256) = pv[q4];
        }
      }
      mine[0] = __builtin_bit_cast(f32x4, pv[0]);
      mine[1] = __builtin_bit_cast(f32x4, pv[1]);
      if constexpr (V & 2048) {
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");  // (stamp only: the gather is back, the 12 activation loads may fly)
        TS(7);
      }
      mineo[0] = __builtin_bit_cast(f32x4, pv[2]);
      mineo[64] = __builtin_bit_cast(f32x4, pv[3]);
    } else {
      mineo[0] = sum[0];
      mineo[64] = sum[1];
      mine[0] = sum[2];
      mine[1] = sum[3];
      TS(4);
      load_step(tn, 0);
      load_step(tn, 1);
    }
    __syncthreads();  // S2: both halves of every cell's sum are in place; the image is no longer read
  }
  if constexpr (V & 2048) {
    if (threadIdx.x == 0 && p.dbg_buf)
      reinterpret_cast<long long*>(p.dbg_buf)[(long long)L * 16 + (pr * 2 + hs) * 4 + 2] = wall_clock64();
  }
}

extern "C" int ws_lstm_bwd_pair(const ws_lstm_pair_args* a, void* stream) {
  WS_REQUIRE(a && a->gates && a->cbuf && a->dhcat && a->wpack && a->xchg && a->flags, "ws_lstm_bwd_pair: null pointer");
  WS_REQUIRE(a->nseq > 0 && a->L > 0, "ws_lstm_bwd_pair: nseq and L must be positive");
  WS_REQUIRE(a->gfmt >= WS_GATES_F32 && a->gfmt <= WS_GATES_H2F && (a->gfmt != WS_GATES_H2S || a->dgates) &&
                 (a->gfmt != WS_GATES_H2F || a->amax),
             "ws_lstm_bwd_pair: gfmt %d (WS_GATES_H2S needs dgates, WS_GATES_H2F needs amax)", a->gfmt);
  WS_REQUIRE(a->rfmt == 0 || (a->rfmt >= 1 && a->rfmt <= 3 && a->gfmt == WS_GATES_H2F),
             "ws_lstm_bwd_pair: rfmt %d (1 / 2 / 3 = fp16 recurrence, fp16 + fp16 / fp16 + fp8 weights / + the lo term on the FP8 "
             "MFMA: WS_GATES_H2F only)", a->rfmt);
  const int npair = 2 * ((a->nseq + SQ - 1) / SQ);
  static int cus = 0;      // same part on every device of a node; queried once
  if (!cus && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0) != hipSuccess) cus = 0;
  WS_REQUIRE(2 * npair <= cus, "ws_lstm_bwd_pair: %d workgroups must be co-resident but the device has %d CUs", 2 * npair,
             cus);
  const int grid = 16 * ((npair + 7) / 8);
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(a->flags, 0, ((size_t)npair * 8 + 8) * sizeof(unsigned), s);
  WS_REQUIRE(e == hipSuccess, "ws_lstm_bwd_pair: hipMemsetAsync failed");
  if (a->rfmt != 0) {   // data-tagged hand-off: every dword of the exchange slots starts with tag 1 (LSB set)
    e = hipMemsetAsync(a->xchg, 0x01, (size_t)npair * 4 * PR_XSLOT, s);
    WS_REQUIRE(e == hipSuccess, "ws_lstm_bwd_pair: hipMemsetAsync failed");
  }
  ws_prof_begin(WS_PROF_LSTM_BWD, s);
  if (a->rfmt == 3) {
    // (no stamped build of this variant: with the stamps in, the register allocator spills 128 registers -- not the kernel any more)
    WS_REQUIRE(!(a->dbg & 2048), "ws_lstm_bwd_pair: cycle stamps (dbg 2048) exist for rfmt 0 .. 2");
    if (a->dbg & 8) hipLaunchKernelGGL((lstm_bwd_pair_kernel<8, WS_GATES_H2F, 3>), dim3(grid), dim3(512), 0, s, *a);
    else hipLaunchKernelGGL((lstm_bwd_pair_kernel<0, WS_GATES_H2F, 3>), dim3(grid), dim3(512), 0, s, *a);
    ws_prof_end(WS_PROF_LSTM_BWD, s);
    return ws_check_launch("ws_lstm_bwd_pair");
  }
  if (a->rfmt == 2) {
    if (a->dbg & 8) hipLaunchKernelGGL((lstm_bwd_pair_kernel<8, WS_GATES_H2F, 2>), dim3(grid), dim3(512), 0, s, *a);
    else if (a->dbg & 2048) hipLaunchKernelGGL((lstm_bwd_pair_kernel<2048, WS_GATES_H2F, 2>), dim3(grid), dim3(512), 0, s, *a);
    else hipLaunchKernelGGL((lstm_bwd_pair_kernel<0, WS_GATES_H2F, 2>), dim3(grid), dim3(512), 0, s, *a);
    ws_prof_end(WS_PROF_LSTM_BWD, s);
    return ws_check_launch("ws_lstm_bwd_pair");
  }
  if (a->rfmt == 1) {
    if (a->dbg & 8) hipLaunchKernelGGL((lstm_bwd_pair_kernel<8, WS_GATES_H2F, 1>), dim3(grid), dim3(512), 0, s, *a);
    else if (a->dbg & 2048) hipLaunchKernelGGL((lstm_bwd_pair_kernel<2048, WS_GATES_H2F, 1>), dim3(grid), dim3(512), 0, s, *a);
    else hipLaunchKernelGGL((lstm_bwd_pair_kernel<0, WS_GATES_H2F, 1>), dim3(grid), dim3(512), 0, s, *a);
    ws_prof_end(WS_PROF_LSTM_BWD, s);
    return ws_check_launch("ws_lstm_bwd_pair");
  }
  switch ((a->dbg & (8 | 2048)) + a->gfmt) {
    case 0: hipLaunchKernelGGL((lstm_bwd_pair_kernel<0, 0>), dim3(grid), dim3(512), 0, s, *a); break;
    case 1: hipLaunchKernelGGL((lstm_bwd_pair_kernel<0, WS_GATES_H2>), dim3(grid), dim3(512), 0, s, *a); break;
    case 2: hipLaunchKernelGGL((lstm_bwd_pair_kernel<0, WS_GATES_H2S>), dim3(grid), dim3(512), 0, s, *a); break;
    case 3: hipLaunchKernelGGL((lstm_bwd_pair_kernel<0, WS_GATES_H2F>), dim3(grid), dim3(512), 0, s, *a); break;
    case 8: hipLaunchKernelGGL((lstm_bwd_pair_kernel<8, 0>), dim3(grid), dim3(512), 0, s, *a); break;   // tests: forced timeout
    case 9: hipLaunchKernelGGL((lstm_bwd_pair_kernel<8, WS_GATES_H2>), dim3(grid), dim3(512), 0, s, *a); break;
    case 10: hipLaunchKernelGGL((lstm_bwd_pair_kernel<8, WS_GATES_H2S>), dim3(grid), dim3(512), 0, s, *a); break;
    case 11: hipLaunchKernelGGL((lstm_bwd_pair_kernel<8, WS_GATES_H2F>), dim3(grid), dim3(512), 0, s, *a); break;
    case 2048: hipLaunchKernelGGL((lstm_bwd_pair_kernel<2048, 0>), dim3(grid), dim3(512), 0, s, *a); break;  // cycle stamps
    case 2048 + 3: hipLaunchKernelGGL((lstm_bwd_pair_kernel<2048, WS_GATES_H2F>), dim3(grid), dim3(512), 0, s, *a); break;
    default: WS_REQUIRE(false, "ws_lstm_bwd_pair: dbg bits 8 and 2048 are exclusive; cycle stamps: WS_GATES_F32 / H2F only");
  }
  ws_prof_end(WS_PROF_LSTM_BWD, s);
  return ws_check_launch("ws_lstm_bwd_pair");
}
