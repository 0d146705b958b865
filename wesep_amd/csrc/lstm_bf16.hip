// Split-bf16 ("bf16x3") bidirectional LSTM recurrence for gfx950 (nn.LSTM inside ResRNN,
// wesep/models/bsrnn.py:27-33,40): fp32 operands are carried as hi = bf16(x), lo = bf16(x - hi)
// and every product is  W_hi*h_hi + W_lo*h_hi + W_hi*h_lo  on v_mfma_f32_32x32x16_bf16 with fp32
// accumulation (the dropped lo*lo term is <= 2^-16 relative); 16x the fp32-MFMA rate per
// instruction, 5.3x per fp32-equivalent product.
//
// The product is computed TRANSPOSED:  G^T[gate col][seq] = W_hh[gate col][k] * h^T[k][seq].
//   * A operand = weight fragments, pre-split and pre-ordered once per layer by
//     ws_lstm_pack(mode 3); they cannot live in a CU (1 MB per direction), so each wave streams
//     its 128 KB slice from the XCD's L2 every step through a 2-slot register ring that never
//     drains across step boundaries (the stream is the same every step).
//   * B operand = h_{t-1} (fwd) / d(gates) (bwd) of the workgroup's 32 sequences, bf16 hi/lo in
//     LDS, one ds_read_b128 per fragment, row stride = 4 banks mod 64 -> conflict-free.
//   * D: lane = one sequence (lane & 31), registers = 16 hidden units in 4 runs of 4 consecutive
//     units; wave w owns units [32w, 32w+32) of all four gates, so the cell update is lane-local
//     and every global load/store of gates / c / h is a 16-byte vector per lane.
// Sequences are columns of the MFMA, so they cannot contaminate each other; a padded lane simply
// duplicates the last valid sequence of its workgroup (same wave, same instruction, same address,
// same value), which keeps the step body free of branches -- a single basic block, so the
// compiler's s_waitcnt counting stays exact and the one-step-ahead HBM prefetches never drain.
//
// Per step a 32-sequence workgroup needs 1 MB from L2 (~7.5 us at the per-CU L1 fill rate) against
// ~5 us of MFMA: the kernel is L2-stream bound in the time view, by design -- see DESIGN.md.
#include <stdlib.h>

#include "lstm_bf16_common.h"

// ---------------------------------------------------------------------------------------------
// weight packing (16-byte units of 8 bf16; `lane` = the MFMA lane that will load the unit)
//   fwd: unit ((((d*8 + w)*16 + ks)*4 + g)*2 + part)*64 + lane, element j
//          = part( W_hh[d][ g*256 + 32w + (lane&31) ][ 16ks + 8(lane>>5) + j ] )
//   bwd: unit (((d*8 + w)*64 + ks)*2 + part)*64 + lane, element j
//          = part( W_hh[d][ 16ks + 8(lane>>5) + j ][ 32w + (lane&31) ] )
// ---------------------------------------------------------------------------------------------
__global__ void lstm_pack_bf16_kernel(const float* __restrict__ whh_f, const float* __restrict__ whh_r,
                                      __bf16* __restrict__ pf, __bf16* __restrict__ pb) {
  const int total = 2 * LG * LH;  // weights per pass
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    {
      int r = idx;
      const int j = r & 7; r >>= 3;
      const int lane = r & 63; r >>= 6;
      const int g = r & 3; r >>= 2;
      const int ks = r & 15; r >>= 4;
      const int w = r & 7; r >>= 3;
      const int d = r;
      const float* W = d ? whh_r : whh_f;
      const int row = g * 256 + 32 * w + (lane & 31);
      const int k = 16 * ks + 8 * (lane >> 5) + j;
      const float v = W[row * LH + k];
      const __bf16 hi = (__bf16)v;
      const long long unit = ((((long long)(d * 8 + w) * 16 + ks) * 4 + g) * 2) * 64 + lane;
      pf[unit * 8 + j] = hi;
      pf[(unit + 64) * 8 + j] = (__bf16)(v - (float)hi);
    }
    {
      int r = idx;
      const int j = r & 7; r >>= 3;
      const int lane = r & 63; r >>= 6;
      const int ks = r & 63; r >>= 6;
      const int w = r & 7; r >>= 3;
      const int d = r;
      const float* W = d ? whh_r : whh_f;
      const int row = 16 * ks + 8 * (lane >> 5) + j;  // gate column = contraction index
      const int u = 32 * w + (lane & 31);
      const float v = W[row * LH + u];
      const __bf16 hi = (__bf16)v;
      const long long unit = (((long long)(d * 8 + w) * 64 + ks) * 2) * 64 + lane;
      pb[unit * 8 + j] = hi;
      pb[(unit + 64) * 8 + j] = (__bf16)(v - (float)hi);
    }
  }
}

int ws_launch_lstm_pack_bf16(const float* whh_f, const float* whh_r, float* pack_fwd, float* pack_bwd,
                             hipStream_t s) {
  hipLaunchKernelGGL(lstm_pack_bf16_kernel, dim3(512), dim3(256), 0, s, whh_f, whh_r,
                     reinterpret_cast<__bf16*>(pack_fwd), reinterpret_cast<__bf16*>(pack_bwd));
  return 0;
}

// BPTT pack of ws_lstm_args.rfmt = 2 (ABI v18): W_hh as fp16 hi of 256 w + OCP e4m3 codes of the remainder over a power-of-two
// scale -- what lstm_pair.hip's rfmt 2 keeps resident, here as a STREAM of 96 instead of 128 KB per wave and step.  Per (d, w)
// region of 128 KB (the bf16 pack's): 16 chunks of 4 k-steps, chunk c at c * 6 KB = [4 hi fragments of 1 KB: lane * 16 bytes]
// [4 code fragments of 512 B: lane * 8 bytes]; at byte 96 K eight floats, the scale of each GROUP of 8 k-steps (two chunks):
// S = 2^(e - 20) for the group's max |256 w| in [2^(e-1), 2^e), so that the largest possible remainder maps to 256 (e4m3
// returns NaN above 448).  One workgroup per (d, w, group).  Element order as the bf16 BPTT pack above.
__global__ __launch_bounds__(512) void lstm_pack_bwd_f8_kernel(const float* __restrict__ whh_f, const float* __restrict__ whh_r,
                                                               char* __restrict__ pb) {
  __shared__ float red[8];
  const int grp = blockIdx.x & 7, w = (blockIdx.x >> 3) & 7, d = blockIdx.x >> 6;
  const float* W = d ? whh_r : whh_f;
  const int tid = threadIdx.x;
  float v0[4], v1[4], m = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {   // 8 k-steps x 64 lanes x 4 element pairs = 2048 pairs, 4 per thread
    const int pi = tid + 512 * i, j2 = pi & 3, lane = (pi >> 2) & 63, ks = 8 * grp + (pi >> 8);
    const int row = 16 * ks + 8 * (lane >> 5) + 2 * j2, u = 32 * w + (lane & 31);
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
  const float S = (eb > 19 && eb < 255) ? __builtin_bit_cast(float, (unsigned)(eb - 19) << 23) : 1.f;
  char* ob = pb + (long long)(d * 8 + w) * (128 * 1024);
  if (tid == 0) reinterpret_cast<float*>(ob + 96 * 1024)[grp] = S;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int pi = tid + 512 * i, j2 = pi & 3, lane = (pi >> 2) & 63, ks = 8 * grp + (pi >> 8);
    const _Float16 h0 = (_Float16)v0[i], h1 = (_Float16)v1[i];
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    typedef short s16x2 __attribute__((ext_vector_type(2)));
    char* cb = ob + (ks >> 2) * 6144;
    *reinterpret_cast<f16x2*>(cb + (ks & 3) * 1024 + lane * 16 + j2 * 4) = f16x2{h0, h1};
    const s16x2 c = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(s16x2{0, 0}, v0[i] - (float)h0, v1[i] - (float)h1, S, false);
    *reinterpret_cast<short*>(cb + 4096 + (ks & 3) * 512 + lane * 8 + j2 * 2) = c[0];
  }
}

int ws_launch_lstm_pack_bwd_f8(const float* whh_f, const float* whh_r, float* pack_bwd, hipStream_t s) {
  hipLaunchKernelGGL(lstm_pack_bwd_f8_kernel, dim3(128), dim3(512), 0, s, whh_f, whh_r, reinterpret_cast<char*>(pack_bwd));
  return 0;
}

// d(xn) inside the BPTT (ws_lstm_args.dxn, ABI v19): the A operand W_ih^T of  d(xn)^T[input][seq] = W_ih^T[input][gate col] *
// d(gates)^T[gate col][seq]  in the arithmetic of rfmt 2 -- fp16 hi of 256 w + e4m3 codes of the remainder over a power-of-two
// scale -- on the SAME v_mfma_f32_32x32x16_f16 B fragments the recurrent product reads from the LDS image (a second pass over
// the image with the 16 x 16 shape was measured: the kernel became bound by LDS reads, 1.98 -> 2.65 ms per launch).  wcat:
// [2][4H][128] (ws_lstm_cat_ih).  Wave w of direction d owns the 32 inputs [32 (w & 3), + 32) for the k-steps ks with
// (ks & 1) == (w >> 2); the two halves' partial sums meet in LDS.  Per (d, w) region of WS_DX_REGION bytes: 16 chunks (one per
// four k-steps of the W_hh stream), chunk c at c * 3 KB = [2 hi fragments of 1 KB: lane * 16 bytes][2 code fragments of 512 B:
// lane * 8 bytes] for ks = 4c + (w >> 2) and 4c + (w >> 2) + 2; at byte 48 K eight floats, the scale of each GROUP of two chunks.
// Fragment (ks, lane): element j = 256 * wcat[d][16 ks + 8 (lane >> 5) + j][32 (w & 3) + (lane & 31)].  One workgroup per
// (d, w, group).
#define WS_DX_REGION (48 * 1024 + 64)
__global__ __launch_bounds__(512) void lstm_pack_dx_f8_kernel(const float* __restrict__ wcat, char* __restrict__ px) {
  __shared__ float red[8];
  const int grp = blockIdx.x & 7, w = (blockIdx.x >> 3) & 7, d = blockIdx.x >> 6;
  const float* W = wcat + (long long)d * LG * 128;
  const int tid = threadIdx.x;
  float v0[2], v1[2], m = 0.f;
  auto ks_of = [&](int f) { return 4 * (2 * grp + (f >> 1)) + (w >> 2) + 2 * (f & 1); };   // f = 0..3: the group's four k-steps
#pragma unroll
  for (int i = 0; i < 2; ++i) {   // 4 k-steps x 64 lanes x 4 element pairs = 1024 pairs, 2 per thread
    const int pi = tid + 512 * i, j2 = pi & 3, lane = (pi >> 2) & 63, ks = ks_of(pi >> 8);
    const int row = 16 * ks + 8 * (lane >> 5) + 2 * j2, u = 32 * (w & 3) + (lane & 31);
    v0[i] = 256.f * W[row * 128 + u];
    v1[i] = 256.f * W[(row + 1) * 128 + u];
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
  const float S = (eb > 19 && eb < 255) ? __builtin_bit_cast(float, (unsigned)(eb - 19) << 23) : 1.f;
  char* ob = px + (long long)(d * 8 + w) * WS_DX_REGION;
  if (tid == 0) reinterpret_cast<float*>(ob + 48 * 1024)[grp] = S;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int pi = tid + 512 * i, j2 = pi & 3, lane = (pi >> 2) & 63, f = pi >> 8;
    const _Float16 h0 = (_Float16)v0[i], h1 = (_Float16)v1[i];
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    typedef short s16x2 __attribute__((ext_vector_type(2)));
    char* cb = ob + (2 * grp + (f >> 1)) * 3072;
    *reinterpret_cast<f16x2*>(cb + (f & 1) * 1024 + lane * 16 + j2 * 4) = f16x2{h0, h1};
    const s16x2 c = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(s16x2{0, 0}, v0[i] - (float)h0, v1[i] - (float)h1, S, false);
    *reinterpret_cast<short*>(cb + 2048 + (f & 1) * 512 + lane * 8 + j2 * 2) = c[0];
  }
}

int ws_launch_lstm_pack_dx_f8(const float* wcat, float* pack, hipStream_t s) {
  hipLaunchKernelGGL(lstm_pack_dx_f8_kernel, dim3(128), dim3(512), 0, s, wcat, reinterpret_cast<char*>(pack));
  return 0;
}

// ---------------------------------------------------------------------------------------------
// forward recurrence
// ---------------------------------------------------------------------------------------------
// DBG (probe builds only, mode bits 8..10): 1 = skip global stores, 2 = skip the x-projection /
// step-input prefetch, 4 = skip the weight-stream refills.  DBG = 0 is the product kernel.
//
// BLK selects the activation layout: false = plain rows ([P][2][4H] gates, [P][2H] c/h, rows by
// the sequence map), true = the blocked layout BL of include/wesep_hip.h, in which the 16 bytes
// a lane moves and those of its 31 neighbours are one contiguous 512-byte run (block = the
// workgroup's 32 sequences at one step), so every wave-level load/store is 1 KB contiguous.
// GF (BLK only): WS_GATES_* storage of the saved gates; GF != 0: pre-activations from p.gates_in (fp32 BL), activated
// gates to p.gates as unorm16 (BLH) -- lstm_bf16_common.h
template <bool BLK, int DBG, int GF = 0>
__global__ __launch_bounds__(512, 2) void lstm_fwd_bf16_kernel(const ws_lstm_args p) {
  __shared__ __attribute__((aligned(16))) __bf16 hl[2][2][SQ * HROW];  // [buf][part][seq][k] 66 KB
  __shared__ __attribute__((aligned(16))) float cl[SQ * (LH + 4)];     // cell state [seq][unit] 33 KB
  if (p.run_if && *p.run_if == 0u) return;  // predicated fall-back launch (wesep_hip.h): uniform
  const int d = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int L = p.L;

  {  // h_{-1} = 0
    uint32_t* z = reinterpret_cast<uint32_t*>(&hl[0][0][0]);
    for (int i = tid; i < 2 * SQ * HROW / 2; i += 512) z[i] = 0u;
  }
  const int ss = min((int)blockIdx.x * SQ + l31, p.nseq - 1);  // plain: padded lanes duplicate the last sequence
  const long long rowbase =
      BLK ? 0 : (long long)(ss / p.sq_div) * p.sq_s1 + (long long)(ss % p.sq_div) * p.sq_s2;
  const int ubase = 32 * w + 4 * half;  // unit of register 4j + r: ubase + 8j + r
  // activation I/O.  BLK: one SGPR buffer descriptor per (array, step) block + one VGPR lane offset
  // (no 64-bit VGPR addresses); plain: row pointers.
  const int glane = ((d * 256 + 8 * w + half) * 32 + l31) * 16;  // bytes; + (g*64 + 2j)*512
  const int clane = ((d * 64 + 8 * w + half) * 32 + l31) * 16;   // bytes; + 2j*512
  auto goff = [&](int t, int g, int j) -> long long {
    return ((rowbase + (long long)t * p.step_rows) * 2 + d) * LG + g * 256 + ubase + 8 * j;
  };
  auto coff = [&](int t, int j) -> long long {
    return (rowbase + (long long)t * p.step_rows) * (2 * LH) + d * LH + ubase + 8 * j;
  };
  const float* gsrc = GF ? p.gates_in : p.gates;
  auto grs = [&](int t) { return mkrsrc(gsrc + (long long)(blockIdx.x * L + t) * (SQ * 2 * LG), SQ * 2 * LG * 4); };
  auto hrs = [&](int t) { return mkrsrc(p.gates + (long long)(blockIdx.x * L + t) * (SQ * LG), SQ * 2 * LG * 2); };  // BLH
  auto crs = [&](float* b, int t) { return mkrsrc(b + (long long)(blockIdx.x * L + t) * (SQ * 2 * LH), SQ * 2 * LH * 4); };
  auto ld_gate = [&](int t, int g, int j) -> f32x4 {
    if constexpr (BLK) return bld(grs(t), glane, (g * 64 + 2 * j) * 512);
    else return *reinterpret_cast<const f32x4*>(p.gates + goff(t, g, j));
  };
  auto st_gate = [&](const f32x4& v, int t, int g, int j) {
    if constexpr (GF != 0) bst8(g == 2 ? enc_u16x4<true>(v) : enc_u16x4<false>(v), hrs(t), glane >> 1, (g * 64 + 2 * j) * 256);
    else if constexpr (BLK) bst(v, grs(t), glane, (g * 64 + 2 * j) * 512);
    else *reinterpret_cast<f32x4*>(p.gates + goff(t, g, j)) = v;
  };
  auto st_ch = [&](const f32x4& v, float* b, int t, int j) {
    if constexpr (BLK) bst(v, crs(b, t), clane, 2 * j * 512);
    else *reinterpret_cast<f32x4*>(b + coff(t, j)) = v;
  };

  float* cme = &cl[l31 * (LH + 4) + ubase];  // this lane's 4 runs of 4 units: cme + 8j (private)
#pragma unroll
  for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(cme + 8 * j) = f32x4{0.f, 0.f, 0.f, 0.f};

  // weight stream: per k-step 8 fragments (4 gates x {hi, lo}), 1 KB each per wave
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.wpack) + (long long)(d * 8 + w) * (16 * 8 * 64 * 4), 0, 16 * 8 * 1024, 0x00020000);
  const int wlane = lane * 16;
  bf16x8 wr[2][8];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int f = 0; f < 8; ++f) wr[s][f] = wload(wrs, wlane + (f & 3) * 1024, s * 8192 + (f >> 2) * 4096);

  // x-projection (+biases) of the next step, prefetched one step ahead
  f32x4 xg[4][4];  // [gate][run]
  {
    const int t0 = d == 0 ? 0 : L - 1;
#pragma unroll
    for (int gi = 0; gi < 4; ++gi)
#pragma unroll
      for (int j = 0; j < 4; ++j) xg[gi][j] = ld_gate(t0, gi, j);
  }
  __syncthreads();

  for (int step = 0; step < L; ++step) {
    const int t = d == 0 ? step : L - 1 - step;
    const int cur = step & 1;
    int zo = 0;
    asm volatile("" : "+s"(zo));
    const __bf16* hhi = &hl[cur][0][l31 * HROW + 8 * half];
    const __bf16* hlo = &hl[cur][1][l31 * HROW + 8 * half];
    f32x16 acc[4];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const int s = ks & 1;
      const bf16x8 bh = *reinterpret_cast<const bf16x8*>(hhi + 16 * ks);
      const bf16x8 bl = *reinterpret_cast<const bf16x8*>(hlo + 16 * ks);
      if (ks == 0) {
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = mfma32(wr[s][2 * g], bh, zero);
      } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = mfma32(wr[s][2 * g], bh, acc[g]);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[g] = mfma32(wr[s][2 * g + 1], bh, acc[g]);
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[g] = mfma32(wr[s][2 * g], bl, acc[g]);
      // refill this slot with k-step ks+2 (wraps into the next step: the stream never drains)
      const int kn = (ks + 2) & 15;
      if (!(DBG & 4)) {
#pragma unroll
        for (int f = 0; f < 8; ++f)
          wr[s][f] = wload(wrs, wlane + (f & 3) * 1024, zo + kn * 8192 + (f >> 2) * 4096);
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the k-steps in program order: loads stay 2 k-steps ahead
    }

    // cell update, lane-local; global traffic as 16-byte vectors
    const int sn = min(step + 1, L - 1);  // last step: harmless reload of its own row
    const int tn = d == 0 ? sn : L - 1 - sn;
    __bf16* nhi = &hl[cur ^ 1][0][l31 * HROW + ubase];
    __bf16* nlo = &hl[cur ^ 1][1][l31 * HROW + ubase];
    // pre-activations = recurrent part + x-projection; then ALL of the next step's x-projection
    // loads at once, as early as possible: every later load of this wave (the weight stream of the
    // next step) returns behind them, so their HBM latency has to start running now
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[g][4 * j + r] += xg[g][j][r];
    if (!(DBG & 2)) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) xg[g][j] = ld_gate(tn, g, j);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x4 pre[4];
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) pre[g][r] = acc[g][4 * j + r];
      f32x4 vi, vf, vg, vo, vc, vh;
      const f32x4 cold = *reinterpret_cast<const f32x4*>(cme + 8 * j);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float ig = fsig(pre[0][r]);
        const float fg = fsig(pre[1][r]);
        const float gg = ftanh(pre[2][r]);
        const float og = fsig(pre[3][r]);
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
      *reinterpret_cast<bf16x4*>(nhi + 8 * j) = h_hi;
      *reinterpret_cast<bf16x4*>(nlo + 8 * j) = h_lo;
      *reinterpret_cast<f32x4*>(cme + 8 * j) = vc;
      if (!(DBG & 1) || step == L - 1) {
        st_gate(vi, t, 0, j);
        st_gate(vf, t, 1, j);
        st_gate(vg, t, 2, j);
        st_gate(vo, t, 3, j);
        st_ch(vc, p.cbuf, t, j);
        if constexpr (BLK) st_ch(pack_hl4(h_hi, h_lo), p.hcat, t, j);  // BLS: h leaves as the split pair
        else st_ch(vh, p.hcat, t, j);
      }
      __builtin_amdgcn_sched_barrier(0);  // one run at a time: bounds the live temporaries
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// backward recurrence (BPTT): walks the steps in the reverse of the forward order.
//   dh_{t-1}^T[unit][seq] = W_hh^T[unit][gate col] * dgates_t^T[gate col][seq]   (K = 1024)
// ---------------------------------------------------------------------------------------------
// GF (BLK only): WS_GATES_H2: unorm16 gates in, bf16 d(gates) out -- in place on the BLH buffer, or to p.dgates (BLH) when
// given, which leaves the saved gates intact (the predicated fall-back behind ws_lstm_bwd_pair); WS_GATES_H2S: unorm16
// gates in, d(gates) as BLS pairs to p.dgates
// RF = 2 (ws_lstm_args.rfmt, ABI v18; BLK and WS_GATES_H2F only): the recurrent product on v_mfma_f32_32x32x16_f16 with the STORED
// scaled-fp16 d(gates) as its one B operand (one LDS image plane) against W_hh as fp16 hi + scaled-FP8 lo of 256 w
// (lstm_pack_bwd_f8_kernel): two MFMAs per product instead of three and 96 instead of 128 KB streamed per wave and step; the
// codes become fp16 fragments on the way in (v_cvt_scalef32_pk_f16_fp8 with the group's scale: no separate accumulator scale).
// RF = 3 (ABI v20): RF = 2's stream and codes with the lo term on v_mfma_scale_f32_32x32x64_f8f6f4 -- a chunk's four 8-byte code
// units ARE a lane's 32 operand bytes of a K = 64 fragment (the instruction pairs byte b of A's lane (row, half) with byte b of B's
// lane (seq, half); which column a byte holds is ours to choose), the B operand is e4m3 of the chunk's four fp16 d(gates)
// fragments / 256, converted in registers, the group's power-of-two scale goes in as the instruction's E8M0 exponent: four fp16
// MFMAs + one FP8 MFMA at twice the rate per chunk instead of eight (profiles/r06_c20_f8_probe.txt).
// DX (ABI v19, RF = 2 only): d(xn) = d(gates) W_ih of this direction computed HERE, from the B fragments of the d(gates) image
// the recurrent product loads anyway: wave w adds the 32 x 32 tile (inputs [32 (w & 3), + 32) x the workgroup's 32 sequences)
// over the k-steps of its parity (w >> 2) -- 64 more MFMAs per wave and step beside the 128 of d(h): the matrix pipe was idle two
// thirds of this kernel's step --, streams its W_ih^T slice beside W_hh (48 KB per wave and step: lstm_pack_dx_f8_kernel), the two
// parities' partial tiles meet in 16 KB of LDS behind the step's closing barrier, and waves 0..3 store plain rows of p.dxn + d *
// p.dxn_dir_stride at the sequence map's positions.  ws_gemm_b2p(a_fmt 2) over d(gates) -- 2.1 GB read per band-view layer at
// R = 32 -- is not launched; the GroupNorm backward adds the two directions (ws_gn_bwd_fused2).
template <bool BLK, int DBG, int GF = 0, int RF = 0, bool DX = false>
__global__ __launch_bounds__(512, 2) void lstm_bwd_bf16_kernel(const ws_lstm_args p) {
  static_assert(RF == 0 || (BLK && GF == WS_GATES_H2F), "the fp16 recurrence takes the scaled-fp16 d(gates) of WS_GATES_H2F");
  static_assert(!DX || RF == 2, "d(xn) in the BPTT rides on the fp16 d(gates) image of rfmt 2");
  __shared__ __attribute__((aligned(16))) __bf16 dgl[RF ? 1 : 2][SQ * DROW];  // [part][seq][gate col] 129 / 65 KB
  __shared__ __attribute__((aligned(16))) f32x4 xred[DX ? 8 * 2 * 64 : 1];    // DX: the half of each wave's partial d(xn) tile its partner stores, 16 KB
  if (p.run_if && *p.run_if == 0u) return;  // predicated fall-back launch (wesep_hip.h): uniform
  // (round 6 tried the two directions interleaved in one grid dimension, so that every XCD serves both weight streams at any
  //  time instead of all 256 CUs pulling ONE direction's fragments in near lock-step: no difference, 2.03 vs 2.02 ms --
  //  profiles/r06_c9_band_probe_dm{0,1}.txt; the L2 -> CU stream bound is not a hot spot of that kind)
  const int d = blockIdx.y, bx = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int L = p.L;
  const int ss = min(bx * SQ + l31, p.nseq - 1);  // plain: padded lanes duplicate the last sequence
  const long long rowbase =
      BLK ? 0 : (long long)(ss / p.sq_div) * p.sq_s1 + (long long)(ss % p.sq_div) * p.sq_s2;
  const int ubase = 32 * w + 4 * half;
  const int glane = ((d * 256 + 8 * w + half) * 32 + l31) * 16;  // see lstm_fwd_bf16_kernel
  const int clane = ((d * 64 + 8 * w + half) * 32 + l31) * 16;
  auto goff = [&](int t, int g, int j) -> long long {
    return ((rowbase + (long long)t * p.step_rows) * 2 + d) * LG + g * 256 + ubase + 8 * j;
  };
  auto coff = [&](int t, int j) -> long long {
    return (rowbase + (long long)t * p.step_rows) * (2 * LH) + d * LH + ubase + 8 * j;
  };
  float* gdst = GF == WS_GATES_H2S ? p.dgates : p.gates;
  auto grs = [&](int t) { return mkrsrc(gdst + (long long)(bx * L + t) * (SQ * 2 * LG), SQ * 2 * LG * 4); };
  auto hrs = [&](int t) { return mkrsrc(p.gates + (long long)(bx * L + t) * (SQ * LG), SQ * 2 * LG * 2); };  // BLH
  constexpr bool G2 = GF == WS_GATES_H2 || GF == WS_GATES_H2F;  // 2-byte d(gates): bf16, or fp16 scaled by dS
  float* hdst = (G2 && p.dgates) ? p.dgates : p.gates;
  auto ors = [&](int t) { return mkrsrc(hdst + (long long)(bx * L + t) * (SQ * LG), SQ * 2 * LG * 2); };     // BLH out
  const float dS = GF == WS_GATES_H2F ? ws_dgates_scale(*p.amax) : 1.f;
  auto crs = [&](const float* b, int t) { return mkrsrc(b + (long long)(bx * L + t) * (SQ * 2 * LH), SQ * 2 * LH * 4); };
  typedef typename gate_cell<GF>::type gcell;
  auto ld_gate = [&](int t, int g, int j) -> gcell {
    if constexpr (GF != 0) return bld8(hrs(t), glane >> 1, (g * 64 + 2 * j) * 256);
    else if constexpr (BLK) return bld(grs(t), glane, (g * 64 + 2 * j) * 512);
    else return *reinterpret_cast<const f32x4*>(p.gates + goff(t, g, j));
  };
  auto st_gate = [&](const f32x4& v, int t, int g, int j) {
    if constexpr (BLK) bst(v, grs(t), glane, (g * 64 + 2 * j) * 512);
    else *reinterpret_cast<f32x4*>(p.gates + goff(t, g, j)) = v;
  };
  auto ld_ch = [&](const float* b, int t, int j) -> f32x4 {
    if constexpr (BLK) return bld(crs(b, t), clane, 2 * j * 512);
    else return *reinterpret_cast<const f32x4*>(b + coff(t, j));
  };

  // this wave's d(xn) k-steps of a chunk are kpar and kpar + 2, kpar = w >> 2 (the pack's order).  Waves 4..7 walk every PAIR of
  // k-steps in swapped order (ring slot q holds k-step q ^ kpar: a uniform address bit in wfill, a second base address for the B
  // fragments), so that for every wave the even slots are its d(xn) k-steps -- static register indices, no branch, no select;
  // d(h) sums the same products in another order
  const int kpar = DX ? (w >> 2) : 0;
  // weight stream: per k-step 2 fragments (hi, lo); ring slots hold 4 k-steps
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.wpack) + (long long)(d * 8 + w) * (64 * 2 * 64 * 4), 0, 64 * 2 * 1024, 0x00020000);
  const int wlane = lane * 16;
  bf16x8 wr[2][RF ? 4 : 8];
  u32x2 wq[2][RF ? 4 : 1];   // RF = 2: the FP8 codes of the slot's four k-steps
  float wS[RF ? 8 : 1];      // RF = 2: scale per group of 8 k-steps (uniform)
  int wE[RF == 3 ? 8 : 1];   // RF = 3: the same scales as E8M0 exponents
  auto wfill = [&](int s, int chunk, int zo) {   // ring slot s <- chunk (4 k-steps) of the stream
    if constexpr (RF != 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int fq = q ^ kpar;   // (DX, waves 4..7: the pair's k-steps in swapped order -- see kpar)
        wr[s][q] = wload(wrs, wlane, zo + chunk * 6144 + fq * 1024);
        wq[s][q] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(wrs, lane * 8, zo + chunk * 6144 + 4096 + fq * 512, 0));
      }
    } else {
#pragma unroll
      for (int f = 0; f < 8; ++f) wr[s][f] = wload(wrs, wlane + (f & 3) * 1024, zo + chunk * 8192 + (f >> 2) * 4096);
    }
  };
  // DX: the W_ih^T stream of this wave's (input tile, k-step parity) (lstm_pack_dx_f8_kernel): two k-steps per chunk
  typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
  typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(p.wxpack)) + (DX ? (long long)(d * 8 + w) * WS_DX_REGION : 0), 0,
      DX ? WS_DX_REGION : 0, 0x00020000);
  bf16x8 xr[2][DX ? 2 : 1];
  u32x2 xq[2][DX ? 2 : 1];
  float xS[DX ? 8 : 1];
  f32x16 dxa;
  auto xfill = [&](int s, int chunk, int zo) {
    if constexpr (DX) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        xr[s][q] = wload(xrs, wlane + q * 1024, zo + chunk * 3072);
        xq[s][q] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(xrs, lane * 8 + q * 512, zo + chunk * 3072 + 2048, 0));
      }
    }
  };
  if constexpr (RF != 0) {
    const float* st = p.wpack + (long long)(d * 8 + w) * (64 * 2 * 64 * 4) + 96 * 256;
#pragma unroll
    for (int g8 = 0; g8 < 8; ++g8) wS[g8] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, st[g8])));
    if constexpr (RF == 3) {
#pragma unroll
      for (int g8 = 0; g8 < 8; ++g8) wE[g8] = (__builtin_bit_cast(int, wS[g8]) >> 23) & 255;
    }
  }
  if constexpr (DX) {
    const float* st = reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.wxpack) + (long long)(d * 8 + w) * WS_DX_REGION + 48 * 1024);
#pragma unroll
    for (int g8 = 0; g8 < 8; ++g8) xS[g8] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, st[g8])));
  }
  // DX: this lane's d(xn) cells -- D of the 32 x 32 tile: sequence l31, inputs 32 (w & 3) + 8 j + 4 half .. + 3, j = 0..3 -- in
  // the plain [P][128] buffer of this direction, as a 32-bit byte offset of a buffer store: row = (seq / sq_div) * sq_s1 + (seq
  // % sq_div) * sq_s2 (+ t * step_rows, added per step) -- ws_seqmap; a padded slot gets an offset beyond the descriptor's size
  // (the buffer is < 2 GB: lstm_check): the hardware drops the store
  unsigned xoff = 0u;
  if constexpr (DX) {
    const int sq = bx * SQ + l31;
    const long long row = (long long)(sq / p.sq_div) * p.sq_s1 + (long long)(sq % p.sq_div) * p.sq_s2;
    // (this wave stores the rows j = 2 kpar, 2 kpar + 1 of the tile: inputs 32 mt + 16 kpar + 8 i + 4 half .. + 3, i = 0, 1)
    xoff = sq < p.nseq ? (unsigned)(row * 512 + (32 * (w & 3) + 16 * kpar + 4 * half) * 4) : 0x80000000u;   // (+ t * step_rows * 512 < 2^31)
  }
  const __amdgpu_buffer_rsrc_t xors = __builtin_amdgcn_make_buffer_rsrc(
      DX ? p.dxn + (long long)d * p.dxn_dir_stride : nullptr, 0, DX ? (unsigned)(p.dxn_dir_stride * 4) : 0u, 0x00020000);
  gcell n_i[4], n_f[4], n_g[4], n_o[4];
  f32x4 n_dh[4], n_cp[4], c_cur[4], dc[4];  // [run]
  f32x16 dhr;
  const f32x4 zero4v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 4; ++j) dc[j] = zero4v;
#pragma unroll
  for (int i = 0; i < 16; ++i) dhr[i] = 0.f;

  // inputs of step `t` for register run j (c_cur is carried: c_t = the previous step's c_{prev})
  auto load_step = [&](int t, int j) {
    n_i[j] = ld_gate(t, 0, j);
    n_f[j] = ld_gate(t, 1, j);
    n_g[j] = ld_gate(t, 2, j);
    n_o[j] = ld_gate(t, 3, j);
    n_dh[j] = ld_ch(p.dhcat, t, j);
    // c_{t-1}; at the sequence start the step is clamped and the value is masked at its use
    const int tp = d == 0 ? max(t - 1, 0) : min(t + 1, L - 1);
    n_cp[j] = ld_ch(p.cbuf, tp, j);
  };
  {
    const int t0 = d == 0 ? L - 1 : 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      load_step(t0, j);
      c_cur[j] = ld_ch(p.cbuf, t0, j);
    }
  }
  // the ring's first two chunks BEHIND the first step's inputs, in the order of the steady state (there the inputs of step t + 1
  // are requested in the cell phase of step t, the ring refills follow in its MFMA loop): the loop header's s_waitcnt is the
  // minimum over both ways into it, and with the ring first the prologue made it vmcnt(0) -- every step then drained the ring's
  // fragments and the acknowledgements of its last stores before its cell phase (found in the ISA of the DX instantiation)
  __builtin_amdgcn_sched_barrier(0);
  wfill(0, 0, 0);
  xfill(0, 0, 0);
  wfill(1, 1, 0);
  xfill(1, 1, 0);
  __builtin_amdgcn_sched_barrier(0);

  for (int step = 0; step < L; ++step) {
    const int t = d == 0 ? L - 1 - step : step;
    const int sn = min(step + 1, L - 1);  // last step: harmless reload of its own row
    const int tn = d == 0 ? L - 1 - sn : sn;
    const bool has_prev = d == 0 ? (t > 0) : (t < L - 1);  // uniform
    int zo = 0;  // see wload()
    asm volatile("" : "+s"(zo));
    __bf16* dhi = &dgl[0][l31 * DROW + ubase];
    __bf16* dlo = &dgl[RF ? 0 : 1][l31 * DROW + ubase];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x4 pi, pf, pg, po;
      const f32x4 vi = gate_val<false>(n_i[j]), vf = gate_val<false>(n_f[j]), vg = gate_val<true>(n_g[j]),
                  vo = gate_val<false>(n_o[j]);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float ig = vi[r], fg = vf[r], gg = vg[r], og = vo[r];
        const float dhv = dh_in<GF>(n_dh[j][r], dhr[4 * j + r], dS);
        const float tc = ftanh(c_cur[j][r]);
        const float dov = dhv * tc;
        const float dcv = dc[j][r] + dhv * og * (1.f - tc * tc);
        dc[j][r] = dcv * fg;
        pi[r] = dcv * gg * ig * (1.f - ig);
        pf[r] = dcv * (has_prev ? n_cp[j][r] : 0.f) * fg * (1.f - fg);
        pg[r] = dcv * ig * (1.f - gg * gg);
        po[r] = dov * og * (1.f - og);
      }
      c_cur[j] = n_cp[j];
      // d(gates) go to the LDS image (B operand of this step's product) and to HBM; blocked layout: as the same
      // split pair (BLS) that the weight-gradient and d(x) GEMMs consume
      const bool st = !(DBG & 1) || step == L - 1;
      auto emit = [&](const f32x4& v, int g) {
        if constexpr (RF != 0) {
          const u32x2 code = enc_f16x4(v);   // what goes to HBM IS the B operand
          *reinterpret_cast<u32x2*>(dhi + 256 * g + 8 * j) = code;
          if (st) bst8(code, ors(t), glane >> 1, (g * 64 + 2 * j) * 256);
          return;
        }
        bf16x4 hi, lo;
        split4(v, hi, lo);
        *reinterpret_cast<bf16x4*>(dhi + 256 * g + 8 * j) = hi;
        *reinterpret_cast<bf16x4*>(dlo + 256 * g + 8 * j) = lo;
        if (st) {
          if constexpr (G2) bst8(enc_dgates<GF>(v, hi), ors(t), glane >> 1, (g * 64 + 2 * j) * 256);
          else if constexpr (BLK) st_gate(pack_hl4(hi, lo), t, g, j);
          else st_gate(v, t, g, j);
        }
      };
      emit(pi, 0);
      emit(pf, 1);
      emit(pg, 2);
      emit(po, 3);
      if (!(DBG & 2)) load_step(tn, j);  // into the registers just consumed
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();

    const __bf16* bhi = &dgl[0][l31 * DROW + 8 * half];
    const __bf16* blo = &dgl[RF ? 0 : 1][l31 * DROW + 8 * half];
    const __bf16* bhx[2] = {bhi + 16 * kpar, bhi - 16 * kpar};
    f32x16 acc0;  // one accumulator: the other wave of the SIMD fills the dependent-issue gaps
    f32x16 acc1;  // (RF = 2 / 3: the lo terms)
    v8i b8;       // RF = 3: e4m3 of the chunk's d(gates) / 256 (|fp16| / 256 < 256: inside e4m3, which has no infinity)
#pragma unroll
    for (int ch = 0; ch < 16; ++ch) {
      const int s = ch & 1;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int ks = 4 * ch + q;
        // (DX: slot q holds k-step ks ^ kpar -- base bhx[q & 1] = bhi +- 16 kpar, the k-step in the immediate offset)
        const bf16x8 bh = *reinterpret_cast<const bf16x8*>((DX ? bhx[q & 1] : bhi) + 16 * ks);
        if constexpr (RF == 3) {
          typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
          typedef short s16x2 __attribute__((ext_vector_type(2)));
          const f16x8 b16 = __builtin_bit_cast(f16x8, bh), ah16 = __builtin_bit_cast(f16x8, wr[s][q]);
          const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah16, b16, ks == 0 ? zero : acc0, 0, 0, 0);
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            s16x2 c = {0, 0};
            c = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(c, f16x2{b16[4 * i], b16[4 * i + 1]}, 256.f, false);
            c = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(c, f16x2{b16[4 * i + 2], b16[4 * i + 3]}, 256.f, true);
            b8[2 * q + i] = __builtin_bit_cast(int, c);
          }
          if (q == 3) {
            const v8i a8 = {(int)wq[s][0][0], (int)wq[s][0][1], (int)wq[s][1][0], (int)wq[s][1][1],
                            (int)wq[s][2][0], (int)wq[s][2][1], (int)wq[s][3][0], (int)wq[s][3][1]};
            acc1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, ch == 0 ? zero : acc1, 0, 0, 0, wE[ch >> 1], 0, 127 + 8);
          }
        } else if constexpr (RF != 0) {
          typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
          typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
          const float sg = wS[ch >> 1];
          const u32x2 c8 = wq[s][q];
          const f16x2 a0 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(c8[0], sg, false);
          const f16x2 a1 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(c8[0], sg, true);
          const f16x2 a2 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(c8[1], sg, false);
          const f16x2 a3 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(c8[1], sg, true);
          const f16x8 al8 = {a0[0], a0[1], a1[0], a1[1], a2[0], a2[1], a3[0], a3[1]};
          const f16x8 b16 = __builtin_bit_cast(f16x8, bh), ah16 = __builtin_bit_cast(f16x8, wr[s][q]);
          const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah16, b16, ks == 0 ? zero : acc0, 0, 0, 0);
          if constexpr (DX) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al8, b16, acc0, 0, 0, 0);   // (one chain: the d(xn) tile
          else acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al8, b16, ks == 0 ? zero : acc1, 0, 0, 0);   // fills the gaps; 16 registers)
          if constexpr (DX) {
            // d(xn): the even slots hold this wave's k-steps (see kpar): same B fragment, this wave's W_ih^T fragments
            if (!(q & 1)) {
              const float sx = xS[ch >> 1];
              const u32x2 x8 = xq[s][q >> 1];
              const f16x2 e0 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(x8[0], sx, false);
              const f16x2 e1 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(x8[0], sx, true);
              const f16x2 e2 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(x8[1], sx, false);
              const f16x2 e3 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(x8[1], sx, true);
              const f16x8 xl8 = {e0[0], e0[1], e1[0], e1[1], e2[0], e2[1], e3[0], e3[1]};
              dxa = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, xr[s][q >> 1]), b16, ks == 0 ? zero : dxa, 0, 0, 0);
              dxa = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl8, b16, dxa, 0, 0, 0);
            }
          }
        } else {
          const bf16x8 bl = *reinterpret_cast<const bf16x8*>(blo + 16 * ks);
          if (ks == 0) {
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc0 = mfma32(wr[s][2 * q], bh, zero);
          } else {
            acc0 = mfma32(wr[s][2 * q], bh, acc0);
          }
          acc0 = mfma32(wr[s][2 * q + 1], bh, acc0);
          acc0 = mfma32(wr[s][2 * q], bl, acc0);
        }
      }
      const int cn = (ch + 2) & 15;  // wraps into the next step
      if (!(DBG & 4)) {
        wfill(s, cn, zo);
        xfill(s, cn, zo);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (RF != 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) dhr[i] = (DX ? acc0[i] : acc0[i] + acc1[i]) * (1.f / 256.f);   // the fp16 weights are 256 w (exact to undo)
    } else {
      dhr = acc0;
    }
    // DX: the two k-step parities' partial tiles meet in LDS.  EVERY wave runs the same instructions (a branch around the stores
    // made the compiler drain vmcnt to 0 at the top of the step -- the ring's first fragments and the stores' acknowledgements:
    // +5 us per step, measured): wave (mt, kpar) keeps the rows j = 2 kpar, 2 kpar + 1 of its partial (register 4 j + r = input
    // 32 mt + 8 j + 4 half + r), sends the other two to its partner w ^ 4, adds what the partner sent and stores two 16-byte
    // cells per lane; own + partner's = even + odd k-steps either way (one fp32 add: commutative, so the bits do not depend on
    // which wave adds)
    f32x4 keep[2];
    if constexpr (DX) {
      const bool hi_rows = kpar != 0;   // (uniform) selects as data, not as control flow
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        f32x4 send;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float lo_v = dxa[4 * i + r], hi_v = dxa[8 + 4 * i + r];
          send[r] = hi_rows ? lo_v : hi_v;
          keep[i][r] = hi_rows ? hi_v : lo_v;
        }
        xred[(w * 2 + i) * 64 + lane] = send;
      }
    }
    __syncthreads();
    if constexpr (DX) {
      // the accumulators hold 256 w x S d(gates): both powers of two leave here.  (xred is rewritten after the NEXT step's MFMA
      // loop: the barrier behind the cell phase lies in between)
      const float us = (1.f / 256.f) / dS;
      const unsigned trow = (unsigned)t * (unsigned)p.step_rows * 512u;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const f32x4 o = xred[((w ^ 4) * 2 + i) * 64 + lane];
        // (no register soffset: lstm_bf16_common.h bst -- the compiler pads the store-data hazard only for that form)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, (keep[i] + o) * us), xors, xoff + trow + 32 * i, 0, 0);
      }
    }
  }
}

#define WS_DBG_DISPATCH(KERNEL)                                                          \
  if ((a->mode & 255) == WS_LSTM_BF16X3_BLK && a->gfmt == WS_GATES_H2) {                 \
    hipLaunchKernelGGL((KERNEL<true, 0, WS_GATES_H2>), grid, block, 0, s, *a);           \
  } else if ((a->mode & 255) == WS_LSTM_BF16X3_BLK && a->gfmt == WS_GATES_H2S) {         \
    hipLaunchKernelGGL((KERNEL<true, 0, WS_GATES_H2S>), grid, block, 0, s, *a);          \
  } else if ((a->mode & 255) == WS_LSTM_BF16X3_BLK && a->gfmt == WS_GATES_H2F) {         \
    hipLaunchKernelGGL((KERNEL<true, 0, WS_GATES_H2F>), grid, block, 0, s, *a);          \
  } else if ((a->mode & 255) == WS_LSTM_BF16X3_BLK) {                                    \
    switch ((a->mode >> 8) & 7) {                                                        \
      case 0: hipLaunchKernelGGL((KERNEL<true, 0>), grid, block, 0, s, *a); break;       \
      case 1: hipLaunchKernelGGL((KERNEL<true, 1>), grid, block, 0, s, *a); break;       \
      case 2: hipLaunchKernelGGL((KERNEL<true, 2>), grid, block, 0, s, *a); break;       \
      case 4: hipLaunchKernelGGL((KERNEL<true, 4>), grid, block, 0, s, *a); break;       \
      default: hipLaunchKernelGGL((KERNEL<true, 7>), grid, block, 0, s, *a); break;      \
    }                                                                                    \
  } else {                                                                               \
    hipLaunchKernelGGL((KERNEL<false, 0>), grid, block, 0, s, *a);                       \
  }

int ws_launch_lstm_fwd_bf16(const ws_lstm_args* a, hipStream_t s) {
  dim3 grid((a->nseq + SQ - 1) / SQ, 2), block(512);
  WS_DBG_DISPATCH(lstm_fwd_bf16_kernel)
  return 0;
}

int ws_launch_lstm_bwd_bf16(const ws_lstm_args* a, hipStream_t s) {
  dim3 grid((a->nseq + SQ - 1) / SQ, 2), block(512);
  if (a->rfmt == 2 && a->dxn) {   // (lstm_check: rfmt 2, wxpack and a sequence map given)
    hipLaunchKernelGGL((lstm_bwd_bf16_kernel<true, 0, WS_GATES_H2F, 2, true>), grid, block, 0, s, *a);
    return 0;
  }
  if (a->rfmt == 2) {   // (lstm_check: WS_LSTM_BF16X3_BLK + WS_GATES_H2F only)
    hipLaunchKernelGGL((lstm_bwd_bf16_kernel<true, 0, WS_GATES_H2F, 2>), grid, block, 0, s, *a);
    return 0;
  }
  if (a->rfmt == 3) {   // (the same pack; the lo term on the FP8 matrix instruction)
    hipLaunchKernelGGL((lstm_bwd_bf16_kernel<true, 0, WS_GATES_H2F, 3>), grid, block, 0, s, *a);
    return 0;
  }
  WS_DBG_DISPATCH(lstm_bwd_bf16_kernel)
  return 0;
}
