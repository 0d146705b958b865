// 16-sequence workgroups of the split-bf16 LSTM recurrence (blocked layout BL only), for views whose
// 32-sequence tiles would leave most of the chip idle: the time view of pBSRNN has R*K = 1024
// sequences = 64 workgroups of 32 for 256 CUs, and a step is a latency-bound chain
// (MFMA -> cell update -> HBM stores / next loads), so halving the sequences per workgroup halves the
// MFMA, VALU and HBM work on that chain and doubles the CUs in use.  Same structure as
// lstm_bf16.hip (transposed product, weights streamed from L2 through a register ring, h / dgates
// as bf16 hi/lo in LDS, branch-free step body) on v_mfma_f32_16x16x32_bf16:
//   A fragment (weights): lane = (row m = lane&15, k quarter kq = lane>>4), 8 consecutive k
//   B fragment (h / dgates): lane = (sequence n = lane&15, kq), 8 consecutive k
//   D (16 x 16): lane = (sequence n, mq = lane>>4), 4 consecutive rows 4mq..4mq+3 = one BL cell.
// Workgroup (tile, hb) owns slots [16hb, 16hb+16) of the blocks of `tile`; wave w owns hidden units
// [32w, 32w+32) = two 16-row tiles (tu) per gate.
#include "lstm_bf16_common.h"

#define S16 16

__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// ---------------------------------------------------------------------------------------------
// weight packing (16-byte units; `lane` = the MFMA lane that will load the unit)
//   fwd: unit ((((d*8 + w)*8 + ks)*8 + g*2 + tu)*2 + part)*64 + lane, element j
//          = part( W_hh[d][ g*256 + 32w + 16tu + (lane&15) ][ 32ks + 8(lane>>4) + j ] )
//   bwd: unit ((((d*8 + w)*32 + ks)*2 + tu)*2 + part)*64 + lane, element j
//          = part( W_hh[d][ 32ks + 8(lane>>4) + j ][ 32w + 16tu + (lane&15) ] )
// ---------------------------------------------------------------------------------------------
__global__ void lstm_pack_s16_kernel(const float* __restrict__ whh_f, const float* __restrict__ whh_r,
                                     __bf16* __restrict__ pf, __bf16* __restrict__ pb) {
  const int total = 2 * LG * LH;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    {
      int r = idx;
      const int j = r & 7; r >>= 3;
      const int lane = r & 63; r >>= 6;
      const int tile = r & 7; r >>= 3;  // g*2 + tu
      const int ks = r & 7; r >>= 3;
      const int w = r & 7; r >>= 3;
      const int d = r;
      const float* W = d ? whh_r : whh_f;
      const int row = (tile >> 1) * 256 + 32 * w + 16 * (tile & 1) + (lane & 15);
      const int k = 32 * ks + 8 * (lane >> 4) + j;
      const float v = W[row * LH + k];
      const __bf16 hi = (__bf16)v;
      const long long unit = ((((long long)(d * 8 + w) * 8 + ks) * 8 + tile) * 2) * 64 + lane;
      pf[unit * 8 + j] = hi;
      pf[(unit + 64) * 8 + j] = (__bf16)(v - (float)hi);
    }
    {
      int r = idx;
      const int j = r & 7; r >>= 3;
      const int lane = r & 63; r >>= 6;
      const int tu = r & 1; r >>= 1;
      const int ks = r & 31; r >>= 5;
      const int w = r & 7; r >>= 3;
      const int d = r;
      const float* W = d ? whh_r : whh_f;
      const int row = 32 * ks + 8 * (lane >> 4) + j;  // gate column = contraction index
      const int u = 32 * w + 16 * tu + (lane & 15);
      const float v = W[row * LH + u];
      const __bf16 hi = (__bf16)v;
      const long long unit = ((((long long)(d * 8 + w) * 32 + ks) * 2 + tu) * 2) * 64 + lane;
      pb[unit * 8 + j] = hi;
      pb[(unit + 64) * 8 + j] = (__bf16)(v - (float)hi);
    }
  }
}

int ws_launch_lstm_pack_s16(const float* whh_f, const float* whh_r, float* pack_fwd, float* pack_bwd,
                            hipStream_t s) {
  hipLaunchKernelGGL(lstm_pack_s16_kernel, dim3(512), dim3(256), 0, s, whh_f, whh_r,
                     reinterpret_cast<__bf16*>(pack_fwd), reinterpret_cast<__bf16*>(pack_bwd));
  return 0;
}

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
template <int GF>  // WS_GATES_* (lstm_bf16.hip)
__global__ __launch_bounds__(512, 2) void lstm_fwd_s16_kernel(const ws_lstm_args p) {
  __shared__ __attribute__((aligned(16))) __bf16 hl[2][2][S16 * HROW];  // [buf][part][seq][k] 33 KB
  __shared__ __attribute__((aligned(16))) float cl[S16 * (LH + 4)];     // cell state [seq][unit]
  if (p.run_if && *p.run_if == 0u) return;  // predicated fall-back launch (wesep_hip.h): uniform
  const int d = blockIdx.y;
  const int tile = blockIdx.x >> 1, hb = blockIdx.x & 1;
  const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, mq = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int L = p.L;
  {
    uint32_t* z = reinterpret_cast<uint32_t*>(&hl[0][0][0]);
    for (int i = tid; i < 2 * S16 * HROW / 2; i += 512) z[i] = 0u;
  }
  // BL cells of this lane: quad 8w + 4tu + mq of (direction, gate), slot 16hb + n
  const int glane = ((d * 256 + 8 * w + mq) * 32 + 16 * hb + n) * 16;  // bytes; + (g*64 + 4tu)*512
  const int clane = ((d * 64 + 8 * w + mq) * 32 + 16 * hb + n) * 16;   // bytes; + 4tu*512
  const float* gsrc = GF ? p.gates_in : p.gates;
  auto grs = [&](int t) { return mkrsrc(gsrc + (long long)(tile * L + t) * (SQ * 2 * LG), SQ * 2 * LG * 4); };
  auto hrs = [&](int t) { return mkrsrc(p.gates + (long long)(tile * L + t) * (SQ * LG), SQ * 2 * LG * 2); };  // BLH
  auto crs = [&](float* b, int t) { return mkrsrc(b + (long long)(tile * L + t) * (SQ * 2 * LH), SQ * 2 * LH * 4); };
  auto ld_gate = [&](int t, int g, int tu) -> f32x4 { return bld(grs(t), glane, (g * 64 + 4 * tu) * 512); };
  auto st_gate = [&](const f32x4& v, int t, int g, int tu) {
    if constexpr (GF != 0) bst8(g == 2 ? enc_u16x4<true>(v) : enc_u16x4<false>(v), hrs(t), glane >> 1, (g * 64 + 4 * tu) * 256);
    else bst(v, grs(t), glane, (g * 64 + 4 * tu) * 512);
  };
  auto st_ch = [&](const f32x4& v, float* b, int t, int tu) { bst(v, crs(b, t), clane, 4 * tu * 512); };

  const int ubase = 32 * w + 4 * mq;  // unit of (tu, r): ubase + 16tu + r
  float* cme = &cl[n * (LH + 4) + ubase];
#pragma unroll
  for (int tu = 0; tu < 2; ++tu) *reinterpret_cast<f32x4*>(cme + 16 * tu) = f32x4{0.f, 0.f, 0.f, 0.f};

  // weight stream: per k-step (32 k) 16 fragments (8 tiles x {hi, lo}), 16 KB per wave
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.wpack) + (long long)(d * 8 + w) * (8 * 16 * 64 * 4), 0, 8 * 16 * 1024, 0x00020000);
  const int wlane = lane * 16;
  bf16x8 wr[2][16];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int f = 0; f < 16; ++f) wr[s][f] = wload(wrs, wlane + (f & 3) * 1024, s * 16384 + (f >> 2) * 4096);

  f32x4 xg[4][2];  // [gate][tu]
  {
    const int t0 = d == 0 ? 0 : L - 1;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int tu = 0; tu < 2; ++tu) xg[g][tu] = ld_gate(t0, g, tu);
  }
  __syncthreads();

  for (int step = 0; step < L; ++step) {
    const int t = d == 0 ? step : L - 1 - step;
    const int cur = step & 1;
    int zo = 0;
    asm volatile("" : "+s"(zo));
    const __bf16* hhi = &hl[cur][0][n * HROW + 8 * mq];
    const __bf16* hlo = &hl[cur][1][n * HROW + 8 * mq];
    f32x4 acc[4][2];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int s = ks & 1;
      const bf16x8 bh = *reinterpret_cast<const bf16x8*>(hhi + 32 * ks);
      const bf16x8 bl = *reinterpret_cast<const bf16x8*>(hlo + 32 * ks);
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int tu = 0; tu < 2; ++tu) {
          const int f = (g * 2 + tu) * 2;
          if (ks == 0) {
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            acc[g][tu] = mfma16(wr[s][f], bh, zero);
          } else {
            acc[g][tu] = mfma16(wr[s][f], bh, acc[g][tu]);
          }
        }
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int tu = 0; tu < 2; ++tu) acc[g][tu] = mfma16(wr[s][(g * 2 + tu) * 2 + 1], bh, acc[g][tu]);
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int tu = 0; tu < 2; ++tu) acc[g][tu] = mfma16(wr[s][(g * 2 + tu) * 2], bl, acc[g][tu]);
      const int kn = (ks + 2) & 7;  // wraps into the next step: the stream never drains
#pragma unroll
      for (int f = 0; f < 16; ++f)
        wr[s][f] = wload(wrs, wlane + (f & 3) * 1024, zo + kn * 16384 + (f >> 2) * 4096);
      __builtin_amdgcn_sched_barrier(0);
    }

    const int sn = min(step + 1, L - 1);
    const int tn = d == 0 ? sn : L - 1 - sn;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int tu = 0; tu < 2; ++tu) acc[g][tu] += xg[g][tu];
#pragma unroll
    for (int tu = 0; tu < 2; ++tu)
#pragma unroll
      for (int g = 0; g < 4; ++g) xg[g][tu] = ld_gate(tn, g, tu);
    __builtin_amdgcn_sched_barrier(0);
    __bf16* nhi = &hl[cur ^ 1][0][n * HROW + ubase];
    __bf16* nlo = &hl[cur ^ 1][1][n * HROW + ubase];
#pragma unroll
    for (int tu = 0; tu < 2; ++tu) {
      f32x4 vi, vf, vg, vo, vc, vh;
      const f32x4 cold = *reinterpret_cast<const f32x4*>(cme + 16 * tu);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float ig = fsig(acc[0][tu][r]);
        const float fg = fsig(acc[1][tu][r]);
        const float gg = ftanh(acc[2][tu][r]);
        const float og = fsig(acc[3][tu][r]);
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
      *reinterpret_cast<bf16x4*>(nhi + 16 * tu) = h_hi;
      *reinterpret_cast<bf16x4*>(nlo + 16 * tu) = h_lo;
      *reinterpret_cast<f32x4*>(cme + 16 * tu) = vc;
      st_gate(vi, t, 0, tu);
      st_gate(vf, t, 1, tu);
      st_gate(vg, t, 2, tu);
      st_gate(vo, t, 3, tu);
      st_ch(vc, p.cbuf, t, tu);
      st_ch(pack_hl4(h_hi, h_lo), p.hcat, t, tu);  // BLS
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// backward (BPTT): dh_{t-1}^T[unit][seq] = W_hh^T[unit][gate col] * dgates_t^T[gate col][seq]
// ---------------------------------------------------------------------------------------------

#ifndef S16_DBG
#define S16_DBG 0  // probe builds: 1 no next-step loads, 2 no gate-gradient stores, 4 no weight reloads
#endif
template <int GF>
__global__ __launch_bounds__(512, 2) void lstm_bwd_s16_kernel(const ws_lstm_args p) {
  __shared__ __attribute__((aligned(16))) __bf16 dgl[2][S16 * DROW];  // [part][seq][gate col] 66 KB
  if (p.run_if && *p.run_if == 0u) return;  // predicated fall-back launch (wesep_hip.h): uniform
  const int d = blockIdx.y;
  const int tile = blockIdx.x >> 1, hb = blockIdx.x & 1;
  const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, mq = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int L = p.L;
  const int glane = ((d * 256 + 8 * w + mq) * 32 + 16 * hb + n) * 16;
  const int clane = ((d * 64 + 8 * w + mq) * 32 + 16 * hb + n) * 16;
  float* gdst = GF == WS_GATES_H2S ? p.dgates : p.gates;
  auto grs = [&](int t) { return mkrsrc(gdst + (long long)(tile * L + t) * (SQ * 2 * LG), SQ * 2 * LG * 4); };
  auto hrs = [&](int t) { return mkrsrc(p.gates + (long long)(tile * L + t) * (SQ * LG), SQ * 2 * LG * 2); };  // BLH
  constexpr bool G2 = GF == WS_GATES_H2 || GF == WS_GATES_H2F;  // 2-byte d(gates): bf16, or fp16 scaled by dS
  float* hdst = (G2 && p.dgates) ? p.dgates : p.gates;  // in place, or to their own BLH buffer
  auto ors = [&](int t) { return mkrsrc(hdst + (long long)(tile * L + t) * (SQ * LG), SQ * 2 * LG * 2); };
  const float dS = GF == WS_GATES_H2F ? ws_dgates_scale(*p.amax) : 1.f;
  auto crs = [&](const float* b, int t) { return mkrsrc(b + (long long)(tile * L + t) * (SQ * 2 * LH), SQ * 2 * LH * 4); };
  typedef typename gate_cell<GF>::type gcell;
  auto ld_gate = [&](int t, int g, int tu) -> gcell {
    if constexpr (GF != 0) return bld8(hrs(t), glane >> 1, (g * 64 + 4 * tu) * 256);
    else return bld(grs(t), glane, (g * 64 + 4 * tu) * 512);
  };
  auto st_gate = [&](const f32x4& v, int t, int g, int tu) { bst(v, grs(t), glane, (g * 64 + 4 * tu) * 512); };
  auto ld_ch = [&](const float* b, int t, int tu) -> f32x4 { return bld(crs(b, t), clane, 4 * tu * 512); };
  const int ubase = 32 * w + 4 * mq;

  // weight stream: per k-step (32 gate columns) 4 fragments (2 tiles x {hi, lo}); ring slots hold 4 k-steps
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.wpack) + (long long)(d * 8 + w) * (32 * 4 * 64 * 4), 0, 32 * 4 * 1024, 0x00020000);
  const int wlane = lane * 16;
  bf16x8 wr[2][16];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int f = 0; f < 16; ++f) wr[s][f] = wload(wrs, wlane + (f & 3) * 1024, s * 16384 + (f >> 2) * 4096);

  gcell n_i[2], n_f[2], n_g[2], n_o[2];
  f32x4 n_dh[2], n_cp[2], c_cur[2], dc[2], dhr[2];
  const f32x4 zero4v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int tu = 0; tu < 2; ++tu) dc[tu] = dhr[tu] = zero4v;
  auto load_step = [&](int t, int tu) {
    n_i[tu] = ld_gate(t, 0, tu);
    n_f[tu] = ld_gate(t, 1, tu);
    n_g[tu] = ld_gate(t, 2, tu);
    n_o[tu] = ld_gate(t, 3, tu);
    n_dh[tu] = ld_ch(p.dhcat, t, tu);
    const int tp = d == 0 ? max(t - 1, 0) : min(t + 1, L - 1);  // clamped; masked at its use
    n_cp[tu] = ld_ch(p.cbuf, tp, tu);
  };
  {
    const int t0 = d == 0 ? L - 1 : 0;
#pragma unroll
    for (int tu = 0; tu < 2; ++tu) {
      load_step(t0, tu);
      c_cur[tu] = ld_ch(p.cbuf, t0, tu);
    }
  }

  for (int step = 0; step < L; ++step) {
    const int t = d == 0 ? L - 1 - step : step;
    const int sn = min(step + 1, L - 1);
    const int tn = d == 0 ? L - 1 - sn : sn;
    const bool has_prev = d == 0 ? (t > 0) : (t < L - 1);
    int zo = 0;
    asm volatile("" : "+s"(zo));
    __bf16* dhi = &dgl[0][n * DROW + ubase];
    __bf16* dlo = &dgl[1][n * DROW + ubase];
#pragma unroll
    for (int tu = 0; tu < 2; ++tu) {
      f32x4 pi, pf, pg, po;
      const f32x4 vi = gate_val<false>(n_i[tu]), vf = gate_val<false>(n_f[tu]), vg = gate_val<true>(n_g[tu]),
                  vo = gate_val<false>(n_o[tu]);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float ig = vi[r], fg = vf[r], gg = vg[r], og = vo[r];
        const float dhv = dh_in<GF>(n_dh[tu][r], dhr[tu][r], dS);
        const float tc = ftanh(c_cur[tu][r]);
        const float dov = dhv * tc;
        const float dcv = dc[tu][r] + dhv * og * (1.f - tc * tc);
        dc[tu][r] = dcv * fg;
        pi[r] = dcv * gg * ig * (1.f - ig);
        pf[r] = dcv * (has_prev ? n_cp[tu][r] : 0.f) * fg * (1.f - fg);
        pg[r] = dcv * ig * (1.f - gg * gg);
        po[r] = dov * og * (1.f - og);
      }
      c_cur[tu] = n_cp[tu];
      // d(gates): LDS image (B operand) + HBM as the same split pair (BLS)
      auto emit = [&](const f32x4& v, int g) {
        bf16x4 hi, lo;
        split4(v, hi, lo);
        *reinterpret_cast<bf16x4*>(dhi + 256 * g + 16 * tu) = hi;
        *reinterpret_cast<bf16x4*>(dlo + 256 * g + 16 * tu) = lo;
        if constexpr (G2) bst8(enc_dgates<GF>(v, hi), ors(t), glane >> 1, (g * 64 + 4 * tu) * 256);
        else if (!(S16_DBG & 2)) st_gate(pack_hl4(hi, lo), t, g, tu);
      };
      emit(pi, 0);
      emit(pf, 1);
      emit(pg, 2);
      emit(po, 3);
      if (!(S16_DBG & 1)) load_step(tn, tu);  // into the registers just consumed
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();

    const __bf16* bhi = &dgl[0][n * DROW + 8 * mq];
    const __bf16* blo = &dgl[1][n * DROW + 8 * mq];
    f32x4 acc[2];
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) {
      const int s = ch & 1;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int ks = 4 * ch + q;
        const bf16x8 bh = *reinterpret_cast<const bf16x8*>(bhi + 32 * ks);
        const bf16x8 bl = *reinterpret_cast<const bf16x8*>(blo + 32 * ks);
#pragma unroll
        for (int tu = 0; tu < 2; ++tu) {
          const int f = (q * 2 + tu) * 2;
          if (ks == 0)
            acc[tu] = mfma16(wr[s][f], bh, zero4v);
          else
            acc[tu] = mfma16(wr[s][f], bh, acc[tu]);
        }
#pragma unroll
        for (int tu = 0; tu < 2; ++tu) acc[tu] = mfma16(wr[s][(q * 2 + tu) * 2 + 1], bh, acc[tu]);
#pragma unroll
        for (int tu = 0; tu < 2; ++tu) acc[tu] = mfma16(wr[s][(q * 2 + tu) * 2], bl, acc[tu]);
      }
      const int cn = (ch + 2) & 7;  // wraps into the next step
      if (!(S16_DBG & 4)) {
#pragma unroll
      for (int f = 0; f < 16; ++f)
        wr[s][f] = wload(wrs, wlane + (f & 3) * 1024, zo + cn * 16384 + (f >> 2) * 4096);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    dhr[0] = acc[0];
    dhr[1] = acc[1];
    __syncthreads();
  }
}

int ws_launch_lstm_fwd_s16(const ws_lstm_args* a, hipStream_t s) {
  dim3 grid(2 * ((a->nseq + SQ - 1) / SQ), 2), block(512);
  if (a->gfmt) hipLaunchKernelGGL(lstm_fwd_s16_kernel<WS_GATES_H2>, grid, block, 0, s, *a);
  else hipLaunchKernelGGL(lstm_fwd_s16_kernel<0>, grid, block, 0, s, *a);
  return 0;
}

int ws_launch_lstm_bwd_s16(const ws_lstm_args* a, hipStream_t s) {
  dim3 grid(2 * ((a->nseq + SQ - 1) / SQ), 2), block(512);
  if (a->gfmt == WS_GATES_H2) hipLaunchKernelGGL(lstm_bwd_s16_kernel<WS_GATES_H2>, grid, block, 0, s, *a);
  else if (a->gfmt == WS_GATES_H2S) hipLaunchKernelGGL(lstm_bwd_s16_kernel<WS_GATES_H2S>, grid, block, 0, s, *a);
  else if (a->gfmt == WS_GATES_H2F) hipLaunchKernelGGL(lstm_bwd_s16_kernel<WS_GATES_H2F>, grid, block, 0, s, *a);
  else hipLaunchKernelGGL(lstm_bwd_s16_kernel<0>, grid, block, 0, s, *a);
  return 0;
}
