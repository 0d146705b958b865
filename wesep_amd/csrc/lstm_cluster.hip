// Weight-stationary BLSTM forward recurrence across CLUSTERS of 8 workgroups (blocked layout BL).
//
// The per-step weight stream of lstm_bf16*.hip (1 MB from L2 per workgroup per step) is what bounds
// the time view of pBSRNN: 1024 sequences x 501 latency-bound steps.  Here the weights never move:
// a cluster of 8 co-resident workgroups owns 64 sequences of one direction; workgroup j keeps the
// W_hh rows of hidden units [32j, 32j+32) (all four gates: 128 rows x 256, split-bf16 = 128 KB) in the
// REGISTERS of its 8 waves (wave (uo, st): the 8 units 32j + 8uo.., sequence tile st; rows ordered
// [gate][unit] so a lane's D fragment holds i,f,g,o of 4 units of its sequence -> lane-local cell
// update).  What moves instead is h: each step every workgroup publishes its 8 KB slice of h_t
// (bf16 hi|lo) and gathers the other seven (MI355X_MICROARCH.md / cdna_hip_programming.md
// Guideline 16, recipe R1): payload with write-through (sc1) 16-byte stores, every storing wave
// drains vmcnt, __syncthreads, ONE relaxed agent-scope flag store; the consumer polls the seven flags
// relaxed from one wave, then reads the payload with sc1 loads (L1 bypassed: no acquire fence needed
// because the producer stored sc1).  Exchange buffers are double-buffered by step parity (a workgroup
// can publish step t+2 only after everyone published t+1, i.e. consumed t).  All spins are bounded;
// a timeout poisons the output with NaN and sets *status instead of hanging the GPU.
//
// Residency: the grid (8 workgroups per 64 sequences per direction) must be co-resident, i.e.
// <= 1 workgroup per CU of the device -- the launcher checks it against the CU count.
#include "lstm_bf16_common.h"

typedef __attribute__((address_space(1))) unsigned gu32;
#define SC1 16  // aux bit of raw buffer ops: sc1 = system-coherent level 1 (write-through / L1 bypass)
#define CL_SEQ 64
#define CL_SPIN_LIMIT (1u << 22)

__device__ __forceinline__ void split8r(const f32x4& a, const f32x4& b, bf16x8& hi, bf16x8& lo) {
  const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    hi[j] = (__bf16)v[j];
    lo[j] = (__bf16)(v[j] - (float)hi[j]);
  }
}

__global__ __launch_bounds__(512, 2) void lstm_fwd_cluster_kernel(const ws_lstm_cluster_args p) {
  __shared__ __attribute__((aligned(16))) __bf16 hl[2][CL_SEQ * HROW];  // [part][seq][k] 66 KB
  __shared__ int dead_s;
  const int ntile = p.nseq / 32, ncl_dir = ntile / 2, ncl = 2 * ncl_dir;
  const int c = blockIdx.x % ncl, j = blockIdx.x / ncl;  // member j of cluster c (same c -> same XCD when ncl % 8 == 0)
  const int d = c / ncl_dir, cc = c % ncl_dir;
  const int tid = threadIdx.x, lane = tid & 63, n = lane & 31, half = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int uo = w & 3, st = w >> 2;
  const int tile = 2 * cc + st;
  const int L = p.L;

  // ---- resident weights: rows m = lane&31 -> (gate = m>>3, unit 32j + 8uo + (m&7)) ----------------
  bf16x8 wh[16], wl[16];
  {
    const float* W = d ? p.whh_r : p.whh_f;
    const int row = (n >> 3) * LH + 32 * j + 8 * uo + (n & 7);
    const float* wr = W + (long long)row * LH + 8 * half;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
      split8r(*reinterpret_cast<const f32x4*>(wr + 16 * ks), *reinterpret_cast<const f32x4*>(wr + 16 * ks + 4),
              wh[ks], wl[ks]);
  }
  {
    uint32_t* z = reinterpret_cast<uint32_t*>(&hl[0][0]);
    for (int i = tid; i < 2 * CL_SEQ * HROW / 2; i += 512) z[i] = 0u;
    if (tid == 0) dead_s = 0;
  }
  // ---- activation I/O in BL: this lane's cell = (quad 8j + 2uo + half, slot n) of block (tile, t) --
  const int glane = ((d * 256 + 8 * j + 2 * uo + half) * 32 + n) * 16;  // bytes; + g*64*512
  const int clane = ((d * 64 + 8 * j + 2 * uo + half) * 32 + n) * 16;
  auto grs = [&](int t) { return mkrsrc(p.gates + (long long)(tile * L + t) * (SQ * 2 * LG), SQ * 2 * LG * 4); };
  auto crs = [&](float* b, int t) { return mkrsrc(b + (long long)(tile * L + t) * (SQ * 2 * LH), SQ * 2 * LH * 4); };
  // ---- exchange: X[cluster][parity][producer][seq 64][quad 8] x 16 B (hi x4 | lo x4) ---------------
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<char*>(p.xchg) + (long long)c * (2 * 8 * 8192), 0, 2 * 8 * 8192, 0x00020000);
  const int xpub = (j * 512 + (st * 32 + n) * 8 + 2 * uo + half) * 16;  // my published chunk (bytes, parity 0)
  const int xget = tid * 16;                                             // chunk tid of each producer
  const int gs = tid >> 3, gq = tid & 7;                                 // ... = (seq, quad) for the LDS fill
  gu32* flags = (gu32*)(p.flags) + c * 8;

  // x-projection prefetched TWO steps ahead (xg = this step, xn = next): the loads are issued behind the
  // payload gather of a step and would otherwise be needed one short MFMA phase later
  f32x4 xg[4], xn[4], c4 = {0.f, 0.f, 0.f, 0.f};
  {
    const int t0 = d == 0 ? 0 : L - 1, t1 = d == 0 ? min(1, L - 1) : max(L - 2, 0);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      xg[g] = bld(grs(t0), glane, g * 64 * 512);
      xn[g] = bld(grs(t1), glane, g * 64 * 512);
    }
  }
  __syncthreads();

  for (int step = 0; step < L; ++step) {
    const int t = d == 0 ? step : L - 1 - step;
    const int par = step & 1;
    // ---- G^T tile [4 gates x 8 units][32 seqs] = W slice * h^T --------------------------------------
    const __bf16* hhi = &hl[0][(st * 32 + n) * HROW + 8 * half];
    const __bf16* hlo = &hl[1][(st * 32 + n) * HROW + 8 * half];
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 16; ks += 2) {
      const bf16x8 bh0 = *reinterpret_cast<const bf16x8*>(hhi + 16 * ks);
      const bf16x8 bl0 = *reinterpret_cast<const bf16x8*>(hlo + 16 * ks);
      const bf16x8 bh1 = *reinterpret_cast<const bf16x8*>(hhi + 16 * ks + 16);
      const bf16x8 bl1 = *reinterpret_cast<const bf16x8*>(hlo + 16 * ks + 16);
      acc0 = mfma32(wh[ks], bh0, acc0);
      acc1 = mfma32(wh[ks + 1], bh1, acc1);
      acc0 = mfma32(wl[ks], bh0, acc0);
      acc1 = mfma32(wl[ks + 1], bh1, acc1);
      acc0 = mfma32(wh[ks], bl0, acc0);
      acc1 = mfma32(wh[ks + 1], bl1, acc1);
    }
    // ---- cell update (register 4q + r = gate q, unit r of this lane's 4) ----------------------------
    f32x4 vi, vf, vg, vo, vh;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float ig = fsig(acc0[r] + acc1[r] + xg[0][r]);
      const float fg = fsig(acc0[4 + r] + acc1[4 + r] + xg[1][r]);
      const float gg = ftanh(acc0[8 + r] + acc1[8 + r] + xg[2][r]);
      const float og = fsig(acc0[12 + r] + acc1[12 + r] + xg[3][r]);
      const float cn = fg * c4[r] + ig * gg;
      c4[r] = cn;
      vi[r] = ig;
      vf[r] = fg;
      vg[r] = gg;
      vo[r] = og;
      vh[r] = og * ftanh(cn);
    }
    const bool dead = dead_s != 0;
    if (dead) vh = f32x4{__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf("")};
    // ---- publish my slice of h_t: write-through store, drain, barrier, one flag store ----------------
    {
      bf16x4 hi, lo;
      split4(vh, hi, lo);
      struct { bf16x4 a, b; } pk = {hi, lo};
      if (!(p.dbg & 4)) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, pk), xrs, xpub, par * (8 * 8192), SC1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(flags + j, (unsigned)(step + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // ---- wait for the other seven (one wave polls, relaxed; bounded) ---------------------------------
    if (w == 0 && !dead && !(p.dbg & 1)) {
      unsigned spins = 0;
      while (true) {
        const unsigned v = lane < 8 ? __hip_atomic_load(flags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                    : 0xffffffffu;
        if (__all((int)(v >= (unsigned)(step + 1)))) break;
        if (++spins > CL_SPIN_LIMIT) {
          if (lane == 0) {
            dead_s = 1;
            if (p.status) __hip_atomic_store((gu32*)(p.status), 1u, __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_AGENT);
          }
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
    }
    __syncthreads();
    // ---- gather all eight slices (sc1 loads: L1 bypassed), THEN the HBM traffic of this step, so the
    //      payload's return is not queued behind it (VMEM returns in order per wave) ------------------
    u32x4 pv[8];
#pragma unroll
    for (int jj = 0; jj < 8; ++jj)
      pv[jj] = (p.dbg & 2) ? u32x4{0u, 0u, 0u, 0u}
                           : __builtin_amdgcn_raw_buffer_load_b128(xrs, xget + jj * 8192, par * (8 * 8192), SC1);
    {
      const int sn = min(step + 2, L - 1);
      const int tn = d == 0 ? sn : L - 1 - sn;
      bst(vi, grs(t), glane, 0);
      bst(vf, grs(t), glane, 64 * 512);
      bst(vg, grs(t), glane, 128 * 512);
      bst(vo, grs(t), glane, 192 * 512);
      bst(c4, crs(p.cbuf, t), clane, 0);
      bst(vh, crs(p.hcat, t), clane, 0);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        xg[g] = xn[g];
        xn[g] = bld(grs(tn), glane, g * 64 * 512);
      }
    }
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
      const int o = gs * HROW + 32 * jj + 4 * gq;
      *reinterpret_cast<uint2*>(&hl[0][o]) = uint2{pv[jj][0], pv[jj][1]};
      *reinterpret_cast<uint2*>(&hl[1][o]) = uint2{pv[jj][2], pv[jj][3]};
    }
    __syncthreads();
  }
}

extern "C" int ws_lstm_fwd_cluster(const ws_lstm_cluster_args* a, void* stream) {
  WS_REQUIRE(a && a->gates && a->cbuf && a->hcat && a->whh_f && a->whh_r && a->xchg && a->flags,
             "ws_lstm_fwd_cluster: null pointer");
  WS_REQUIRE(a->nseq > 0 && a->nseq % 64 == 0 && a->L > 0, "ws_lstm_fwd_cluster: nseq must be a multiple of 64");
  const int nwg = (a->nseq / 32) * 8;
  int dev = 0, cus = 0;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  WS_REQUIRE(nwg <= cus, "ws_lstm_fwd_cluster: %d workgroups must be co-resident but the device has %d CUs", nwg, cus);
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(a->flags, 0, (size_t)(a->nseq / 32) * 8 * sizeof(unsigned), s);
  WS_REQUIRE(e == hipSuccess, "ws_lstm_fwd_cluster: hipMemsetAsync failed");
  ws_prof_begin(WS_PROF_LSTM_FWD, s);
  hipLaunchKernelGGL(lstm_fwd_cluster_kernel, dim3(nwg), dim3(512), 0, s, *a);
  ws_prof_end(WS_PROF_LSTM_FWD, s);
  return ws_check_launch("ws_lstm_fwd_cluster");
}
