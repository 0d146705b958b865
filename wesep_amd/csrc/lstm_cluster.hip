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
// a timeout poisons the output with NaN and sets the launch's timeout word (+ *status) instead of hanging the
// GPU; the callers enqueue the streaming kernels predicated on that word behind every launch (wesep_hip.h).
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

// cold path of a bounded wait: poison from here on, set this launch's timeout word (flags[ncl * 8], zeroed with the
// flags) and the caller's sticky status word
__device__ __noinline__ void cluster_timed_out(unsigned* tword, unsigned* status, int* dead_s) {
  *dead_s = 1;
  __hip_atomic_store((gu32*)tword, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (status) __hip_atomic_store((gu32*)status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Cluster -> workgroup mapping.  Workgroups are dispatched round-robin over the 8 XCDs (blockIdx % 8), each with its
// own L2.  All 8 members of a cluster are placed on ONE XCD for every cluster count: XCD x hosts the clusters
// c = x, x + 8, ..., member j of cluster c is block ((j * cpx + c / 8) * 8 + x) with cpx = ceil(ncl / 8) clusters per
// XCD; the grid is 64 * cpx blocks and blocks whose cluster index is >= ncl exit at once.  (Round 1 used
// c = blockIdx % ncl, which equals this mapping when ncl % 8 == 0 -- training at R = 32 -- but spread the members of
// small launches -- inference, 2 rows: ncl = 2 -- over XCDs, where the sc1 payload / relaxed flag hand-off has no
// ordering guarantee between two L2s: concurrent engines produced rare one-LSB differences, round 2.)
__device__ __forceinline__ bool cluster_of_block(int ncl, int legacy, int& c, int& j) {
  if (legacy) {
    c = blockIdx.x % ncl, j = blockIdx.x / ncl;
    return j < 8;
  }
  const int cpx = (ncl + 7) >> 3, x = blockIdx.x & 7, q = blockIdx.x >> 3;
  c = (q % cpx) * 8 + x;
  j = q / cpx;
  return c < ncl;
}
__host__ inline int cluster_grid(int ncl, int legacy) { return legacy ? ncl * 8 : 64 * ((ncl + 7) / 8); }

// GF: WS_GATES_* (lstm_bf16_common.h); != 0: pre-activations from p.gates_in (fp32 BL), activated gates to p.gates as
// unorm16 (BLH): the stage of the four gates shrinks from 32 to 16 KB, their HBM stores from 16 to 8 bytes per cell
template <bool FORCE, int GF>
__global__ __launch_bounds__(512, 2) void lstm_fwd_cluster_kernel(const ws_lstm_cluster_args p) {
  // VMEM operations of one wave complete in order, so an exchange wave must never have HBM traffic in
  // its queue (measured: +3.2 us per step when it does).  All 8 waves run the MFMAs and the cell
  // update of their own tile; then the traffic is split by role through LDS staging:
  //   X-waves (0..3): publish (from PUB), drain, flag, poll, gather -> h image
  //   M-waves (4..7): HBM stores of gates / c / h (from OUT), x-projection prefetch (-> XIN)
  __shared__ __attribute__((aligned(16))) __bf16 hl[2][CL_SEQ * HROW];  // [part][seq][k] 66 KB
  __shared__ __attribute__((aligned(16))) f32x4 xin[4][512];             // x-projection of the step, 32 KB
  __shared__ __attribute__((aligned(16))) f32x4 outl[GF ? 4 : 6][512];   // i, f, g, o, c, h of the step, 48 KB (GF: i|f, g|o
                                                                         // as unorm16 pairs, c, h: 32 KB)
  __shared__ __attribute__((aligned(16))) u32x4 publ[512];               // h chunks (bf16 hi x4 | lo x4), 8 KB
  __shared__ int dead_s;
  const int ntile = p.nseq / 32, ncl_dir = ntile / 2, ncl = 2 * ncl_dir;
  int c, j;  // member j of cluster c
  if (!cluster_of_block(ncl, p.dbg & 16, c, j)) return;
  const int d = c / ncl_dir, cc = c % ncl_dir;
  const int tid = threadIdx.x, lane = tid & 63, n = lane & 31, half = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool xrole = w < 4;
  const int uo = w & 3, st = w >> 2;
  const int L = p.L;
  const long long gblk = (long long)SQ * 2 * LG, cblk = (long long)SQ * 2 * LH;  // floats per block

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
  // ---- HBM side (M-waves): thread mt serves the cells of threads mt (tile 2cc) and mt + 256 (tile 2cc+1):
  //      same BL cell position, blocks one tile apart
  const int mt = tid & 255;
  const int m_uo = (mt >> 6) & 3, m_n = mt & 31, m_half = (mt >> 5) & 1;
  const int gvo = ((d * 256 + 8 * j + 2 * m_uo + m_half) * 32 + m_n) * 16;  // bytes; + g*64*512; + e * tile stride
  const int cvo = ((d * 64 + 8 * j + 2 * m_uo + m_half) * 32 + m_n) * 16;
  const int gts = (int)(L * gblk * 4), cts = (int)(L * cblk * 4);             // bytes between the two tiles
  const float* gsrc = GF ? p.gates_in : p.gates;
  auto grs = [&](int t) { return mkrsrc(gsrc + ((long long)2 * cc * L + t) * gblk, 0x7fffffffu); };
  auto hrs = [&](int t) { return mkrsrc(p.gates + ((long long)2 * cc * L + t) * (gblk / 2), 0x7fffffffu); };  // BLH
  auto crs = [&](float* b, int t) { return mkrsrc(b + ((long long)2 * cc * L + t) * cblk, 0x7fffffffu); };
  constexpr int OC = GF ? 2 : 4, OH = OC + 1;  // stage rows of c and h
  // ---- exchange side (X-waves): X[cluster][parity][producer][chunk 512] x 16 B ----------------------
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<char*>(p.xchg) + (long long)c * (2 * 8 * 8192), 0, 2 * 8 * 8192, 0x00020000);
  const int mychunk = (st * 32 + n) * 8 + 2 * uo + half;  // chunk of this thread's h in a producer's slice
  gu32* flags = (gu32*)(p.flags) + c * 8;

  f32x4 c4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 xpre[2][4];  // M-waves: x-projection of the NEXT step for their two cells
  {
    const int t0 = d == 0 ? 0 : L - 1, t1 = d == 0 ? min(1, L - 1) : max(L - 2, 0);
    if (!xrole) {
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          xin[g][mt + 256 * e] = bld(grs(t0), gvo + e * gts, g * 64 * 512);
          xpre[e][g] = bld(grs(t1), gvo + e * gts, g * 64 * 512);
        }
    }
  }
  __syncthreads();

  // HBM traffic is issued at the top of the NEXT step, under the MFMAs, not under the exchange: a CU's
  // hand-off latency doubles when its memory queue is streaming.  The outputs wait in the LDS stage
  // (barrier 0 below keeps the next cell update from overwriting it first).
  auto hbm_io = [&](int tprev, int stepn) {  // stores of step tprev, x-projection prefetch of step `stepn`
    const int sn = min(stepn, L - 1);
    const int tn = d == 0 ? sn : L - 1 - sn;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int ct = mt + 256 * e;
      if constexpr (GF != 0) {
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2) {
          const u32x4 pr = __builtin_bit_cast(u32x4, outl[g2][ct]);
          bst8(u32x2{pr[0], pr[1]}, hrs(tprev), (gvo + e * gts) >> 1, (2 * g2) * 64 * 256);
          bst8(u32x2{pr[2], pr[3]}, hrs(tprev), (gvo + e * gts) >> 1, (2 * g2 + 1) * 64 * 256);
        }
      } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) bst(outl[g][ct], grs(tprev), gvo + e * gts, g * 64 * 512);
      }
      bst(outl[OC][ct], crs(p.cbuf, tprev), cvo + e * cts, 0);
      bst(outl[OH][ct], crs(p.hcat, tprev), cvo + e * cts, 0);
#pragma unroll
      for (int g = 0; g < 4; ++g) xpre[e][g] = bld(grs(tn), gvo + e * gts, g * 64 * 512);
    }
  };
  for (int step = 0; step < L; ++step) {
    const int t = d == 0 ? step : L - 1 - step;
    const int par = step & 1;
    if (!xrole && step > 0) hbm_io(d == 0 ? step - 1 : L - step, step + 1);
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
    __syncthreads();  // 0: the previous step's outputs have left the LDS stage
    // ---- cell update (register 4q + r = gate q, unit r of this lane's 4); results to the LDS stages --
    {
      const f32x4 x0 = xin[0][tid], x1 = xin[1][tid], x2 = xin[2][tid], x3 = xin[3][tid];
      f32x4 vi, vf, vg, vo, vh;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float ig = fsig(acc0[r] + acc1[r] + x0[r]);
        const float fg = fsig(acc0[4 + r] + acc1[4 + r] + x1[r]);
        const float gg = ftanh(acc0[8 + r] + acc1[8 + r] + x2[r]);
        const float og = fsig(acc0[12 + r] + acc1[12 + r] + x3[r]);
        const float cn = fg * c4[r] + ig * gg;
        c4[r] = cn;
        vi[r] = ig;
        vf[r] = fg;
        vg[r] = gg;
        vo[r] = og;
        vh[r] = og * ftanh(cn);
      }
      if (dead_s != 0) vh = f32x4{__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf("")};
      if constexpr (GF != 0) {
        const u32x2 ei = enc_u16x4<false>(vi), ef = enc_u16x4<false>(vf), eg = enc_u16x4<true>(vg), eo = enc_u16x4<false>(vo);
        outl[0][tid] = __builtin_bit_cast(f32x4, u32x4{ei[0], ei[1], ef[0], ef[1]});
        outl[1][tid] = __builtin_bit_cast(f32x4, u32x4{eg[0], eg[1], eo[0], eo[1]});
      } else {
        outl[0][tid] = vi;
        outl[1][tid] = vf;
        outl[2][tid] = vg;
        outl[3][tid] = vo;
      }
      outl[OC][tid] = c4;
      bf16x4 hi, lo;
      split4(vh, hi, lo);
      outl[OH][tid] = pack_hl4(hi, lo);  // hcat in HBM carries the split pair (BLS); NaN poison survives in hi
      struct { bf16x4 a, b; } pk = {hi, lo};
      publ[mychunk] = __builtin_bit_cast(u32x4, pk);
    }
    __syncthreads();  // 1: stages complete, xin consumed, h image no longer read
    if (xrole) {
      // publish both halves of the slice: write-through stores, then drain
#pragma unroll
      for (int e = 0; e < 2; ++e)
        __builtin_amdgcn_raw_buffer_store_b128(publ[mt + 256 * e], xrs,  // (no register soffset: lstm_bf16_common.h bst)
                                               (j * 512 + mt + 256 * e) * 16 + par * (8 * 8192), 0, SC1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      // next step's x-projection into place (LDS only; the HBM traffic waits for the next MFMA phase)
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int g = 0; g < 4; ++g) xin[g][mt + 256 * e] = xpre[e][g];
    }
    __syncthreads();  // 2: every publishing wave has drained; xin holds the next step
    if (tid == 0) __hip_atomic_store(flags + j, (unsigned)(step + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (w == 0 && dead_s == 0 && !(p.dbg & 1)) {
      unsigned spins = 0;
      while (true) {
        const unsigned v = lane < 8 ? __hip_atomic_load(flags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                    : 0xffffffffu;
        const bool force = FORCE && step == 2 && c == 0 && j == 0;  // test instantiation: a timeout on demand
        if (!force && __all((int)(v >= (unsigned)(step + 1)))) break;
        if (force || ++spins > CL_SPIN_LIMIT) {
          if (lane == 0) cluster_timed_out(p.flags + ncl * 8, p.status, &dead_s);
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
    }
    __syncthreads();  // 3: all eight slices of h_t are visible
    if (xrole) {
      // gather (sc1 loads: L1 bypassed) and rebuild the h image: chunk -> (seq, unit quad) of producer jj
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int ch = mt + 256 * e, gs = ch >> 3, gq = ch & 7;
        u32x4 pv[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj)
          pv[jj] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (jj * 512 + ch) * 16, par * (8 * 8192), SC1);
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
          const int o = gs * HROW + 32 * jj + 4 * gq;
          *reinterpret_cast<uint2*>(&hl[0][o]) = uint2{pv[jj][0], pv[jj][1]};
          *reinterpret_cast<uint2*>(&hl[1][o]) = uint2{pv[jj][2], pv[jj][3]};
        }
      }
    }
    __syncthreads();  // 4: h image of the next step complete
  }
  if (!xrole) hbm_io(d == 0 ? L - 1 : 0, L);  // the last step's stores
}

extern "C" int ws_lstm_fwd_cluster(const ws_lstm_cluster_args* a, void* stream) {
  WS_REQUIRE(a && a->gates && a->cbuf && a->hcat && a->whh_f && a->whh_r && a->xchg && a->flags,
             "ws_lstm_fwd_cluster: null pointer");
  WS_REQUIRE(a->nseq > 0 && a->nseq % 64 == 0 && a->L > 0, "ws_lstm_fwd_cluster: nseq must be a multiple of 64");
  WS_REQUIRE(a->gfmt >= WS_GATES_F32 && a->gfmt <= WS_GATES_H2F && (a->gfmt == 0 || a->gates_in),
             "ws_lstm_fwd_cluster: gfmt %d needs gates_in", a->gfmt);
  const int nwg = (a->nseq / 32) * 8;
  int dev = 0, cus = 0;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  WS_REQUIRE(nwg <= cus, "ws_lstm_fwd_cluster: %d workgroups must be co-resident but the device has %d CUs", nwg, cus);
  const int grid = cluster_grid(a->nseq / 32, a->dbg & 16);
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(a->flags, 0, ((size_t)(a->nseq / 32) * 8 + 8) * sizeof(unsigned), s);
  WS_REQUIRE(e == hipSuccess, "ws_lstm_fwd_cluster: hipMemsetAsync failed");
  ws_prof_begin(WS_PROF_LSTM_FWD, s);
  if (a->gfmt && (a->dbg & 8))
    hipLaunchKernelGGL((lstm_fwd_cluster_kernel<true, WS_GATES_H2>), dim3(grid), dim3(512), 0, s, *a);
  else if (a->gfmt)
    hipLaunchKernelGGL((lstm_fwd_cluster_kernel<false, WS_GATES_H2>), dim3(grid), dim3(512), 0, s, *a);
  else if (a->dbg & 8)
    hipLaunchKernelGGL((lstm_fwd_cluster_kernel<true, 0>), dim3(grid), dim3(512), 0, s, *a);
  else
    hipLaunchKernelGGL((lstm_fwd_cluster_kernel<false, 0>), dim3(grid), dim3(512), 0, s, *a);
  ws_prof_end(WS_PROF_LSTM_FWD, s);
  return ws_check_launch("ws_lstm_fwd_cluster");
}

// =============================================================================================
// Backward (BPTT) over the same clusters.
//   dh_{t-1}[seq][u'] = sum_k dgates_t[seq][k] * W_hh[k][u'],  k over all 1024 gate columns.
// Workgroup j owns the gate columns of ITS 32 units (128 rows of W_hh, resident in registers) and
// computes, from its own dgates only, a PARTIAL dh for all 256 units; the eight partials are
// exchanged as a reduce-scatter (partial rows of units 32i.. go to workgroup i: 8 x 8 KB fp32 out,
// 8 x 8 KB in, per step) and summed in fixed order (deterministic).
// Waves are specialised so that the latency-critical exchange never queues behind HBM traffic
// (VMEM operations of one wave complete in order):
//   M-waves (4..7): all HBM traffic -- saved gates / c / d(h) prefetched two steps ahead, the
//                   element-wise BPTT update of the workgroup's 512 (seq, 4-unit) cells, d(gates)
//                   stores, and the bf16 B-operand image of d(gates) in LDS;
//   X-waves (0..3): the MFMAs (weights resident: 128 VGPRs), the write-through publish, the flag,
//                   the poll, the gather + sum, and the reduced dh back to LDS for the M-waves.
// =============================================================================================
#define BK_ROW 136  // bf16 per LDS row of the local d(gates) image (128 + 8: 272 B = 4 banks mod 64)

template <bool FORCE>
__global__ __launch_bounds__(512, 2) void lstm_bwd_cluster_kernel(const ws_lstm_cluster_args p) {
  __shared__ __attribute__((aligned(16))) __bf16 dgl[2][CL_SEQ * BK_ROW];  // [part][seq][local gate col] 34 KB
  __shared__ __attribute__((aligned(16))) f32x4 rec[512];                  // reduced dh per cell, 8 KB
  __shared__ int dead_s;
  const int ntile = p.nseq / 32, ncl_dir = ntile / 2, ncl = 2 * ncl_dir;
  int c, j;  // member j of cluster c
  if (!cluster_of_block(ncl, p.dbg & 16, c, j)) return;
  const int d = c / ncl_dir, cc = c % ncl_dir;
  const int tid = threadIdx.x, lane = tid & 63, n = lane & 31, half = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool xrole = w < 4;
  const int L = p.L;
  const long long gblk = (long long)SQ * 2 * LG, cblk = (long long)SQ * 2 * LH;  // floats per block
  for (int i = tid; i < 512; i += 512) rec[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (tid == 0) dead_s = 0;

  // exchange: X[cluster][parity][dest][src][cell 512] x 16 B
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<char*>(p.xchg) + (long long)c * (2 * 64 * 8192), 0, 2 * 64 * 8192, 0x00020000);
  gu32* flags = (gu32*)(p.flags) + c * 8;

  auto step_t = [&](int step) { return d == 0 ? L - 1 - step : step; };
  __syncthreads();

  // The two roles are separate code paths with their own step loops (and matching barrier sequences
  // A, A2, A3, B), so each gets its own register allocation: 128 VGPRs of resident weights for the
  // X-waves, the two-deep HBM prefetch for the M-waves.
  if (xrole) {
    // ---- resident W^T fragments of unit tiles 2w, 2w+1 (rows = out unit, k = local gate column) ----
    bf16x8 wh[2][8], wl[2][8];
    {
      const float* W = d ? p.whh_r : p.whh_f;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int u = 32 * (2 * w + e) + n;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          float v[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const int kl = 16 * ks + 8 * half + q;  // local gate column -> W_hh row
            v[q] = W[(long long)((kl >> 5) * LH + 32 * j + (kl & 31)) * LH + u];
          }
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            wh[e][ks][q] = (__bf16)v[q];
            wl[e][ks][q] = (__bf16)(v[q] - (float)wh[e][ks][q]);
          }
        }
      }
    }
    for (int step = 0; step < L; ++step) {
      const int par = step & 1;
      __syncthreads();  // A: d(gates) image of this step complete
      const bool dead = dead_s != 0;
      f32x16 acc[2][2];
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[e][st][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        bf16x8 bh[2], bl[2];
#pragma unroll
        for (int st = 0; st < 2; ++st) {
          const int o = (st * 32 + n) * BK_ROW + 16 * ks + 8 * half;
          bh[st] = *reinterpret_cast<const bf16x8*>(&dgl[0][o]);
          bl[st] = *reinterpret_cast<const bf16x8*>(&dgl[1][o]);
        }
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
          for (int st = 0; st < 2; ++st) acc[e][st] = mfma32(wh[e][ks], bh[st], acc[e][st]);
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
          for (int st = 0; st < 2; ++st) acc[e][st] = mfma32(wl[e][ks], bh[st], acc[e][st]);
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
          for (int st = 0; st < 2; ++st) acc[e][st] = mfma32(wh[e][ks], bl[st], acc[e][st]);
      }
      // tile (e, st): rows = units 32(2w+e) + 8q4 + 4half + r -> destination workgroup 2w+e,
      // cell (quad 2q4 + half, seq 32st + n)
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) {
            f32x4 v = {acc[e][st][4 * q4], acc[e][st][4 * q4 + 1], acc[e][st][4 * q4 + 2], acc[e][st][4 * q4 + 3]};
            if (dead) v = f32x4{__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf("")};
            const int dest = 2 * w + e, cell = (2 * q4 + half) * 64 + 32 * st + n;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), xrs,
                                                   cell * 16 + ((par * 8 + dest) * 8 + j) * 8192, 0, SC1);
          }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();  // A2: every publishing wave has drained
      if (tid == 0) __hip_atomic_store(flags + j, (unsigned)(step + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (w == 0 && !dead && !(p.dbg & 1)) {
        unsigned spins = 0;
        while (true) {
          const unsigned v = lane < 8 ? __hip_atomic_load(flags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                      : 0xffffffffu;
          const bool force = FORCE && step == 2 && c == 0 && j == 0;  // test instantiation: a timeout on demand
          if (!force && __all((int)(v >= (unsigned)(step + 1)))) break;
          if (force || ++spins > CL_SPIN_LIMIT) {
            if (lane == 0) cluster_timed_out(p.flags + ncl * 8, p.status, &dead_s);
            break;
          }
          __builtin_amdgcn_s_sleep(2);
        }
      }
      __syncthreads();  // A3: all eight partials of this step are visible
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int ce = tid + 256 * e;  // tid < 256 for X-waves
        u32x4 pv[8];
#pragma unroll
        for (int src = 0; src < 8; ++src)
          pv[src] = __builtin_amdgcn_raw_buffer_load_b128(xrs, ce * 16, ((par * 8 + j) * 8 + src) * 8192, SC1);
        f32x4 sum = __builtin_bit_cast(f32x4, pv[0]);
#pragma unroll
        for (int src = 1; src < 8; ++src) sum += __builtin_bit_cast(f32x4, pv[src]);
        rec[ce] = sum;
      }
      __syncthreads();  // B: reduced dh of this step in LDS
    }
  } else {
    // ---- M-waves: two cells per thread, inputs prefetched two steps ahead ---------------------------
    const int mt = tid & 255;
    const float* hin = p.dhcat;  // BL(512) d(hcat)
    f32x4 pin[2][2][6];          // [slot][cell e][i, f, g, o, dh_in, c_prev]
    f32x4 c_cur[2], dc[2];
    // BL cells through buffer descriptors: base = block (tile 2cc, t) of the array (SGPRs), one VGPR
    // byte offset per cell = tile-in-cluster * (L blocks) + cell, + a scalar gate offset
    int gvo[2], cvo[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int ce = mt + 256 * e, s = ce & 63, q = ce >> 6;
      gvo[e] = (s >> 5) * (int)(L * gblk * 4) + ((d * 256 + 8 * j + q) * 32 + (s & 31)) * 16;
      cvo[e] = (s >> 5) * (int)(L * cblk * 4) + ((d * 64 + 8 * j + q) * 32 + (s & 31)) * 16;
    }
    auto grs = [&](int t) { return mkrsrc(p.gates + ((long long)2 * cc * L + t) * gblk, 0x7fffffffu); };
    auto crs = [&](const float* b, int t) { return mkrsrc(b + ((long long)2 * cc * L + t) * cblk, 0x7fffffffu); };
    auto load_cell = [&](int slot, int e, int t) {
#pragma unroll
      for (int g = 0; g < 4; ++g) pin[slot][e][g] = bld(grs(t), gvo[e], g * 64 * 512);
      pin[slot][e][4] = bld(crs(hin, t), cvo[e], 0);
      const int tp = d == 0 ? max(t - 1, 0) : min(t + 1, L - 1);  // clamped; masked at its use
      pin[slot][e][5] = bld(crs(p.cbuf, tp), cvo[e], 0);
    };
    {
      const int t0 = step_t(0);
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        load_cell(0, e, t0);
        load_cell(1, e, step_t(min(1, L - 1)));
        c_cur[e] = bld(crs(p.cbuf, t0), cvo[e], 0);
        dc[e] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    for (int step = 0; step < L; ++step) {
      const int t = step_t(step);
      const bool has_prev = d == 0 ? (t > 0) : (t < L - 1);
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int ce = mt + 256 * e;
        const int s = ce & 63, q = ce >> 6;
        const f32x4 dhr = rec[ce];
        f32x4 pg[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float ig = pin[0][e][0][r], fg = pin[0][e][1][r], gg = pin[0][e][2][r], og = pin[0][e][3][r];
          const float dhv = pin[0][e][4][r] + dhr[r];
          const float tc = ftanh(c_cur[e][r]);
          const float dov = dhv * tc;
          const float dcv = dc[e][r] + dhv * og * (1.f - tc * tc);
          dc[e][r] = dcv * fg;
          pg[0][r] = dcv * gg * ig * (1.f - ig);
          pg[1][r] = dcv * (has_prev ? pin[0][e][5][r] : 0.f) * fg * (1.f - fg);
          pg[2][r] = dcv * ig * (1.f - gg * gg);
          pg[3][r] = dov * og * (1.f - og);
        }
        c_cur[e] = pin[0][e][5];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          bf16x4 hi, lo;
          split4(pg[g], hi, lo);
          bst(pack_hl4(hi, lo), grs(t), gvo[e], g * 64 * 512);  // BLS
          const int o = s * BK_ROW + g * 32 + 4 * q;
          *reinterpret_cast<bf16x4*>(&dgl[0][o]) = hi;
          *reinterpret_cast<bf16x4*>(&dgl[1][o]) = lo;
        }
        // rotate the two-deep prefetch and refill it two steps ahead
#pragma unroll
        for (int a = 0; a < 6; ++a) pin[0][e][a] = pin[1][e][a];
        load_cell(1, e, step_t(min(step + 2, L - 1)));
      }
      __syncthreads();  // A
      __syncthreads();  // A2
      __syncthreads();  // A3
      __syncthreads();  // B
    }
  }
}

extern "C" int ws_lstm_bwd_cluster(const ws_lstm_cluster_args* a, void* stream) {
  WS_REQUIRE(a && a->gates && a->cbuf && a->dhcat && a->whh_f && a->whh_r && a->xchg && a->flags,
             "ws_lstm_bwd_cluster: null pointer");
  WS_REQUIRE(a->nseq > 0 && a->nseq % 64 == 0 && a->L > 0, "ws_lstm_bwd_cluster: nseq must be a multiple of 64");
  WS_REQUIRE(a->gfmt == WS_GATES_F32, "ws_lstm_bwd_cluster: WS_GATES_F32 only (gfmt %d)", a->gfmt);
  const int nwg = (a->nseq / 32) * 8;
  int dev = 0, cus = 0;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  WS_REQUIRE(nwg <= cus, "ws_lstm_bwd_cluster: %d workgroups must be co-resident but the device has %d CUs", nwg, cus);
  const int grid = cluster_grid(a->nseq / 32, a->dbg & 16);
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(a->flags, 0, ((size_t)(a->nseq / 32) * 8 + 8) * sizeof(unsigned), s);
  WS_REQUIRE(e == hipSuccess, "ws_lstm_bwd_cluster: hipMemsetAsync failed");
  ws_prof_begin(WS_PROF_LSTM_BWD, s);
  if (a->dbg & 8)
    hipLaunchKernelGGL(lstm_bwd_cluster_kernel<true>, dim3(grid), dim3(512), 0, s, *a);
  else
    hipLaunchKernelGGL(lstm_bwd_cluster_kernel<false>, dim3(grid), dim3(512), 0, s, *a);
  ws_prof_end(WS_PROF_LSTM_BWD, s);
  return ws_check_launch("ws_lstm_bwd_cluster");
}
