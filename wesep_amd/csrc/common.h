// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels of wesep_amd.
// wave = 64 lanes everywhere in this tree.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/wesep_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define WS_WAVE 64

// v_perm_b32 selectors of the split-bf16 storage format BLS (include/wesep_hip.h: a 4-byte element = bf16 hi << 16 |
// bf16 lo): __builtin_amdgcn_perm(S0, S1, sel) picks bytes from {S0 = bytes 7..4, S1 = bytes 3..0}
#define WS_SEL_LO16 0x05040100u  // S1.lo16 | S0.lo16 << 16
#define WS_SEL_HI16 0x07060302u  // S1.hi16 | S0.hi16 << 16

// ---- error plumbing ---------------------------------------------------------
void ws_set_error(const char* fmt, ...);
int ws_check_launch(const char* what);

#define WS_REQUIRE(cond, ...)            \
  do {                                   \
    if (!(cond)) {                       \
      ws_set_error(__VA_ARGS__);         \
      return WS_ERR_INVALID;             \
    }                                    \
  } while (0)

// ---- profiling hooks (HIP events on the launch stream; see prof.hip) ---------
void ws_prof_begin(int kind, hipStream_t s);
void ws_prof_end(int kind, hipStream_t s);

// ---- split-bf16 GEMM launchers (gemm_bf16.hip), selected by bit 2 of the `vec` argument ------
int ws_launch_gemm_nt_bf16(const ws_gemm_nt_args* a, dim3 grid, hipStream_t s);
int ws_launch_gemm_tn_bf16(const ws_gemm_tn_args* a, dim3 grid, hipStream_t s);

// ---- split-bf16 LSTM launchers (lstm_bf16.hip), selected by ws_lstm_args.mode ----------------
int ws_launch_lstm_pack_bf16(const float* whh_f, const float* whh_r, float* pack_fwd, float* pack_bwd,
                             hipStream_t s);
int ws_launch_lstm_fwd_bf16(const ws_lstm_args* a, hipStream_t s);
int ws_launch_lstm_bwd_bf16(const ws_lstm_args* a, hipStream_t s);
// 16-sequence workgroups on the blocked layout (lstm_bf16_s16.hip), mode WS_LSTM_BF16X3_BLK16
int ws_launch_lstm_pack_s16(const float* whh_f, const float* whh_r, float* pack_fwd, float* pack_bwd,
                            hipStream_t s);
int ws_launch_lstm_fwd_s16(const ws_lstm_args* a, hipStream_t s);
int ws_launch_lstm_bwd_s16(const ws_lstm_args* a, hipStream_t s);

// ---- device helpers -----------------------------------------------------------
__device__ __forceinline__ float ws_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Block-wide sum for blockDim.x <= 1024 (multiple of 64). `red` = >= 16 floats of LDS.
// Every thread gets the total.  Contains two barriers.
__device__ __forceinline__ float ws_block_sum(float v, float* red) {
  v = ws_wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}

// WS_GATES_H2F (wesep_hip.h): the power of two that scales d(gates) into fp16 -- max |d(hcat)| of the launch (float bits in
// `amax_bits`) lands in [2^8, 2^9) (rounds 3-4: [2^10, 2^11); 2^7 of headroom below fp16's 65504 for what the BPTT
// accumulates on top of d(hcat), full 11-bit precision down to 2^-22 of the maximum); 1 for a zero / non-finite maximum.
// Exact to undo (ws_dgates_scale_inv).
__device__ __forceinline__ float ws_dgates_scale(unsigned amax_bits) {
  const int e = (int)((amax_bits >> 23) & 0xffu);
  if (e == 0 || e == 255) return 1.f;
  const int se = min(max(WS_DGATES_EXP + 254 - e, 1), 253);  // 2^(WS_DGATES_EXP - (e - 127)); S and 1 / S both stay normal numbers
  return __uint_as_float((unsigned)se << 23);
}
__device__ __forceinline__ float ws_dgates_scale_inv(unsigned amax_bits) {
  const float s = ws_dgates_scale(amax_bits);
  return __uint_as_float((unsigned)(254 - (int)(__float_as_uint(s) >> 23)) << 23);  // 2^-k, exact
}

// Agent-scope (sc1) single-word accesses: a store that is written through to memory and a load that does not trust this
// XCD's L2 -- coherent across the eight XCDs access by access, WITHOUT a cache-wide fence.  (An agent-scope release
// fence -- __threadfence() -- writes back the whole L2 of the XCD; inside kernels that stream hundreds of MB through that
// L2 it cost 0.3-0.4 ms per launch: measured in round 4, the first version of the epilogues below.)
__device__ __forceinline__ void ws_st_agent(float* p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ws_ld_agent(const float* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// True in exactly one workgroup of the launch -- the last one to get here -- for all of its threads.  Protocol (recipe R1
// of cdna_hip_programming.md Guideline 16, as in the cluster recurrences): the partials a workgroup leaves for the last one
// are stored with ws_st_agent, every thread drains its stores (vmcnt(0)), barrier, ONE relaxed agent-scope atomic; the
// last workgroup reads the others' partials with ws_ld_agent.  `counter`: a device word that is 0 at launch; the last
// workgroup leaves it at 0 again.  For "the workgroups add their partials up themselves" epilogues: sums run over the
// partials in index order, so the result does not depend on WHICH workgroup came last (deterministic), and the separate
// reduction launch -- which on a busy GPU can sit out a whole weight-gradient GEMM of another stream before it gets a CU --
// disappears.  Contains two barriers.
// (the ordering below -- RELAXED agent-scope atomics + a hand-written `s_waitcnt vmcnt(0)` -- is a statement
//  about gfx9 (gfx942 / gfx950): stores are tracked by vmcnt and sc1 accesses by-pass the XCD's L2.  It is NOT what the HIP
//  memory model promises in general: targets with a separate store counter (vscnt, gfx10+) would read stale partials.)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx942__) && !defined(__gfx950__)
#error "ws_last_block / ws_tree_sum256 rely on gfx942 / gfx950 memory ordering (vmcnt tracks stores; sc1 = write-through)"
#endif
__device__ __forceinline__ bool ws_last_block(unsigned* counter, unsigned nblocks) {
  __shared__ unsigned last_s;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last_s = t == nblocks - 1u;
    if (last_s) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  return last_s != 0u;
}

// Deterministic sum of the launch's per-workgroup partial vectors of 256 floats, pslab[b][256], b < nblocks = gridDim.x,
// into pout[256], done by the workgroups themselves (blockDim.x = 256) in two levels so that no thread walks more than
// 32 + nblocks / 32 partials: the last finisher of every group of 32 consecutive workgroups adds its group's partials up in
// index order into pslab[nblocks + group][256]; the last of those adds the group sums up in group order.  The result does
// not depend on arrival order.  pslab needs nblocks + ceil(nblocks / 32) rows; counters: 1 + ceil(nblocks / 32) words,
// zero at launch, left at zero.  Call with all threads of every workgroup after the workgroup's own row is written WITH
// ws_st_agent (see ws_last_block).
__device__ __forceinline__ void ws_tree_sum256(float* pslab, unsigned nblocks, float* pout, unsigned* counters) {
  const unsigned grp = blockIdx.x >> 5, ngrp = (nblocks + 31u) >> 5;
  const unsigned in_grp = min(32u, nblocks - grp * 32u);
  if (!ws_last_block(counters + 1 + grp, in_grp)) return;
  float t = 0.f;
  for (unsigned k = 0; k < in_grp; ++k) t += ws_ld_agent(pslab + (long long)(grp * 32u + k) * 256 + threadIdx.x);
  ws_st_agent(pslab + (long long)(nblocks + grp) * 256 + threadIdx.x, t);
  if (!ws_last_block(counters, ngrp)) return;
  float s = 0.f;
  for (unsigned g = 0; g < ngrp; ++g) s += ws_ld_agent(pslab + (long long)(nblocks + g) * 256 + threadIdx.x);
  pout[threadIdx.x] = s;
}

__device__ __forceinline__ float ws_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }

// row index -> element offset under the two-level row addressing used across the C ABI:
//   off(m) = (m / div) * s1 + (m % div) * s2
__device__ __forceinline__ long long ws_row_off(int m, int div, long long s1, long long s2) {
  const int q = m / div;
  return (long long)q * s1 + (long long)(m - q * div) * s2;
}
