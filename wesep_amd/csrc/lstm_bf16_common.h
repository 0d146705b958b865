// Shared device helpers of the split-bf16 LSTM recurrence kernels (lstm_bf16.hip, lstm_bf16_s16.hip).
#pragma once
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#define LH WS_LSTM_H  // 256
#define LG (4 * LH)   // 1024
#define SQ 32         // sequences per workgroup (MFMA N)
#define HROW 264      // bf16 per LDS row of h      (256 + 8: 528 B = 4 banks mod 64)
#define DROW 1032     // bf16 per LDS row of dgates (1024 + 8: 2064 B = 4 banks mod 64)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Weight stream: raw buffer loads (SGPR descriptor + SGPR step offset + one VGPR lane offset), so the
// stream costs no 64-bit VGPR address arithmetic.  `soff` carries an opaque zero so the compiler
// cannot see that the addresses repeat every step (hoisting 128 fragments out of the loop = spills).
__device__ __forceinline__ bf16x8 wload(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff) {
  return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0));
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t mkrsrc(const float* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, bytes, 0x00020000);
}
// cache policy of the once-through activation streams (gates, cells, h): 2 = nt (stream, evict first)
#ifndef WS_STREAM_AUX
#define WS_STREAM_AUX 2
#endif
__device__ __forceinline__ f32x4 bld(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, WS_STREAM_AUX));
}
// Stores carry their whole offset in the VGPR address / the instruction's immediate and NO register soffset.  Found on
// the MI355X in round 3 (profiles/r03_store_hazard.md, tools/store_hazard_repro.hip): a 16-byte buffer store needs one
// wait state before a VALU instruction may overwrite its data registers ALSO when its soffset is an SGPR, but hipcc
// (ROCm 7.2) pads that hazard only for stores without a register soffset (GCNHazardRecognizer::createsVALUHazard) -- the
// store then writes the NEW register contents in the lanes it reads last (0.2 % of the dwords in the reproducer; the
// round-2 recurrence kernels had 59 such sites).  With a constant-zero soffset the compiler inserts the wait states
// itself; tools/scan_store_hazard.py checks the generated ISA of every kernel for the pattern.
__device__ __forceinline__ void bst(const f32x4& v, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff + soff, 0, WS_STREAM_AUX);
}

__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float fsig(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float ftanh(float x) {
  const float ax = fabsf(x);
  const float e = __expf(-2.f * ax);  // in (0, 1]: no overflow
  const float t = (1.f - e) * __builtin_amdgcn_rcpf(1.f + e);
  return copysignf(t, x);
}

// The fp16 recurrences (lstm_cluster2.hip, the H16 fused forward of lstm_fused.hip): h in (-1, 1) as ONE fp16 operand of
// v_mfma_f32_32x32x16_f16 against W as fp16 hi / lo of 256 w; the accumulator then carries 256 x the pre-activation.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));   // 32 FP8 operand bytes of v_mfma_scale_f32_32x32x64_f8f6f4
__device__ __forceinline__ f32x16 mfma16h(const f16x8& a, const f16x8& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
// sigmoid / tanh of a / 256 with the scale folded into the argument of v_exp_f32 (2^x): one multiply per activation instead of
// two; the same v_exp / v_rcp as fsig / ftanh
__device__ __forceinline__ float c2_sig256(float a) {
  return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(a * (-1.4426950408889634f / 256.f)));
}
__device__ __forceinline__ float c2_tanh256(float a) {
  const float e = __builtin_amdgcn_exp2f(fabsf(a) * (-2.f * 1.4426950408889634f / 256.f));  // in (0, 1]: no overflow
  const float t = (1.f - e) * __builtin_amdgcn_rcpf(1.f + e);
  return copysignf(t, a);
}

__device__ __forceinline__ void split4(const f32x4& v, bf16x4& hi, bf16x4& lo) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    hi[j] = (__bf16)v[j];
    lo[j] = (__bf16)(v[j] - (float)hi[j]);
  }
}

// Split-bf16 storage ("BLS", include/wesep_hip.h): a 4-byte element holds the two terms of the split
// product -- bits 31..16 = bf16 hi, bits 15..0 = bf16 lo, value = hi + lo -- so whoever consumes the
// buffer as an MFMA operand (gemm_b2p / gemm_tnb / the fused recurrence) rebuilds its fragments with
// two v_perm_b32 per element pair instead of two conversions and a subtraction per element.  Same
// bytes as fp32; pack / unpack are the same 16-bit 2x2 transpose.
__device__ __forceinline__ f32x4 pack_hl4(const bf16x4& hi, const bf16x4& lo) {
  const uint2 h = __builtin_bit_cast(uint2, hi), l = __builtin_bit_cast(uint2, lo);
  u32x4 r;
  r[0] = __builtin_amdgcn_perm(h.x, l.x, WS_SEL_LO16);
  r[1] = __builtin_amdgcn_perm(h.x, l.x, WS_SEL_HI16);
  r[2] = __builtin_amdgcn_perm(h.y, l.y, WS_SEL_LO16);
  r[3] = __builtin_amdgcn_perm(h.y, l.y, WS_SEL_HI16);
  return __builtin_bit_cast(f32x4, r);
}
__device__ __forceinline__ void unpack_hl4(const f32x4& v, bf16x4& hi, bf16x4& lo) {
  const u32x4 u = __builtin_bit_cast(u32x4, v);
  uint2 h, l;
  h.x = __builtin_amdgcn_perm(u[1], u[0], WS_SEL_HI16);
  h.y = __builtin_amdgcn_perm(u[3], u[2], WS_SEL_HI16);
  l.x = __builtin_amdgcn_perm(u[1], u[0], WS_SEL_LO16);
  l.y = __builtin_amdgcn_perm(u[3], u[2], WS_SEL_LO16);
  hi = __builtin_bit_cast(bf16x4, h);
  lo = __builtin_bit_cast(bf16x4, l);
}


// ---- 2-byte storage of the saved recurrence state (WS_GATES_H2 / WS_GATES_H2S, include/wesep_hip.h, ABI v15) ---------
// BLH(C): the BL(C) index formula on 2-byte elements: a lane's 4-column cell is 8 bytes (one buffer_load/store_dwordx2),
// 32 lanes 256 contiguous bytes.  Activated gates travel as unorm16 -- i, f, o in (0, 1): u = floor(x * 65535 + 0.5);
// g in (-1, 1): u = floor((x + 1) * 32767.5 + 0.5) -- i.e. a FIXED-point code: its absolute error (7.7e-6 / 1.6e-5) is
// uniform, where fp16 leaves 2.4e-4 exactly where the gates saturate and the derivative factor (1 - i) is small.
// d(gates) travel as bf16 = the hi term of the split pair (no range problem: fp32's exponent).
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
#define WS_U16_SIG 65535.f
#define WS_U16_TANH 32767.5f

__device__ __forceinline__ u32x2 bld8(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, WS_STREAM_AUX));
}
__device__ __forceinline__ void bst8(const u32x2& v, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
  __builtin_amdgcn_raw_buffer_store_b64(v, r, voff + soff, 0, WS_STREAM_AUX);  // (no register soffset: see bst)
}
// 4 gate values -> 4 unorm16 codes (TANH: the g gate).  floor(x * s + o) with o = 0.5 (+ s for the tanh gate) by
// truncation: the argument is never negative
template <bool TANH>
__device__ __forceinline__ u32x2 enc_u16x4(const f32x4& v) {
  const float s = TANH ? WS_U16_TANH : WS_U16_SIG, o = TANH ? WS_U16_TANH + 0.5f : 0.5f;
  unsigned u[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) u[j] = (unsigned)__builtin_fmaf(v[j], s, o);
  return u32x2{u[0] | (u[1] << 16), u[2] | (u[3] << 16)};
}
template <bool TANH>
__device__ __forceinline__ f32x4 dec_u16x4(const u32x2& c) {
  const float s = TANH ? 1.f / WS_U16_TANH : 1.f / WS_U16_SIG, o = TANH ? -1.f : 0.f;
  f32x4 v;
  v[0] = __builtin_fmaf((float)(c[0] & 0xffffu), s, o);
  v[1] = __builtin_fmaf((float)(c[0] >> 16), s, o);
  v[2] = __builtin_fmaf((float)(c[1] & 0xffffu), s, o);
  v[3] = __builtin_fmaf((float)(c[1] >> 16), s, o);
  return v;
}
__device__ __forceinline__ u32x2 bf16x4_bits(const bf16x4& v) { return __builtin_bit_cast(u32x2, v); }
// WS_GATES_H2F: d(gates) as fp16(x * S), round to nearest even; S = ws_dgates_scale(*amax) (common.h)
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
// The BPTT kernels run their WHOLE recurrence in scaled units when GF == WS_GATES_H2F: d(hcat) enters through one fma
// (dh * S + recurrent part: no extra instruction), everything downstream -- d(c), the d(gates), their B-operand image
// for the recurrent MFMA, the partial d(h) a pair exchanges -- is linear in it, and a power of two scales exactly; so the
// value to store is already x * S and only the conversion remains.  NOT clamped (round 5; rounds 3-4 stored
// fmed3(x, +-65504)): a scaled d(gates) beyond fp16's range becomes +-Inf and a NaN stays a NaN, so both reach the
// weight-gradient GEMMs and d(xn) as non-finite values and trip ws_grad_norms' guard -- the optimizer step is SKIPPED and
// counted, the way a loss-scaler treats an overflow -- where the clamp let silently clipped gradients (and, through
// v_med3's NaN rule, a NaN turned into -65504) through to the weights.  S leaves 2^7 of headroom above max |d(hcat)|.
__device__ __forceinline__ u32x2 enc_f16x4(const f32x4& vs) {
  f16x4 h;
#pragma unroll
  for (int j = 0; j < 4; ++j) h[j] = (_Float16)vs[j];
  return __builtin_bit_cast(u32x2, h);
}
// the 8-byte d(gates) cell of the two in-place-capable 2-byte formats (H2F: v is the SCALED value)
template <int GF>
__device__ __forceinline__ u32x2 enc_dgates(const f32x4& v, const bf16x4& hi) {
  if constexpr (GF == WS_GATES_H2F) return enc_f16x4(v);
  else return bf16x4_bits(hi);
}
// d(h) of this step = d(hcat) from the layer above (scaled on entry for H2F) + the recurrent part
template <int GF>
__device__ __forceinline__ float dh_in(float dh, float rec, float S) {
  if constexpr (GF == WS_GATES_H2F) return __builtin_fmaf(dh, S, rec);
  else return dh + rec;
}

// A saved gate cell as the BPTT kernels keep it between its (prefetching) load and its use one step later: the RAW
// unorm16 codes -- decoding at the load would put the conversions, and with them the wait for the load, in front of the
// scheduling fence that follows the prefetch
template <int GF> struct gate_cell { typedef f32x4 type; };
template <> struct gate_cell<WS_GATES_H2> { typedef u32x2 type; };
template <> struct gate_cell<WS_GATES_H2S> { typedef u32x2 type; };
template <> struct gate_cell<WS_GATES_H2F> { typedef u32x2 type; };
template <bool TANH> __device__ __forceinline__ f32x4 gate_val(const f32x4& c) { return c; }
template <bool TANH> __device__ __forceinline__ f32x4 gate_val(const u32x2& c) { return dec_u16x4<TANH>(c); }
