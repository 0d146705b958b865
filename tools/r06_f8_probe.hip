// Round 6 probe (measurement only, not part of the library): operand layout, scale semantics and issue rate of
// v_mfma_scale_f32_32x32x64_f8f6f4 (FP8 e4m3 operands) next to v_mfma_f32_32x32x16_f16 on gfx950.
//   hipcc --offload-arch=gfx950 -O3 tools/r06_f8_probe.hip -o /tmp/f8probe && /tmp/f8probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void one_mfma(const v8i* a, const v8i* b, const int* sa, const int* sb, f32x16* c) {
  f32x16 acc = {};
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0, sa[threadIdx.x], 0, sb[threadIdx.x]);
  c[threadIdx.x] = acc;
}
__global__ void one_mfma_opsel(const v8i* a, const v8i* b, const int* sa, const int* sb, f32x16* c) {
  f32x16 acc = {};   // scale bytes 2 (A) and 1 (B) of the scale registers
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 2, sa[threadIdx.x], 1, sb[threadIdx.x]);
  c[threadIdx.x] = acc;
}

template <int MODE>   // 0: f16 32x32x16 only; 1: fp8 32x32x64 only; 2: per round 4 f16 + 1 fp8 on the same accumulators (the planned k-step)
__global__ __launch_bounds__(256) void rate(const v8i* a, float* out, int iters) {
  v8i a8 = a[threadIdx.x & 63], b8 = a[64 + (threadIdx.x & 63)];
  f16x8 ah = __builtin_bit_cast(f16x8, __builtin_shufflevector(a8, a8, 0, 1, 2, 3)), bh = __builtin_bit_cast(f16x8, __builtin_shufflevector(b8, b8, 0, 1, 2, 3));
  f32x16 acc[4] = {};
  for (int i = 0; i < iters; i += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (MODE == 0 || MODE == 2) {
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[g], 0, 0, 0);
      }
      if (MODE == 1) {
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[g], 0, 0, 0, 127, 0, 127);
      }
      // one fp8 MFMA per round, the accumulator rotates with the round
      if (MODE == 2) acc[u] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[u], 0, 0, 0, 127, 0, 127);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[g][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// e4m3fn: 1 sign, 4 exponent (bias 7), 3 mantissa; no infinities; 0x7f = NaN
static float f8_to_f(uint8_t v) {
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float x = e == 0 ? std::ldexp((float)m, -9) : std::ldexp(1.f + m / 8.f, e - 7);
  return s ? -x : x;
}

int main() {
  std::vector<uint8_t> A(64 * 32), B(64 * 32);      // [lane][byte]: the physical operand registers
  std::vector<int> SA(64), SB(64);
  srand(7);
  for (auto& v : A) { v = rand() & 0x7f; if ((v & 0x7f) == 0x7f) v = 0x38; if (rand() & 1) v |= 0x80; if (((v >> 3) & 15) > 9) v &= 0xcf; }
  for (auto& v : B) { v = rand() & 0x7f; if ((v & 0x7f) == 0x7f) v = 0x38; if (rand() & 1) v |= 0x80; if (((v >> 3) & 15) > 9) v &= 0xcf; }
  for (int l = 0; l < 64; ++l) {
    SA[l] = (127 + (l % 5) - 2) | ((120 + l % 3) << 16) | (0x55 << 8) | (0x11 << 24);
    SB[l] = (127 - (l % 3)) | ((125 + l % 7) << 8) | (0x22 << 16) | (0x33 << 24);
  }
  v8i *da, *db;
  int *dsa, *dsb;
  f32x16* dc;
  CK(hipMalloc(&da, 64 * 32)); CK(hipMalloc(&db, 64 * 32)); CK(hipMalloc(&dsa, 256)); CK(hipMalloc(&dsb, 256)); CK(hipMalloc(&dc, 64 * 64));
  CK(hipMemcpy(da, A.data(), 2048, hipMemcpyHostToDevice)); CK(hipMemcpy(db, B.data(), 2048, hipMemcpyHostToDevice));
  CK(hipMemcpy(dsa, SA.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dsb, SB.data(), 256, hipMemcpyHostToDevice));
  for (int variant = 0; variant < 3; ++variant) {   // 0: all scales 1.0; 1: per-lane scales, byte 0; 2: bytes 2 (A) / 1 (B) by op_sel
    std::vector<int> sa_(SA), sb_(SB);
    if (variant == 0) for (int l = 0; l < 64; ++l) sa_[l] = sb_[l] = 127;
    CK(hipMemcpy(dsa, sa_.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dsb, sb_.data(), 256, hipMemcpyHostToDevice));
    if (variant < 2) hipLaunchKernelGGL(one_mfma, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dc);
    else hipLaunchKernelGGL(one_mfma_opsel, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dc);
    CK(hipDeviceSynchronize());
    std::vector<float> C(64 * 16);
    CK(hipMemcpy(C.data(), dc, 4096, hipMemcpyDeviceToHost));
    const int sha = variant == 2 ? 16 : 0, shb = variant == 2 ? 8 : 0;
    // D register r of lane l: column l & 31, row (r & 3) + 8 (r >> 2) + 4 (l >> 5).  Operand lanes: row / column l & 31.
    // hypothesis 0: a lane's 32 bytes are ONE scale block (k = 32 (l >> 5) + byte), scaled by the lane's own scale byte
    // hypothesis 1: bytes 0..15 of lanes l, l + 32 form block 0 (k = 16 (l >> 5) + byte), bytes 16..31 block 1; block b of a
    //               row / column takes the scale byte of lane (row | column) + 32 b
    for (int hyp = 0; hyp < 2; ++hyp) {
      double worst = 0, big = 0;
      for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 16; ++r) {
          const int col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
          double ref = 0;
          for (int blk = 0; blk < 2; ++blk) {
            const double sa = std::ldexp(1.0, ((sa_[row + 32 * blk] >> sha) & 255) - 127), sb = std::ldexp(1.0, ((sb_[col + 32 * blk] >> shb) & 255) - 127);
            double part = 0;
            for (int j = 0; j < 32; ++j) {
              const int lane_hf = hyp == 0 ? blk : j >> 4, byte = hyp == 0 ? j : 16 * blk + (j & 15);
              part += (double)f8_to_f(A[(row + 32 * lane_hf) * 32 + byte]) * f8_to_f(B[(col + 32 * lane_hf) * 32 + byte]);
            }
            ref += part * sa * sb;
          }
          worst = std::fmax(worst, std::fabs(ref - C[l * 16 + r]));
          big = std::fmax(big, std::fabs(ref));
        }
      printf("layout check (scales %d, hypothesis %d): max |D - model| = %.3e  (max |model| = %.3e)  %s\n", variant, hyp, worst, big,
             worst <= 1e-5 * big ? "MODEL HOLDS" : "MODEL FAILS");
    }
  }

  // rate
  float* dout;
  const int WG = 2048, iters = 2000;
  CK(hipMalloc(&dout, WG * 256 * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int mode = 0; mode < 3; ++mode) {
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipEventRecord(e0));
      if (mode == 0) hipLaunchKernelGGL(rate<0>, dim3(WG), dim3(256), 0, 0, da, dout, iters);
      if (mode == 1) hipLaunchKernelGGL(rate<1>, dim3(WG), dim3(256), 0, 0, da, dout, iters);
      if (mode == 2) hipLaunchKernelGGL(rate<2>, dim3(WG), dim3(256), 0, 0, da, dout, iters);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      best = std::fmin(best, ms);
    }
    const double waves = (double)WG * 4, n16 = mode == 1 ? 0 : 4.0 * iters, n64 = mode == 0 ? 0 : (mode == 1 ? 4.0 : 1.0) * iters;
    const double flop = waves * (n16 * 32 * 32 * 16 * 2 + n64 * 32 * 32 * 64 * 2);
    printf("rate mode %d (%s): %.3f ms, %.0f TFLOP/s; per wave and iteration %.1f ns\n", mode,
           mode == 0 ? "4 x f16 32x32x16" : mode == 1 ? "4 x fp8 32x32x64" : "4 x f16 32x32x16 + 1 x fp8 32x32x64", best,
           flop / best * 1e-9, best * 1e6 / iters / (WG * 4 / 1024.0));
  }
  return 0;
}
