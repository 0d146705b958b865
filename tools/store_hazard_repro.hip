// Standalone reproducer (no torch, no wesep code) of the store-data hazard found on the MI355X in round 3
// (profiles/r03_store_hazard.md):
//
//     buffer_store_dwordx4 v[8:11], vaddr, s[rsrc], sOFF offen      ; 16-byte MUBUF store, soffset in an SGPR
//     v_mov_b32 v8..v11, POISON                                      ; the very next instructions overwrite its data
//
// hipcc's hazard recognizer (GCNHazardRecognizer::createsVALUHazard) pads a VALU write behind a >64-bit MUBUF store
// only when the store has NO register soffset; with one it assumes the hardware has read the data already.  Each
// variant below issues the exact sequence from inline asm (physical registers v8..v11, so nothing the compiler does
// can separate the two instructions) with N wait states in between, and the host counts the stored dwords that hold
// POISON instead of the data.
//
//     hipcc --offload-arch=gfx950 -O2 tools/store_hazard_repro.hip -o tools/store_hazard_repro && tools/store_hazard_repro
//
// Output: one line per (store form, wait states, with / without a bandwidth aggressor on a second stream).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>

#include <vector>

#define HIP_OK(x)                                                        \
  do {                                                                   \
    hipError_t e_ = (x);                                                 \
    if (e_ != hipSuccess) {                                              \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));            \
      exit(2);                                                           \
    }                                                                    \
  } while (0)

typedef int i32x4 __attribute__((ext_vector_type(4)));
#define POISON 0x7fc12345u

__device__ __forceinline__ unsigned key(unsigned cell, unsigned c) { return (cell * 4u + c) * 2654435761u | 1u; }

// KIND 0: MUBUF, SGPR soffset   1: MUBUF, soffset 0 (the form hipcc pads)   2: global_store (padded by hipcc too)
#define VARIANT(NAME, STORE, NOP)                                                                                    \
  __global__ void NAME(unsigned* out, int iters, int soff) {                                                         \
    const unsigned gt = blockIdx.x * blockDim.x + threadIdx.x, nthr = gridDim.x * blockDim.x;                       \
    const unsigned long long base_ = (unsigned long long)out;                                                        \
    const i32x4 rsrc = {__builtin_amdgcn_readfirstlane((int)(unsigned)base_),                                       \
                        __builtin_amdgcn_readfirstlane((int)((base_ >> 32) & 0xffffu)), 0x7fffffff, 0x00020000};      \
    for (int it = 0; it < iters; ++it) {                                                                             \
      const unsigned cell = it * nthr + gt;                                                                          \
      unsigned a = key(cell, 0), b = key(cell, 1), c = key(cell, 2), d = key(cell, 3);                               \
      const unsigned voff = cell * 16u - (unsigned)soff;                                                             \
      unsigned long long addr = (unsigned long long)out + (unsigned long long)cell * 16u;                            \
      const unsigned poison = POISON;                                                                                \
      asm volatile(STORE "\n" NOP "\n"                                                                               \
                   "v_mov_b32 v8, %7\n v_mov_b32 v9, %7\n v_mov_b32 v10, %7\n v_mov_b32 v11, %7\n"                   \
                   : "+{v8}"(a), "+{v9}"(b), "+{v10}"(c), "+{v11}"(d)                                                \
                   : "v"(voff), "s"(rsrc), "s"(soff), "v"(poison), "v"(addr)                                         \
                   : "memory");                                                                                      \
      if (a != POISON) out[0] = 0; /* keep the asm's outputs alive */                                                \
    }                                                                                                                \
  }

#define ST_SOFF "buffer_store_dwordx4 v[8:11], %4, %5, %6 offen"
#define ST_SOFF_NT "buffer_store_dwordx4 v[8:11], %4, %5, %6 offen nt"
#define ST_ZERO "buffer_store_dwordx4 v[8:11], %4, %5, 0 offen"
#define ST_GLOBAL "global_store_dwordx4 %8, v[8:11], off"
VARIANT(soff_n0, ST_SOFF, "")
VARIANT(soff_n1, ST_SOFF, "s_nop 0")
VARIANT(soff_n2, ST_SOFF, "s_nop 1")
VARIANT(soff_n4, ST_SOFF, "s_nop 3")
VARIANT(soff_n8, ST_SOFF, "s_nop 7")
VARIANT(soffnt_n0, ST_SOFF_NT, "")
VARIANT(soffnt_n2, ST_SOFF_NT, "s_nop 1")
VARIANT(zero_n0, ST_ZERO, "")
VARIANT(zero_n1, ST_ZERO, "s_nop 0")
VARIANT(zero_n2, ST_ZERO, "s_nop 1")
VARIANT(zero_n4, ST_ZERO, "s_nop 3")
VARIANT(glob_n0, ST_GLOBAL, "")
VARIANT(glob_n1, ST_GLOBAL, "s_nop 0")
VARIANT(glob_n2, ST_GLOBAL, "s_nop 1")
VARIANT(glob_n4, ST_GLOBAL, "s_nop 3")

// LDS-write forms of the same question (the round-2 cross-stream corruption hit kernels that exchange through LDS --
// STFT / iSTFT butterflies, rocFFT, the fp32 gemm_nt -- while an MFMA + LDS + HBM kernel ran beside them;
// profiles/r02_kernel_race.md).  hipcc pads nothing behind a ds_write; stft.hip has a VALU write to the data
// registers of a ds_write2_b64 two slots behind it.
#define LVARIANT(NAME, STORE, NOP)                                                                                   \
  __global__ void NAME(unsigned* out, int iters, int soff) {                                                         \
    __shared__ uint4 lds[256];                                                                                       \
    const unsigned gt = blockIdx.x * blockDim.x + threadIdx.x, nthr = gridDim.x * blockDim.x;                       \
    const unsigned laddr = (unsigned)(unsigned long long)(&lds[threadIdx.x]);                                        \
    for (int it = 0; it < iters; ++it) {                                                                             \
      const unsigned cell = it * nthr + gt;                                                                          \
      unsigned a = key(cell, 0), b = key(cell, 1), c = key(cell, 2), d = key(cell, 3);                               \
      const unsigned poison = POISON;                                                                                \
      asm volatile(STORE "\n" NOP "\n"                                                                               \
                   "v_mov_b32 v8, %5\n v_mov_b32 v9, %5\n v_mov_b32 v10, %5\n v_mov_b32 v11, %5\n"                   \
                   "s_waitcnt lgkmcnt(0)\n"                                                                          \
                   : "+{v8}"(a), "+{v9}"(b), "+{v10}"(c), "+{v11}"(d)                                                \
                   : "v"(laddr), "v"(poison)                                                                         \
                   : "memory");                                                                                      \
      reinterpret_cast<uint4*>(out)[cell] = lds[threadIdx.x];                                                        \
      if (a != POISON) out[0] = 0;                                                                                   \
    }                                                                                                                \
  }
#define LW128 "ds_write_b128 %4, v[8:11]"
#define LW2X64 "ds_write2_b64 %4, v[8:9], v[10:11] offset1:1"
LVARIANT(lds128_n0, LW128, "")
LVARIANT(lds128_n1, LW128, "s_nop 0")
LVARIANT(lds128_n2, LW128, "s_nop 1")
LVARIANT(lds128_n4, LW128, "s_nop 3")
LVARIANT(lds2x64_n0, LW2X64, "")
LVARIANT(lds2x64_n1, LW2X64, "s_nop 0")
LVARIANT(lds2x64_n2, LW2X64, "s_nop 1")
LVARIANT(lds2x64_n4, LW2X64, "s_nop 3")

// The shape r02_kernel_race.md found necessary in an aggressor: MFMA + LDS staging + HBM loads inside the loop,
// one resident workgroup per CU (64 KB of LDS, 256 threads) so victim workgroups share its CUs.
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
__global__ void __launch_bounds__(256) mfma_aggressor(const float4* __restrict__ src, float* __restrict__ dst, size_t n,
                                                       int reps) {
  __shared__ float4 stage[4096];  // 64 KB
  f32x16_t acc0 = {}, acc1 = {};
  const int t = threadIdx.x;
  size_t i = ((size_t)blockIdx.x * 256 + t) % n;
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int j = 0; j < 16; ++j) stage[j * 256 + t] = src[(i + (size_t)j * 65536) % n];
    i = (i + 16 * 65536 + 256 * 977) % n;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float4 u = stage[(j * 256 + t * 17) & 4095], v = stage[(j * 256 + t * 33 + 7) & 4095];
      bf16x8_t a, b;
      __builtin_memcpy(&a, &u, 16);
      __builtin_memcpy(&b, &v, 16);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc1, 0, 0, 0);
    }
    __syncthreads();
  }
  float s = 0.f;
  for (int k = 0; k < 16; ++k) s += acc0[k] + acc1[k];
  dst[(size_t)blockIdx.x * 256 + t] = s;
}

__global__ void aggressor(const float4* __restrict__ src, float4* __restrict__ dst, size_t n, int reps) {
  for (int r = 0; r < reps; ++r)
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

struct V {
  const char* name;
  void (*fn)(unsigned*, int, int);
  int soff;
};

int main() {
  const int blocks = 2048, threads = 256, iters = 16;
  const size_t cells = (size_t)blocks * threads * iters, bytes = cells * 16;
  unsigned* out = nullptr;
  HIP_OK(hipMalloc(reinterpret_cast<void**>(&out), bytes));
  const size_t agn = (size_t)1 << 26;  // 1 GiB per buffer of float4
  float4 *as = nullptr, *ad = nullptr;
  HIP_OK(hipMalloc(reinterpret_cast<void**>(&as), agn * 16));
  HIP_OK(hipMalloc(reinterpret_cast<void**>(&ad), agn * 16));
  HIP_OK(hipMemset(as, 1, agn * 16));
  hipStream_t s1, s2;
  HIP_OK(hipStreamCreate(&s1));
  HIP_OK(hipStreamCreate(&s2));
  hipDeviceProp_t prop;
  HIP_OK(hipGetDeviceProperties(&prop, 0));
  printf("# %s; %d x %d threads x %d stores of 16 B per variant; POISON dwords counted on the host\n", prop.name, blocks,
         threads, iters);
  const V vs[] = {{"mubuf sgpr-soffset, +0 wait states", soff_n0, 4096},   {"mubuf sgpr-soffset, +1", soff_n1, 4096},
                  {"mubuf sgpr-soffset, +2", soff_n2, 4096},               {"mubuf sgpr-soffset, +4", soff_n4, 4096},
                  {"mubuf sgpr-soffset, +8", soff_n8, 4096},               {"mubuf sgpr-soffset nt, +0", soffnt_n0, 4096},
                  {"mubuf sgpr-soffset nt, +2", soffnt_n2, 4096},          {"mubuf zero soffset, +0", zero_n0, 0},
                  {"mubuf zero soffset, +1", zero_n1, 0},                  {"mubuf zero soffset, +2", zero_n2, 0},
                  {"mubuf zero soffset, +4", zero_n4, 0},                  {"global_store, +0", glob_n0, 0},
                  {"global_store, +1", glob_n1, 0},                        {"global_store, +2", glob_n2, 0},
                  {"global_store, +4", glob_n4, 0},
                  {"ds_write_b128, +0", lds128_n0, 0},                     {"ds_write_b128, +1", lds128_n1, 0},
                  {"ds_write_b128, +2", lds128_n2, 0},                     {"ds_write_b128, +4", lds128_n4, 0},
                  {"ds_write2_b64, +0", lds2x64_n0, 0},                    {"ds_write2_b64, +1", lds2x64_n1, 0},
                  {"ds_write2_b64, +2", lds2x64_n2, 0},                    {"ds_write2_b64, +4", lds2x64_n4, 0}};
  std::vector<unsigned> h(cells * 4);
  const char* beside[] = {"alone               ", "beside a copy kernel", "beside MFMA+LDS+HBM "};
  for (int load = 0; load < 3; ++load)
    for (const V& v : vs) {
      HIP_OK(hipMemsetAsync(out, 0, bytes, s1));
      HIP_OK(hipStreamSynchronize(s1));
      if (load == 1) hipLaunchKernelGGL(aggressor, dim3(1024), dim3(256), 0, s2, as, ad, agn, 3);
      if (load == 2)
        hipLaunchKernelGGL(mfma_aggressor, dim3(prop.multiProcessorCount), dim3(256), 0, s2, as,
                           reinterpret_cast<float*>(ad), agn, 3000);
      if (load) usleep(3000);  // the aggressor is resident before the victim starts
      hipLaunchKernelGGL(v.fn, dim3(blocks), dim3(threads), 0, s1, out, iters, v.soff);
      HIP_OK(hipStreamSynchronize(s1));
      HIP_OK(hipStreamSynchronize(s2));
      HIP_OK(hipMemcpy(h.data(), out, bytes, hipMemcpyDeviceToHost));
      size_t poison = 0, other = 0, per_c[4] = {0, 0, 0, 0}, per_bank[4] = {0, 0, 0, 0};
      for (size_t cell = 0; cell < cells; ++cell)
        for (unsigned c = 0; c < 4; ++c) {
          const unsigned got = h[cell * 4 + c], want = (unsigned)((cell * 4u + c) * 2654435761u | 1u);
          if (cell == 0 && c == 0) continue;  // out[0] is scribbled by the keep-alive
          if (got == want) continue;
          if (got == POISON) {
            ++poison, ++per_c[c], ++per_bank[(cell % 16) / 4];
          } else {
            ++other;
          }
        }
      printf("%-38s %s: %9zu poisoned dwords of %zu (%.4f%%), other mismatches %zu; by dword [%zu %zu %zu %zu], by "
             "lane%%16/4 [%zu %zu %zu %zu]\n",
             v.name, beside[load], poison, cells * 4,
             100.0 * poison / (cells * 4.0), other, per_c[0], per_c[1], per_c[2], per_c[3], per_bank[0], per_bank[1],
             per_bank[2], per_bank[3]);
    }
  return 0;
}
