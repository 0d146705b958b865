// Weight gradient of a 2-D convolution with few output channels (DPCCN: 16 / 32; wesep/modules/dpccn/convs.py:28-110)
// on channels-last images, one pass over the activation:
//   slab[split][n][(ky*k + kx)*C + c] = sum over the split's output pixels m of dy[m][n] * x[pixel(m) + tap][c]
// The generic TN GEMM on the implicit patch matrix (gemm_bf16.hip) gives every 128-column slice of the k*k*C patch
// columns to a different workgroup, so the activation crosses L2 / HBM once per slice (9x for a 3x3 kernel on 80
// channels: 2.7 ms per launch, 3 % of the HBM roof, the dominant kernel of the DPCCN step).  Here ONE workgroup owns
// all k*k*C columns of its pixels: per tile of 32 consecutive output pixels it stages
//   dy tile   [n <= 32][32 pixels]      bf16 hi | lo   (MFMA A operand: D rows = output channels)
//   patches   [k*k*C][32 pixels]        bf16 hi | lo   (MFMA B operand: D columns = patch columns)
// in LDS -- each loaded float4 (4 channels of one pixel under one tap) goes through a 4 x 4 register transpose with the
// three neighbouring pixels, so the image is written as 8-byte groups of 4 consecutive pixels -- and the 8 waves split
// the patch columns in blocks of 32 (up to 3 accumulators per wave: k*k*C <= 768).  Products are split-bf16
// (3 MFMAs, fp32 accumulate) like every GEMM of the library; the next tile's loads fly under the current tile's MFMAs.
// Splits write slabs; the caller reduces them (deterministic, no atomics).
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#define CW_LD 40       // bf16 per LDS row (32 pixels + 8: 80 B, conflict-free 16-byte fragment reads)
#define CW_MAXK 768    // patch columns per workgroup
#define CW_ITEMS 3     // (pixel quad, tap, channel quad) items per thread and tile: 8 * CW_MAXK / 4 / 512

struct CwMeta {
  long long off;  // element offset of tap (0, 0) of the pixel's window (may lie before the image)
  long long goff; // row of dy
  int tapmask;    // bit t: tap t inside the image (0 for rows past the end of the split)
  int valid;
};

__device__ __forceinline__ void cw_split(float v, __bf16& hi, __bf16& lo) {
  hi = (__bf16)v;
  lo = (__bf16)(v - (float)hi);
}

__global__ __launch_bounds__(512, 1) void conv_wgrad_kernel(const ws_conv_wgrad_args p) {
  extern __shared__ __attribute__((aligned(16))) __bf16 lds[];  // [2 planes][32 + Kk rows][CW_LD]
  __shared__ CwMeta meta[2][32];
  __shared__ float bred[8][32];
  const ws_conv_view cv = p.conv;
  // patch columns [kbase, kbase + Kk) of the K_all = k*k*C: blockIdx.y walks chunks of CW_MAXK columns (the image is
  // then streamed once per chunk -- 2..4 times for the wide DPCCN layers instead of once per 128 columns)
  const int K_all = cv.k * cv.k * cv.C, kbase = blockIdx.y * CW_MAXK, Nn = p.Nn;
  const int Kk = min(CW_MAXK, K_all - kbase);
  const int nblk = (Kk + 31) / 32;            // 32-column blocks of the chunk; wave w owns blocks w, w + 8, w + 16
  const int rows = 32 + 32 * nblk;            // LDS rows per plane: dy tile first, then the patch columns (whole blocks)
  __bf16* const Ph = lds;
  __bf16* const Pl = lds + rows * CW_LD;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int split = blockIdx.x;
  const long long t_begin = (long long)split * p.tiles_per_split;
  const long long ntiles_all = ((long long)p.M + 31) / 32;
  const long long t_end = min(ntiles_all, t_begin + p.tiles_per_split);
  const int c4n = cv.C >> 2, nitems = 8 * (Kk >> 2);

  // this thread's items (the same for every tile): pixel quad, tap, channel quad -> element offset inside a window
  int it_q[CW_ITEMS], it_tap[CW_ITEMS], it_off[CW_ITEMS], it_kk[CW_ITEMS];
  bool it_on[CW_ITEMS];
#pragma unroll
  for (int i = 0; i < CW_ITEMS; ++i) {
    const int it = tid + 512 * i;
    it_on[i] = it < nitems;
    const int itc = it_on[i] ? it : 0;
    it_q[i] = itc & 7;
    const int rest = (itc >> 3) + (kbase >> 2);   // (tap, channel quad) flattened = patch column / 4
    const int c4 = rest % c4n, tap = rest / c4n;
    const int dl = cv.dil > 0 ? cv.dil : 1;
    const int ky = (tap / cv.k) * dl, kx = (tap % cv.k) * dl;
    it_tap[i] = tap;
    it_off[i] = (ky * cv.W + kx) * (cv.ldp > 0 ? cv.ldp : cv.C) + 4 * c4;
    it_kk[i] = tap * cv.C + 4 * c4 - kbase;       // column inside the chunk
  }
  // dy tile: thread (row = tid / 8, channel quad = tid % 8) of the first 256 threads
  const int g_row = (tid >> 3) & 31, g_c4 = tid & 7;
  const bool g_on = tid < 256 && 4 * g_c4 < Nn;

  auto make_meta = [&](long long tile, int slot) {
    if (tid < 32) {
      const long long m = tile * 32 + tid;
      CwMeta r;
      r.valid = tile < t_end && m < p.M;
      r.off = r.goff = 0;
      r.tapmask = 0;
      if (r.valid) {
        const int hw = cv.Ho * cv.Wo;
        const int rr = (int)(m / hw), q = (int)(m - (long long)rr * hw);
        const int ho = q / cv.Wo, wo = q - ho * cv.Wo;
        const int bh = ho * cv.sh - cv.p, bw = wo * cv.sw - cv.p, dl = cv.dil > 0 ? cv.dil : 1;
        const int cld = cv.ldp > 0 ? cv.ldp : cv.C;
        r.off = (long long)rr * cv.H * cv.W * cld + (long long)(bh * cv.W + bw) * cld;
        r.goff = m * p.ldg;
        int mask = 0;
        for (int ky = 0; ky < cv.k; ++ky)
          for (int kx = 0; kx < cv.k; ++kx)
            if ((unsigned)(bh + ky * dl) < (unsigned)cv.H && (unsigned)(bw + kx * dl) < (unsigned)cv.W)
              mask |= 1 << (ky * cv.k + kx);
        r.tapmask = mask;
      }
      meta[slot][tid] = r;
    }
  };

  f32x4 rx[CW_ITEMS][4], rg = {0.f, 0.f, 0.f, 0.f};
  unsigned okx[CW_ITEMS];
  bool okg = false;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  typedef const __attribute__((address_space(1))) f32x4* gf4;
  auto load_tile = [&](int slot) {  // unconditional loads (clamped to offset 0), masks applied at the store
#pragma unroll
    for (int i = 0; i < CW_ITEMS; ++i) {
      unsigned ok = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const CwMeta& r = meta[slot][4 * it_q[i] + j];
        const bool v = it_on[i] && ((r.tapmask >> it_tap[i]) & 1);
        ok |= v ? 1u << j : 0u;
        rx[i][j] = *(gf4)(p.X + (v ? r.off + it_off[i] : 0));
      }
      okx[i] = ok;
    }
    const CwMeta& r = meta[slot][g_row];
    okg = g_on && r.valid;
    rg = *(gf4)(p.G + (okg ? r.goff + 4 * g_c4 : 0));
  };
  float gsum[4] = {0.f, 0.f, 0.f, 0.f};
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < CW_ITEMS; ++i) {
      if (!it_on[i]) continue;  // uniform per (thread, i); no loads depend on it
      // 4 pixels x 4 channels -> per channel the 4 consecutive pixels as one 8-byte group
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        bf16x4 hi, lo;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float v = (okx[i] >> j) & 1u ? rx[i][j][c] : 0.f;
          __bf16 h, l;
          cw_split(v, h, l);
          hi[j] = h;
          lo[j] = l;
        }
        const int o = (32 + it_kk[i] + c) * CW_LD + 4 * it_q[i];
        *reinterpret_cast<bf16x4*>(Ph + o) = hi;
        *reinterpret_cast<bf16x4*>(Pl + o) = lo;
      }
    }
    if (tid < 256) {  // dy tile, transposed: rows = output channel, columns = pixel
      const f32x4 v = okg ? rg : zero4;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        __bf16 h, l;
        cw_split(v[c], h, l);
        gsum[c] += v[c];
        const int o = (4 * g_c4 + c) * CW_LD + g_row;
        Ph[o] = h;
        Pl[o] = l;
      }
    }
  };

  for (int i = tid; i < rows * CW_LD; i += 512) {  // rows no store ever touches (block padding) must not hold NaNs
    reinterpret_cast<uint32_t*>(lds)[i] = 0u;      // (2 planes x rows x CW_LD bf16 = rows x CW_LD words)
  }
  f32x16 acc[CW_ITEMS];
#pragma unroll
  for (int b = 0; b < CW_ITEMS; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;

  if (t_begin < t_end) {
    make_meta(t_begin, 0);
    __syncthreads();
    load_tile(0);
  }
  int it = 0;
  for (long long tile = t_begin; tile < t_end; ++tile, ++it) {
    make_meta(tile + 1, (it + 1) & 1);  // rows past the end are marked invalid: the prefetch below is unconditional
    __syncthreads();                    // previous tile's fragment reads done; next meta visible
    store_tile();
    __syncthreads();
    load_tile((it + 1) & 1);
    __builtin_amdgcn_sched_barrier(0);  // the loads are issued here and fly under the MFMAs
#pragma unroll
    for (int ks = 0; ks < 32; ks += 16) {
      const int ra = l31 * CW_LD + ks + 8 * half;
      const bf16x8 gh = *reinterpret_cast<const bf16x8*>(Ph + ra);
      const bf16x8 gl = *reinterpret_cast<const bf16x8*>(Pl + ra);
#pragma unroll
      for (int b = 0; b < CW_ITEMS; ++b) {
        const int blk = wave + 8 * b;
        if (blk < nblk) {  // uniform
          const int rb = (32 + 32 * blk + l31) * CW_LD + ks + 8 * half;
          const bf16x8 xh = *reinterpret_cast<const bf16x8*>(Ph + rb);
          const bf16x8 xl = *reinterpret_cast<const bf16x8*>(Pl + rb);
          acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gh, xh, acc[b], 0, 0, 0);
          acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gh, xl, acc[b], 0, 0, 0);
          acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gl, xh, acc[b], 0, 0, 0);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }

  float* out = p.slab + (long long)split * p.slab_stride;
#pragma unroll
  for (int b = 0; b < CW_ITEMS; ++b) {
    const int blk = wave + 8 * b;
    const int kk = 32 * blk + l31;
    if (blk < nblk && kk < Kk) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = (r & 3) + 8 * (r >> 2) + 4 * half;
        if (n < Nn) out[(long long)n * K_all + kbase + kk] = acc[b][r];
      }
    }
  }
  if (p.bslab && blockIdx.y == 0) {  // db[n] = sum over pixels of dy[m][n]: rows (tid / 8) of each channel quad through LDS
    __syncthreads();
    if (tid < 256) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float t = gsum[c];
        t += __shfl_xor(t, 8, 64);
        t += __shfl_xor(t, 16, 64);
        t += __shfl_xor(t, 32, 64);
        if ((lane >> 3) == 0) bred[wave][4 * g_c4 + c] = t;  // waves 0..3 hold 8 rows each
      }
    }
    __syncthreads();
    if (tid < Nn) p.bslab[(long long)split * p.bslab_stride + tid] = bred[0][tid] + bred[1][tid] + bred[2][tid] + bred[3][tid];
  }
}

extern "C" int ws_conv_wgrad(const ws_conv_wgrad_args* a, void* stream) {
  WS_REQUIRE(a && a->G && a->X && a->slab, "ws_conv_wgrad: null pointer");
  const ws_conv_view& c = a->conv;
  WS_REQUIRE(c.mode == 0 && c.H > 0 && c.W > 0 && c.C > 0 && c.C % 4 == 0 && c.Ho > 0 && c.Wo > 0 && c.k >= 1 && c.k <= 5 &&
                 c.sh >= 1 && c.sw >= 1 && c.p >= 0,
             "ws_conv_wgrad: bad conv view (mode 0, C %% 4, k <= 5)");
  const int Kk = c.k * c.k * c.C;
  WS_REQUIRE(a->Nn > 0 && a->Nn <= 32 && a->Nn % 4 == 0 && a->ldg >= a->Nn && a->ldg % 4 == 0,
             "ws_conv_wgrad: Nn in 4..32 step 4, ldg %% 4 (Nn=%d ldg=%lld)", a->Nn, (long long)a->ldg);
  WS_REQUIRE(a->M > 0 && a->M % (c.Ho * c.Wo) == 0 && a->nsplit > 0 && a->tiles_per_split > 0 &&
                 (long long)a->nsplit * a->tiles_per_split * 32 >= a->M,
             "ws_conv_wgrad: splits do not cover M");
  WS_REQUIRE(c.ldp == 0 || (c.ldp >= c.C && c.ldp % 4 == 0), "ws_conv_wgrad: pixel stride ldp >= C, %% 4 (got %d)", c.ldp);
  WS_REQUIRE((long long)(c.H + 2 * c.k) * c.W * (c.ldp > 0 ? c.ldp : c.C) < (1LL << 31),
             "ws_conv_wgrad: one image below 2^31 elements");
  const int kchunk = Kk < CW_MAXK ? Kk : CW_MAXK, nchunks = (Kk + CW_MAXK - 1) / CW_MAXK;
  const size_t lds_bytes = (size_t)2 * (32 + 32 * ((kchunk + 31) / 32)) * CW_LD * sizeof(__bf16);
  static bool attr_set = false;
  if (!attr_set) {
    // (a failure here -- no device -- shows up as the launch error below, not as an argument error)
    attr_set = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (32 + CW_MAXK) * CW_LD * 2) == hipSuccess;
    (void)hipGetLastError();
  }
  hipStream_t s = (hipStream_t)stream;
  ws_prof_begin(WS_PROF_GEMM_TN, s);
  hipLaunchKernelGGL(conv_wgrad_kernel, dim3(a->nsplit, nchunks), dim3(512), lds_bytes, s, *a);
  ws_prof_end(WS_PROF_GEMM_TN, s);
  return ws_check_launch("ws_conv_wgrad");
}
