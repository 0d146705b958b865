// 3 x 3, stride-1, "same" convolution on a channels-last image with an LDS halo tile, and its weight gradient (round 3).
//
// DPCCN spends two thirds of its step in the 3 x 3 convolutions of its dense blocks (convs.py:80-112: five layers of
// 16 or 32 output channels on a growing feature map of up to 80 / 160 channels, at up to 2 M pixels).  The implicit-GEMM
// path (gemm_bf16.hip with ws_conv_view) forms every output pixel's k*k*C patch row from global memory: each input pixel
// crosses the L2 -> CU path NINE times, and with 16 .. 32 output columns there is next to no arithmetic to hide it behind.
//
// ws_conv3x3 (forward, and the input gradient with flipped weights).  A workgroup (4 waves) owns a tile of 32 rows x 4P
// columns of output pixels of one image (wave = P adjacent columns, lane = row: DPCCN's grids are 251 frames x 2^k + 1
// bins, so tiles that are long in h and narrow in w waste a few percent of their slots where 128-pixel row segments
// wasted 33-87 %).  For each chunk of 16 input channels it stages the 34 x (4P + 2) pixel halo ONCE in LDS as bf16
// hi / lo planes ([column][row][channel], 48-byte pixel stride: conflict-free 16-byte fragment reads); a pixel fragment
// (one halo column under one ky) then serves the three output columns it is a tap of, and a weight fragment (read from
// the L1-resident packed weights) serves the wave's P columns: per 3 MFMAs the wave reads ~0.5 KB from LDS and ~0.17 KB
// from L1 (v2, one column per wave and nothing shared: 2 KB + 2 KB -- it ran at the L1 rate, no faster than the
// implicit GEMM).
//
//   Y[m][n] = bias[n] + R[m][n] + sum_{ky, kx, c} X[pixel(m) + (ky - 1, kx - 1)][c] * W[n][(ky * 3 + kx) * Cin + c]
//
// m = (b * H + h) * Wd + w; X has pixel stride ldx (>= Cin: the image may be the first Cin columns of a wider tensor),
// Y and R have row stride ldy (the output may be the first Cout columns of a wider tensor, and R may alias Y: every
// element is read by the thread that writes it).  Products are split-bf16 (3 MFMAs), fp32 accumulation, like every
// other GEMM of the library.  The same kernel is the input gradient of such a convolution: X = dY, W = the flipped,
// channel-swapped weights, Y = R = the gradient buffer it accumulates into.
// Orientation: D[m = output channel][n = pixel] -- the weights are the A operand, PRE-PACKED by the host into MFMA
// fragment order as bf16 hi / lo (dev.conv3x3_pack: unit (((chunk*9 + tap)*NTP + t)*2 + part)*64 + lane holds the 8
// channels 16 chunk + 8 (lane >> 5) + j of row t*32 + (lane & 31)); the pixels are the B operand (from LDS).  A lane
// then holds four consecutive output channels of its pixel per register quad and stores them as 16-byte pieces.  More
// than 64 output channels are cut into groups of 64 (blockIdx.z), each re-staging the (then narrow: that is the input
// gradient of a 16 / 32-channel layer) halo.
//
// ws_conv3x3_wgrad (weight gradient).  K = pixels: both MFMA operands want 8 CONSECUTIVE pixels per lane, and a tap
// shifts one operand against the other.  The tile is 30 rows x 4 columns (wave = column); k runs over the 32 halo rows
// of a column.  X is staged transposed ([halo column][channel][row], 80-byte rows) once per tile and needs no shift for
// ky (kx picks the halo column); dY is staged transposed once ([column][channel][8 zeros | 30 rows | 2 zeros]) and the
// ky = 1 / 2 fragments are the aligned 16-byte block and its predecessor funnel-shifted by one / two elements in
// registers (v_alignbit / a register rename).  Nine accumulators (taps) of [32 output channels][32 input channels] per
// wave; a workgroup owns one 32-channel chunk of the input and a range of tiles (a "split"), reduces its four waves
// through LDS and writes one slab -- the caller sums the slabs (deterministic, no atomics).  conv_wgrad.hip, which this
// replaces for 3 x 3 / stride 1, staged every tap's patch separately: nine splits and transposes per input element.
#include <mutex>

#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define C3_TH 32                  // output rows per workgroup (= MFMA N: lane = row)
#define C3_CC 16                  // input channels per LDS chunk (one MFMA k-step)
#define C3_PS 24                  // bf16 per pixel in an LDS plane (16 + 8: 48 B)
#define C3_COL ((C3_TH + 2) * C3_PS)          // one halo column: 34 pixels

__device__ __forceinline__ void c3_split4(const f32x4 v, bf16x4& hi, bf16x4& lo) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    hi[j] = (__bf16)v[j];
    lo[j] = (__bf16)(v[j] - (float)hi[j]);
  }
}

template <int NT, int P, bool PF>  // 32-channel tiles of output channels per workgroup; output columns per wave; prefetch
__global__ __launch_bounds__(256, 2) void conv3x3_kernel(const ws_conv3x3_args p) {
  constexpr int NC = 4 * P + 2, PLANE = NC * C3_COL;                  // halo columns; bf16 per plane
  constexpr int NITEM = (C3_TH + 2) * NC * (C3_CC / 4), NI = (NITEM + 255) / 256;
  __shared__ __attribute__((aligned(16))) __bf16 halo[2][PLANE];      // [plane hi / lo][column][row][channel]  P = 4: 58.8 KB
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = p.H, Wd = p.Wd, Cin = p.Cin, Cout = p.Cout;
  const int ntt = (Cout + 31) / 32, ntp = ntt <= 2 ? ntt : (ntt + 1) & ~1, ng = ntp / NT;
  const int w0 = blockIdx.x * 4 * P, h0 = blockIdx.y * C3_TH, b = blockIdx.z / ng, tg = (blockIdx.z - b * ng) * NT;
  const long long img = (long long)b * H * Wd;

  f32x16 acc[P][NT];
#pragma unroll
  for (int j = 0; j < P; ++j)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][t][r] = 0.f;
  const bf16x8* wpk = reinterpret_cast<const bf16x8*>(p.W);

  // halo item i = tid + 256 k: channel quad i & 3 of halo pixel i >> 2 (column fastest: consecutive threads walk channels, then w)
  f32x4 pre[NI];
  auto halo_load = [&](int c0) {
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      const int i = tid + 256 * k, q = i & 3, pix = i >> 2, col = pix % NC, row = pix / NC;
      const int hh = h0 + row - 1, ww = w0 + col - 1, c = c0 + 4 * q;
      pre[k] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (i < NITEM && (unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)Wd && c < Cin)
        pre[k] = *reinterpret_cast<const f32x4*>(p.X + (img + (long long)hh * Wd + ww) * p.ldx + c);
    }
  };
  auto halo_store = [&]() {       // zeros outside the image and beyond Cin
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      const int i = tid + 256 * k, q = i & 3, pix = i >> 2, col = pix % NC, row = pix / NC;
      if (i < NITEM) {
        bf16x4 hi, lo;
        c3_split4(pre[k], hi, lo);
        const int o = col * C3_COL + row * C3_PS + 4 * q;
        *reinterpret_cast<bf16x4*>(&halo[0][o]) = hi;
        *reinterpret_cast<bf16x4*>(&halo[1][o]) = lo;
      }
    }
  };

  if (PF) halo_load(0);
  for (int c0 = 0; c0 < Cin; c0 += C3_CC) {
    if (!PF) halo_load(c0);
    halo_store();
    __syncthreads();
    if (PF && c0 + C3_CC < Cin) {
      halo_load(c0 + C3_CC);                      // the next chunk's loads fly under this chunk's MFMAs
      __builtin_amdgcn_sched_barrier(0);
    }
    const long long u0 = (long long)(c0 / C3_CC) * 9 * ntp;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      bf16x8 ah[3][NT], al[3][NT];               // this ky's weight fragments: 3 taps x NT tiles, hi / lo
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const long long u = ((u0 + (ky * 3 + kx) * ntp + tg + t) * 2) * 64 + lane;
          ah[kx][t] = wpk[u];
          al[kx][t] = wpk[u + 64];
        }
#pragma unroll
      for (int jc = 0; jc < P + 2; ++jc) {       // halo column jc of the wave: tap kx of its output column jc - kx
        const __bf16* bp = &halo[0][(wv * P + jc) * C3_COL + (l31 + ky) * C3_PS + 8 * half];
        const bf16x8 bh = *reinterpret_cast<const bf16x8*>(bp);
        const bf16x8 bl = *reinterpret_cast<const bf16x8*>(bp + PLANE);
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int j = jc - kx;
          if (j < 0 || j >= P) continue;           // compile time
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[j][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[kx][t], bh, acc[j][t], 0, 0, 0);
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[j][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[kx][t], bh, acc[j][t], 0, 0, 0);
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[j][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[kx][t], bl, acc[j][t], 0, 0, 0);
        }
      }
    }
    __syncthreads();   // the next chunk overwrites the halo
  }
  // ---- epilogue: D[m = channel][n = pixel]; register r of a lane: channel (r & 3) + 8 (r >> 2) + 4 half of its tile ----
  const int h = h0 + l31;
#pragma unroll
  for (int j = 0; j < P; ++j) {
    const int w = w0 + wv * P + j;
    if (w < Wd && h < H) {
      float* yrow = p.Y + (img + (long long)h * Wd + w) * p.ldy;
      const float* rrow = p.R ? p.R + (img + (long long)h * Wd + w) * p.ldy : nullptr;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = (tg + t) * 32 + 8 * g + 4 * half;
          if (n < Cout) {                           // Cout % 4 == 0: a quad is inside or outside as a whole
            f32x4 v = {acc[j][t][4 * g], acc[j][t][4 * g + 1], acc[j][t][4 * g + 2], acc[j][t][4 * g + 3]};
            if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
            if (rrow) v += *reinterpret_cast<const f32x4*>(rrow + n);
            *reinterpret_cast<f32x4*>(yrow + n) = v;
          }
        }
    }
  }
}

template <int NT, int P, bool PF>
static void c3_launch(const ws_conv3x3_args* a, hipStream_t s) {
  const int ntt = (a->Cout + 31) / 32, ntp = ntt <= 2 ? ntt : (ntt + 1) & ~1, ng = ntp / NT;
  const dim3 grid((a->Wd + 4 * P - 1) / (4 * P), (a->H + C3_TH - 1) / C3_TH, a->B * ng), block(256);
  hipLaunchKernelGGL((conv3x3_kernel<NT, P, PF>), grid, block, 0, s, *a);
}

extern "C" int ws_conv3x3(const ws_conv3x3_args* a, void* stream) {
  WS_REQUIRE(a && a->X && a->W && a->Y, "ws_conv3x3: null pointer");
  WS_REQUIRE(a->B > 0 && a->H > 0 && a->Wd > 0 && a->Cin > 0 && a->Cin % 4 == 0 && a->Cout > 0 && a->Cout % 4 == 0 &&
                 a->Cout <= 1024,
             "ws_conv3x3: Cin %% 4, Cout %% 4, Cout <= 1024 (got %d, %d)", a->Cin, a->Cout);
  WS_REQUIRE(a->ldx >= a->Cin && a->ldx % 4 == 0 && a->ldy >= a->Cout && a->ldy % 4 == 0,
             "ws_conv3x3: leading dimensions (ldx >= Cin, ldy >= Cout, both %% 4)");
  WS_REQUIRE(a->H <= 65535 * C3_TH && (long long)a->B * ((a->Cout + 63) / 64) <= 65535,
             "ws_conv3x3: H / 32 and B * ceil(Cout / 64) index the launch grid (<= 65535)");
  hipStream_t s = (hipStream_t)stream;
  static const int variant = [] { const char* e = getenv("WS_CONV3X3_VARIANT"); return e ? atoi(e) : 0; }();   // experiments
  ws_prof_begin(WS_PROF_GEMM_NT, s);
  const bool wide = a->Wd >= 100;                 // 16-column tiles where they fill; 8-column tiles on the small grids
  const bool pf = !(variant & 1) && a->Cin > C3_CC;
  if (a->Cout <= 32) {
    if (wide) { if (pf) c3_launch<1, 4, true>(a, s); else c3_launch<1, 4, false>(a, s); }
    else      { if (pf) c3_launch<1, 2, true>(a, s); else c3_launch<1, 2, false>(a, s); }
  } else {
    // two channel tiles x four columns leave no registers for the prefetch
    if (wide && !(variant & 2)) c3_launch<2, 4, false>(a, s);
    else if (pf) c3_launch<2, 2, true>(a, s);
    else c3_launch<2, 2, false>(a, s);
  }
  ws_prof_end(WS_PROF_GEMM_NT, s);
  return ws_check_launch("ws_conv3x3");
}

// ------------------------------------------------------------------------------------------------------------------
// weight pack (round 6): the fragment order above, written by ONE launch from up to WS_C3_NSRC strided views of weight tensors
// (dev.conv3x3_pack composed it from ~9 ATen launches -- zeros, slice copy, two casts, a subtraction, a cast, stack, permute --
// per layer and pass: ~1 400 of DPCCN's launches per step).  Row n of the logical W[n][tap][c], column c in source k's range:
//   W = src_k.w[n * s_row + (c - col_off) * s_col + (flip ? 8 - tap : tap) * s_tap]
// forward of a layer: one source, w[co][ci][3][3] -> (s_row, s_col, s_tap) = (9 Ci, 9, 1); input gradient of a channel block
// of a dense block: sources = the later layers' weights, rows = the block's input channels, columns = their output channels
// side by side, taps flipped.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv3x3_pack_kernel(const ws_conv3x3_pack_args p) {
  const int ntt = (p.Cout + 31) / 32, ntp = ntt <= 2 ? ntt : ntt + (ntt & 1), nch = (p.Cin + C3_CC - 1) / C3_CC;
  const long long total = (long long)nch * 9 * ntp * 64 * 8;      // elements per part
  __bf16* out = reinterpret_cast<__bf16*>(p.out);
  for (long long idx = blockIdx.x * 256LL + threadIdx.x; idx < total; idx += gridDim.x * 256LL) {
    long long r = idx;
    const int j = r & 7; r >>= 3;
    const int lane = r & 63; r >>= 6;
    const int t = (int)(r % ntp); r /= ntp;
    const int tap = (int)(r % 9);
    const int chunk = (int)(r / 9);
    const int n = t * 32 + (lane & 31), c = chunk * C3_CC + 8 * (lane >> 5) + j;
    float v = 0.f;
    if (n < p.Cout && c < p.Cin) {
#pragma unroll
      for (int k = 0; k < WS_C3_NSRC; ++k)
        if (k < p.nsrc && c >= p.src[k].col_off && c < p.src[k].col_off + p.src[k].cols)
          v = p.src[k].w[(long long)n * p.src[k].s_row + (long long)(c - p.src[k].col_off) * p.src[k].s_col +
                         (long long)(p.flip ? 8 - tap : tap) * p.src[k].s_tap];
    }
    const __bf16 hi = (__bf16)v;
    const long long unit = (((long long)(chunk * 9 + tap) * ntp + t) * 2) * 64 + lane;
    out[unit * 8 + j] = hi;
    out[(unit + 64) * 8 + j] = (__bf16)(v - (float)hi);
  }
}

extern "C" int ws_conv3x3_pack(const ws_conv3x3_pack_args* a, void* stream) {
  WS_REQUIRE(a && a->out && a->Cin > 0 && a->Cout > 0 && a->nsrc > 0 && a->nsrc <= WS_C3_NSRC, "ws_conv3x3_pack: bad arguments");
  for (int k = 0; k < a->nsrc; ++k)
    WS_REQUIRE(a->src[k].w && a->src[k].cols > 0 && a->src[k].col_off >= 0 && a->src[k].col_off + a->src[k].cols <= a->Cin,
               "ws_conv3x3_pack: source %d does not lie inside the %d columns", k, a->Cin);
  const int ntt = (a->Cout + 31) / 32, ntp = ntt <= 2 ? ntt : ntt + (ntt & 1), nch = (a->Cin + C3_CC - 1) / C3_CC;
  const long long total = (long long)nch * 9 * ntp * 64 * 8;
  const int grid = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
  hipLaunchKernelGGL(conv3x3_pack_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, *a);
  return ws_check_launch("ws_conv3x3_pack");
}

// ------------------------------------------------------------------------------------------------------------------
// weight gradient
// ------------------------------------------------------------------------------------------------------------------
#define W3_TH 30                  // output rows per tile: their 3 x 3 windows span 32 halo rows = the MFMA K of two steps
#define W3_LD 40                  // bf16 per LDS row (32 + 8: 80 B, conflict-free 16-byte fragment reads)
#define W3_AB (4 * 32 * W3_LD)    // dY plane: [column 4][output channel 32][8 zeros | rows 0..29 | 2 zeros]

__device__ __forceinline__ bf16x8 w3_frag(const u32x4 v) { return __builtin_bit_cast(bf16x8, v); }

template <int SW>                 // stride along w (1 or 2): halo column = SW * output column + kx
__global__ __launch_bounds__(256, 2) void conv3x3_wgrad_kernel(const ws_conv3x3_wgrad_args p) {
  constexpr int NCOL = 3 * SW + 3;             // halo columns of 4 output columns: 6 / 9
  constexpr int XB = NCOL * 32 * W3_LD;        // X plane: [halo column][channel 32][row]
  constexpr int NXITEM = 8 * NCOL * 8, NXI = (NXITEM + 255) / 256;   // (row quad, halo column, channel quad)
  extern __shared__ __attribute__((aligned(16))) __bf16 lds[];       // 2 XB + 2 W3_AB bf16: 51.2 / 66.6 KB, two workgroups per CU
  __bf16* const xb = lds;                      // planes at xb, xb + XB
  __bf16* const ab = lds + 2 * XB;             // planes at ab, ab + W3_AB
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = p.H, Wd = p.Wd, Wx = p.Wx, Cin = p.Cin;
  const int split = blockIdx.x, c0 = blockIdx.y * 32, n0 = blockIdx.z * 32;
  const int nn = min(32, p.Nn - n0);
  const int ncg = (Wd + 3) / 4, nrt = (H + W3_TH - 1) / W3_TH;
  const long long ntiles = (long long)p.B * nrt * ncg;
  const long long t_begin = (long long)split * p.tiles_per_split, t_end = min(ntiles, t_begin + p.tiles_per_split);

  for (int i = tid; i < (2 * XB + 2 * W3_AB) / 2; i += 256) reinterpret_cast<uint32_t*>(lds)[i] = 0u;   // the pads stay zero

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float gsum[4] = {0.f, 0.f, 0.f, 0.f};
  const bool want_bias = p.bslab && blockIdx.y == 0;

  // staging items: X (8 row quads x NCOL halo columns x 8 channel quads), dY (8 row quads x 4 columns x 8 channel quads = 256)
  int x_cq[NXI], x_hc[NXI], x_hq[NXI];
#pragma unroll
  for (int i = 0; i < NXI; ++i) {
    // lane = (channel quad, row quad), halo column per wave / item: the eight channel quads of a pixel row are one 128-byte
    // global run, and the transposing 8-byte LDS stores of a wave land on bank pairs 16 (cq & 3) + 2 hq (+ 20 e): 32 distinct
    // pairs, two-way instead of the eight-way conflicts of the (cq, column, row quad) order of round 3 (round 6)
    const int it = tid + 256 * i;
    x_cq[i] = it & 7;
    x_hq[i] = (it >> 3) & 7;
    x_hc[i] = it >> 6;
  }
  const int g_nq = tid & 7, g_hq = (tid >> 3) & 7, g_col = tid >> 6;   // (wave = column: see the X items)

  for (long long tile = t_begin; tile < t_end; ++tile) {
    const int cg = (int)(tile % ncg), rt = (int)((tile / ncg) % nrt), b = (int)(tile / ((long long)ncg * nrt));
    const int h0 = rt * W3_TH, w0 = cg * 4;
    const long long img = (long long)b * H * Wd, imgx = (long long)b * H * Wx;
    __syncthreads();                             // the previous tile's fragment reads (and the zero fill) are done
    // ---- X halo, transposed: 4 consecutive rows of one channel become one 8-byte group ----
#pragma unroll
    for (int i = 0; i < NXI; ++i) {
      if (tid + 256 * i >= NXITEM) break;
      const int ww = SW * w0 - 1 + x_hc[i], c = c0 + 4 * x_cq[i];
      f32x4 v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int hh = h0 - 1 + 4 * x_hq[i] + j;
        v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if ((unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)Wx && c < Cin)
          v[j] = *reinterpret_cast<const f32x4*>(p.X + (imgx + (long long)hh * Wx + ww) * p.ldx + c);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        bf16x4 hi, lo;
        c3_split4(f32x4{v[0][e], v[1][e], v[2][e], v[3][e]}, hi, lo);
        const int o = (x_hc[i] * 32 + 4 * x_cq[i] + e) * W3_LD + 4 * x_hq[i];
        *reinterpret_cast<bf16x4*>(xb + o) = hi;
        *reinterpret_cast<bf16x4*>(xb + XB + o) = lo;
      }
    }
    // ---- dY tile, transposed, behind 8 zeros ----
    {
      const int ww = w0 + g_col, n = 4 * g_nq;
      f32x4 v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int ii = 4 * g_hq + j, hh = h0 + ii;
        v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (ii < W3_TH && hh < H && ww < Wd && n < nn)
          v[j] = *reinterpret_cast<const f32x4*>(p.G + (img + (long long)hh * Wd + ww) * p.ldg + n0 + n);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        bf16x4 hi, lo;
        c3_split4(f32x4{v[0][e], v[1][e], v[2][e], v[3][e]}, hi, lo);
        gsum[e] += (v[0][e] + v[1][e]) + (v[2][e] + v[3][e]);
        const int o = (g_col * 32 + n + e) * W3_LD + 8 + 4 * g_hq;
        *reinterpret_cast<bf16x4*>(ab + o) = hi;
        *reinterpret_cast<bf16x4*>(ab + W3_AB + o) = lo;
      }
    }
    __syncthreads();
    // ---- 2 k-steps x 3 kx x 3 ky ----
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int ao = (wv * 32 + l31) * W3_LD + 8 + 16 * ks + 8 * half;
      u32x4 a_h[3], a_l[3];
      {
        const u32x4 ch = *reinterpret_cast<const u32x4*>(ab + ao), ph = *reinterpret_cast<const u32x4*>(ab + ao - 8);
        const u32x4 cl = *reinterpret_cast<const u32x4*>(ab + W3_AB + ao), pl = *reinterpret_cast<const u32x4*>(ab + W3_AB + ao - 8);
        a_h[0] = ch;                             // k pairs X row h0 - 1 + k with dY row k - ky of the tile
        a_l[0] = cl;
        a_h[1] = u32x4{__builtin_amdgcn_alignbit(ch.x, ph.w, 16), __builtin_amdgcn_alignbit(ch.y, ch.x, 16),
                       __builtin_amdgcn_alignbit(ch.z, ch.y, 16), __builtin_amdgcn_alignbit(ch.w, ch.z, 16)};
        a_l[1] = u32x4{__builtin_amdgcn_alignbit(cl.x, pl.w, 16), __builtin_amdgcn_alignbit(cl.y, cl.x, 16),
                       __builtin_amdgcn_alignbit(cl.z, cl.y, 16), __builtin_amdgcn_alignbit(cl.w, cl.z, 16)};
        a_h[2] = u32x4{ph.w, ch.x, ch.y, ch.z};
        a_l[2] = u32x4{pl.w, cl.x, cl.y, cl.z};
      }
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int bo = ((SW * wv + kx) * 32 + l31) * W3_LD + 16 * ks + 8 * half;
        const bf16x8 bh = *reinterpret_cast<const bf16x8*>(xb + bo);
        const bf16x8 bl = *reinterpret_cast<const bf16x8*>(xb + XB + bo);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w3_frag(a_h[ky]), bh, acc[ky * 3 + kx], 0, 0, 0);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w3_frag(a_l[ky]), bh, acc[ky * 3 + kx], 0, 0, 0);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w3_frag(a_h[ky]), bl, acc[ky * 3 + kx], 0, 0, 0);
      }
    }
  }

  // ---- the four waves' partial sums -> wave 0, three taps per round through LDS (9216 floats) ----
  float* const red = reinterpret_cast<float*>(lds);
#pragma unroll
  for (int rr = 0; rr < 3; ++rr) {
    __syncthreads();
    if (wv > 0) {
#pragma unroll
      for (int tp = 0; tp < 3; ++tp)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((wv - 1) * 3 + tp) * 1024 + r * 64 + lane] = acc[3 * rr + tp][r];
    }
    __syncthreads();
    if (wv == 0) {
#pragma unroll
      for (int tp = 0; tp < 3; ++tp)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          acc[3 * rr + tp][r] += (red[tp * 1024 + r * 64 + lane] + red[(3 + tp) * 1024 + r * 64 + lane]) + red[(6 + tp) * 1024 + r * 64 + lane];
    }
  }
  const int K_all = 9 * Cin;
  if (wv == 0 && c0 + l31 < Cin) {
    float* out = p.slab + (long long)split * p.slab_stride;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = (r & 3) + 8 * (r >> 2) + 4 * half;
        if (n < nn) out[(long long)(n0 + n) * K_all + tap * Cin + c0 + l31] = acc[tap][r];
      }
  }
  if (want_bias) {                               // db[n] = sum over the split's pixels of dy[.][n]
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e) red[tid * 4 + e] = gsum[e];
    __syncthreads();
    if (tid < nn) {
      float t = 0.f;
      for (int u = 0; u < 32; ++u) t += red[(u * 8 + (tid >> 2)) * 4 + (tid & 3)];
      p.bslab[(long long)split * p.bslab_stride + n0 + tid] = t;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// weight gradient for <= 16 output channels (round 6): DPCCN's dense blocks (convs.py:80-112, 16 output channels on 16 .. 80
// inputs) were 20 of its 28 ms of weight gradients per step at config 3, with half of every 32-row tile of the kernel above
// empty and no load in flight while a tile is multiplied (nine 32 x 32 accumulators leave no registers for it).  Same tiles
// (30 rows x 4 columns, wave = column), same LDS images and the same funnel-shifted dY fragments, on v_mfma_f32_16x16x32_bf16:
// the 32 halo rows are ONE k-step, A = dY^T [16 output channels][32 rows], B = X^T [16 input channels][32 rows] for the two
// halves of the 32-channel chunk -- 54 MFMAs of half the size per tile; the eighteen 16 x 16 accumulators take 72 registers,
// and the NEXT tile's global loads are requested before the MFMAs of this one (48 registers).
// ------------------------------------------------------------------------------------------------------------------
#define W3_AB16 (4 * 16 * W3_LD)  // dY plane: [column 4][output channel 16][8 zeros | rows 0..29 | 2 zeros]

template <int SW>
__global__ __launch_bounds__(256, 2) void conv3x3_wgrad16_kernel(const ws_conv3x3_wgrad_args p) {
  constexpr int NCOL = 3 * SW + 3;             // halo columns of 4 output columns: 6 / 9
  constexpr int XB = NCOL * 32 * W3_LD;        // X plane: [halo column][channel 32][row]
  constexpr int NXITEM = 8 * NCOL * 8, NXI = (NXITEM + 255) / 256;   // (row quad, halo column, channel quad)
  extern __shared__ __attribute__((aligned(16))) __bf16 lds[];       // 2 XB + 2 W3_AB16 bf16: 41 / 56.3 KB, two workgroups per CU
  __bf16* const xb = lds;                      // planes at xb, xb + XB
  __bf16* const ab = lds + 2 * XB;             // planes at ab, ab + W3_AB16
  const int tid = threadIdx.x, lane = tid & 63, n16 = lane & 15, q4 = lane >> 4;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = p.H, Wd = p.Wd, Wx = p.Wx, Cin = p.Cin;
  const int split = blockIdx.x, c0 = blockIdx.y * 32;
  const int nn = p.Nn;                         // <= 16: one output tile
  const int ncg = (Wd + 3) / 4, nrt = (H + W3_TH - 1) / W3_TH;
  const long long ntiles = (long long)p.B * nrt * ncg;
  const long long t_begin = (long long)split * p.tiles_per_split, t_end = min(ntiles, t_begin + p.tiles_per_split);

  for (int i = tid; i < (2 * XB + 2 * W3_AB16) / 2; i += 256) reinterpret_cast<uint32_t*>(lds)[i] = 0u;   // the pads stay zero

  f32x4 acc[9][2];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int h = 0; h < 2; ++h) acc[t][h] = f32x4{0.f, 0.f, 0.f, 0.f};
  float gsum[4] = {0.f, 0.f, 0.f, 0.f};
  const bool want_bias = p.bslab && blockIdx.y == 0;

  // staging items: X (8 row quads x NCOL halo columns x 8 channel quads), dY (8 row quads x 4 columns x 4 channel quads = 128:
  // the first two waves)
  int x_cq[NXI], x_hc[NXI], x_hq[NXI];
#pragma unroll
  for (int i = 0; i < NXI; ++i) {
    // lane = (channel quad, row quad), halo column per wave / item: the eight channel quads of a pixel row are one 128-byte
    // global run, and the transposing 8-byte LDS stores of a wave land on bank pairs 16 (cq & 3) + 2 hq (+ 20 e): 32 distinct
    // pairs, two-way instead of the eight-way conflicts of the (cq, column, row quad) order of round 3 (round 6)
    const int it = tid + 256 * i;
    x_cq[i] = it & 7;
    x_hq[i] = (it >> 3) & 7;
    x_hc[i] = it >> 6;
  }
  const int g_nq = tid & 3, g_hq = (tid >> 2) & 7, g_col = (tid >> 5) & 3;   // (32 lanes = one column: conflict-free stores)
  const bool g_on = tid < 128;

  f32x4 xv[NXI][4], gv[4];
  auto load_tile = [&](long long tile) {         // global -> registers (zeros outside the image / the channel ranges)
    const int cg = (int)(tile % ncg), rt = (int)((tile / ncg) % nrt), b = (int)(tile / ((long long)ncg * nrt));
    const int h0 = rt * W3_TH, w0 = cg * 4;
    const long long img = (long long)b * H * Wd, imgx = (long long)b * H * Wx;
#pragma unroll
    for (int i = 0; i < NXI; ++i) {
      const int ww = SW * w0 - 1 + x_hc[i], c = c0 + 4 * x_cq[i];
      const bool on = tid + 256 * i < NXITEM;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int hh = h0 - 1 + 4 * x_hq[i] + j;
        xv[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (on && (unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)Wx && c < Cin)
          xv[i][j] = *reinterpret_cast<const f32x4*>(p.X + (imgx + (long long)hh * Wx + ww) * p.ldx + c);
      }
    }
    const int ww = w0 + g_col, n = 4 * g_nq;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ii = 4 * g_hq + j, hh = h0 + ii;
      gv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (g_on && ii < W3_TH && hh < H && ww < Wd && n < nn)
        gv[j] = *reinterpret_cast<const f32x4*>(p.G + (img + (long long)hh * Wd + ww) * p.ldg + n);
    }
  };
  auto stage_tile = [&]() {                      // registers -> the transposed bf16 hi / lo images
#pragma unroll
    for (int i = 0; i < NXI; ++i) {
      if (tid + 256 * i >= NXITEM) break;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        bf16x4 hi, lo;
        c3_split4(f32x4{xv[i][0][e], xv[i][1][e], xv[i][2][e], xv[i][3][e]}, hi, lo);
        const int o = (x_hc[i] * 32 + 4 * x_cq[i] + e) * W3_LD + 4 * x_hq[i];
        *reinterpret_cast<bf16x4*>(xb + o) = hi;
        *reinterpret_cast<bf16x4*>(xb + XB + o) = lo;
      }
    }
    if (g_on) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        bf16x4 hi, lo;
        c3_split4(f32x4{gv[0][e], gv[1][e], gv[2][e], gv[3][e]}, hi, lo);
        gsum[e] += (gv[0][e] + gv[1][e]) + (gv[2][e] + gv[3][e]);
        const int o = (g_col * 16 + 4 * g_nq + e) * W3_LD + 8 + 4 * g_hq;
        *reinterpret_cast<bf16x4*>(ab + o) = hi;
        *reinterpret_cast<bf16x4*>(ab + W3_AB16 + o) = lo;
      }
    }
  };

  if (t_begin < t_end) load_tile(t_begin);
  for (long long tile = t_begin; tile < t_end; ++tile) {
    __syncthreads();                             // the previous tile's fragment reads (and the zero fill) are done
    stage_tile();
    if (tile + 1 < t_end) load_tile(tile + 1);   // in flight under this tile's MFMAs
    __syncthreads();
    // ---- 3 kx x 3 ky x 2 halves of the channel chunk, ONE k-step of 32 halo rows: lane = (channel n16, row block q4) ----
    const int ao = (wv * 16 + n16) * W3_LD + 8 + 8 * q4;
    u32x4 a_h[3], a_l[3];
    {
      const u32x4 ch = *reinterpret_cast<const u32x4*>(ab + ao), ph = *reinterpret_cast<const u32x4*>(ab + ao - 8);
      const u32x4 cl = *reinterpret_cast<const u32x4*>(ab + W3_AB16 + ao), pl = *reinterpret_cast<const u32x4*>(ab + W3_AB16 + ao - 8);
      a_h[0] = ch;                               // k pairs X row h0 - 1 + k with dY row k - ky of the tile
      a_l[0] = cl;
      a_h[1] = u32x4{__builtin_amdgcn_alignbit(ch.x, ph.w, 16), __builtin_amdgcn_alignbit(ch.y, ch.x, 16),
                     __builtin_amdgcn_alignbit(ch.z, ch.y, 16), __builtin_amdgcn_alignbit(ch.w, ch.z, 16)};
      a_l[1] = u32x4{__builtin_amdgcn_alignbit(cl.x, pl.w, 16), __builtin_amdgcn_alignbit(cl.y, cl.x, 16),
                     __builtin_amdgcn_alignbit(cl.z, cl.y, 16), __builtin_amdgcn_alignbit(cl.w, cl.z, 16)};
      a_h[2] = u32x4{ph.w, ch.x, ch.y, ch.z};
      a_l[2] = u32x4{pl.w, cl.x, cl.y, cl.z};
    }
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int bo = ((SW * wv + kx) * 32 + 16 * nt + n16) * W3_LD + 8 * q4;
        const bf16x8 bh = *reinterpret_cast<const bf16x8*>(xb + bo);
        const bf16x8 bl = *reinterpret_cast<const bf16x8*>(xb + XB + bo);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) acc[ky * 3 + kx][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w3_frag(a_h[ky]), bh, acc[ky * 3 + kx][nt], 0, 0, 0);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) acc[ky * 3 + kx][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w3_frag(a_l[ky]), bh, acc[ky * 3 + kx][nt], 0, 0, 0);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) acc[ky * 3 + kx][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w3_frag(a_h[ky]), bl, acc[ky * 3 + kx][nt], 0, 0, 0);
      }
    }
  }

  // ---- the four waves' partial sums -> wave 0, three taps per round through LDS (3 x 1536 floats) ----
  float* const red = reinterpret_cast<float*>(lds);
#pragma unroll
  for (int rr = 0; rr < 3; ++rr) {
    __syncthreads();
    if (wv > 0) {
#pragma unroll
      for (int tp = 0; tp < 3; ++tp)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
          *reinterpret_cast<f32x4*>(red + ((((wv - 1) * 3 + tp) * 2 + nt) * 64 + lane) * 4) = acc[3 * rr + tp][nt];
    }
    __syncthreads();
    if (wv == 0) {
#pragma unroll
      for (int tp = 0; tp < 3; ++tp)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const f32x4 s1 = *reinterpret_cast<const f32x4*>(red + (((0 * 3 + tp) * 2 + nt) * 64 + lane) * 4);
          const f32x4 s2 = *reinterpret_cast<const f32x4*>(red + (((1 * 3 + tp) * 2 + nt) * 64 + lane) * 4);
          const f32x4 s3 = *reinterpret_cast<const f32x4*>(red + (((2 * 3 + tp) * 2 + nt) * 64 + lane) * 4);
          acc[3 * rr + tp][nt] += (s1 + s2) + s3;
        }
    }
  }
  const int K_all = 9 * Cin;
  if (wv == 0) {                                 // D: lane = (input channel n16 of the half, 4 output channels 4 q4 .. + 3)
    float* out = p.slab + (long long)split * p.slab_stride;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int c = c0 + 16 * nt + n16;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = 4 * q4 + r;
          if (n < nn && c < Cin) out[(long long)n * K_all + tap * Cin + c] = acc[tap][nt][r];
        }
      }
  }
  if (want_bias) {                               // db[n] = sum over the split's pixels of dy[.][n]
    __syncthreads();
    if (g_on) {
#pragma unroll
      for (int e = 0; e < 4; ++e) red[tid * 4 + e] = gsum[e];
    }
    __syncthreads();
    if (tid < nn) {
      float t = 0.f;
      for (int u = 0; u < 32; ++u) t += red[(u * 4 + (tid >> 2)) * 4 + (tid & 3)];
      p.bslab[(long long)split * p.bslab_stride + tid] = t;
    }
  }
}

extern "C" int ws_conv3x3_wgrad(const ws_conv3x3_wgrad_args* a, void* stream) {
  WS_REQUIRE(a && a->G && a->X && a->slab, "ws_conv3x3_wgrad: null pointer");
  WS_REQUIRE(a->B > 0 && a->H > 0 && a->Wd > 0 && a->Cin > 0 && a->Cin % 4 == 0 && a->Nn > 0 && a->Nn % 4 == 0,
             "ws_conv3x3_wgrad: Cin %% 4, Nn %% 4 (got %d, %d)", a->Cin, a->Nn);
  WS_REQUIRE((a->sw == 1 || a->sw == 2) && a->Wx > 0 && (a->Wx - 1) / a->sw + 1 == a->Wd,
             "ws_conv3x3_wgrad: stride %d along w (1 or 2), image width %d, gradient width %d = (Wx - 1) / sw + 1", a->sw,
             a->Wx, a->Wd);
  WS_REQUIRE(a->ldx >= a->Cin && a->ldx % 4 == 0 && a->ldg >= a->Nn && a->ldg % 4 == 0,
             "ws_conv3x3_wgrad: leading dimensions (ldx >= Cin, ldg >= Nn, both %% 4)");
  const long long ntiles = (long long)a->B * ((a->H + W3_TH - 1) / W3_TH) * ((a->Wd + 3) / 4);
  WS_REQUIRE(a->nsplit > 0 && a->tiles_per_split > 0 && (long long)a->nsplit * a->tiles_per_split >= ntiles,
             "ws_conv3x3_wgrad: %d splits of %d tiles do not cover the %lld tiles (30 rows x 4 columns)", a->nsplit,
             a->tiles_per_split, ntiles);
  WS_REQUIRE(a->slab_stride >= (long long)a->Nn * 9 * a->Cin && (!a->bslab || a->bslab_stride >= a->Nn),
             "ws_conv3x3_wgrad: slab strides");
  WS_REQUIRE((a->Cin + 31) / 32 <= 65535 && (a->Nn + 31) / 32 <= 65535, "ws_conv3x3_wgrad: channel counts index the launch grid");
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(a->nsplit, (a->Cin + 31) / 32, (a->Nn + 31) / 32);
  const size_t lds1 = (size_t)(2 * 6 * 32 * W3_LD + 2 * W3_AB) * sizeof(__bf16), lds2 = (size_t)(2 * 9 * 32 * W3_LD + 2 * W3_AB) * sizeof(__bf16);
  // the attribute is PER DEVICE: one flag per device of the node, set under a lock (engines on several GPUs of one process,
  // separate_main --jobs).  A failure here -- no device -- shows up as the launch error below, not as an
  // argument error
  if (a->sw != 1) {
    static std::mutex mu;
    static bool attr_set[64] = {};
    int devid = 0;
    if (hipGetDevice(&devid) != hipSuccess || devid < 0 || devid >= 64) devid = 0;
    std::lock_guard<std::mutex> lk(mu);
    if (!attr_set[devid]) {
      attr_set[devid] = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wgrad_kernel<2>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2) == hipSuccess;
      (void)hipGetLastError();
    }
  }
  ws_prof_begin(WS_PROF_GEMM_TN, s);
  static const int v16 = [] { const char* e = getenv("WS_CONV3X3_WGRAD16"); return e ? atoi(e) : 1; }();   // (0: A/B against the 32-row kernel)
  if (a->Nn <= 16 && v16) {   // one 16-row output tile: the 16 x 16 x 32 kernel (41 / 56.3 KB of LDS: under the default limit)
    const size_t l1 = (size_t)(2 * 6 * 32 * W3_LD + 2 * W3_AB16) * sizeof(__bf16), l2 = (size_t)(2 * 9 * 32 * W3_LD + 2 * W3_AB16) * sizeof(__bf16);
    if (a->sw == 1) hipLaunchKernelGGL(conv3x3_wgrad16_kernel<1>, grid, dim3(256), l1, s, *a);
    else hipLaunchKernelGGL(conv3x3_wgrad16_kernel<2>, grid, dim3(256), l2, s, *a);
  } else if (a->sw == 1) hipLaunchKernelGGL(conv3x3_wgrad_kernel<1>, grid, dim3(256), lds1, s, *a);
  else hipLaunchKernelGGL(conv3x3_wgrad_kernel<2>, grid, dim3(256), lds2, s, *a);
  ws_prof_end(WS_PROF_GEMM_TN, s);
  return ws_check_launch("ws_conv3x3_wgrad");
}
