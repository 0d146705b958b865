// Bidirectional LSTM recurrence (hidden 256) for gfx950 -- the 60 % of pBSRNN's FLOPs that
// cannot be batched over time (nn.LSTM inside ResRNN, wesep/models/bsrnn.py:27-33,40).
//
// One workgroup (8 waves) owns a tile of 16*MT independent sequences of ONE direction and
// walks all L steps; sequences never talk to each other, so there is no inter-workgroup sync.
// Per step it needs G[16*MT][1024] = h[16*MT][256] * W_hh^T (+ the x-projection, already in
// `gates`).  fp32 MFMA (v_mfma_f32_16x16x4_f32, exact f32): the sequence index is the MFMA
// row, wave w owns hidden units [32w, 32w+32) = 8 column tiles (4 gates x 2), so i/f/g/o of
// one (sequence, unit) land in the same lane and register index -> the cell update is
// lane-local.  W_hh (1 MB fp32 per direction) cannot live in a CU (160 KB LDS + 512 KB VGPR),
// so it is streamed every step from the XCD's L2 in a pre-packed B-fragment order (one
// coalesced 16-B load per lane per 4 k-steps; MT sequence tiles share each fragment); the
// first fragment block of a step never changes, so it stays resident in registers and the
// step boundary exposes no load.  h_{t-1} sits in LDS in A-fragment order (ds_read_b128, row
// stride 68/260 floats = conflict-free).  Cell state c stays in registers.  The next step's
// x-projection is loaded straight into the (dead) accumulators before this step's stores.
#include "common.h"

#define LH WS_LSTM_H  // 256
#define LG (4 * LH)   // 1024
#define HL_LD 68      // 64 k-quads + 4 pad  (h as A operand, K = 256)
#define DG_LD 260     // 256 k-quads + 4 pad (dgates as A operand, K = 1024)

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// v_exp_f32 / v_rcp_f32 based activations (~1e-7 absolute error; the cell math tolerates it,
// tests/test_kernels_gpu.py::test_lstm_fwd_bwd_vs_torch holds 1e-5 against torch's LSTM).
__device__ __forceinline__ float fast_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.f + __expf(-x));
}
__device__ __forceinline__ float fast_tanh(float x) {
  const float ax = fabsf(x);
  const float e = __expf(-2.f * ax);                      // in (0, 1]: no overflow
  const float t = (1.f - e) * __builtin_amdgcn_rcpf(1.f + e);
  return copysignf(t, x);
}

// ---------------------------------------------------------------------------------------------
// weight packing
// fwd pack: idx = ((((d*8 + w)*16 + ks4)*8 + tile)*64 + lane)*4 + q
//           = W_hh[d][ (tile>>1)*256 + 32w + 16*(tile&1) + (lane&15) ][ 16*ks4 + 4q + (lane>>4) ]
// bwd pack: idx = ((((d*8 + w)*64 + ks4)*2 + s)*64 + lane)*4 + q
//           = W_hh[d][ 16*ks4 + 4q + (lane>>4) ][ 32w + 16s + (lane&15) ]
// ---------------------------------------------------------------------------------------------
__global__ void lstm_pack_kernel(const float* __restrict__ whh_f, const float* __restrict__ whh_r,
                                 float* __restrict__ pf, float* __restrict__ pb) {
  const int total = 2 * LG * LH;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    {
      int r = idx;
      const int q = r & 3; r >>= 2;
      const int lane = r & 63; r >>= 6;
      const int tile = r & 7; r >>= 3;
      const int ks4 = r & 15; r >>= 4;
      const int w = r & 7; r >>= 3;
      const int d = r;
      const float* W = d ? whh_r : whh_f;
      const int col = (tile >> 1) * 256 + 32 * w + 16 * (tile & 1) + (lane & 15);
      const int k = 16 * ks4 + 4 * q + (lane >> 4);
      pf[idx] = W[col * LH + k];
    }
    {
      int r = idx;
      const int q = r & 3; r >>= 2;
      const int lane = r & 63; r >>= 6;
      const int s = r & 1; r >>= 1;
      const int ks4 = r & 63; r >>= 6;
      const int w = r & 7; r >>= 3;
      const int d = r;
      const float* W = d ? whh_r : whh_f;
      const int row = 16 * ks4 + 4 * q + (lane >> 4);
      const int u = 32 * w + 16 * s + (lane & 15);
      pb[idx] = W[row * LH + u];
    }
  }
}

int ws_launch_lstm_pack_bwd_f8(const float* whh_f, const float* whh_r, float* pack_bwd, hipStream_t s);
extern "C" int ws_lstm_pack_bwd_f8(const float* whh_f, const float* whh_r, float* pack_bwd, void* stream) {
  WS_REQUIRE(whh_f && whh_r && pack_bwd, "ws_lstm_pack_bwd_f8: null pointer");
  ws_launch_lstm_pack_bwd_f8(whh_f, whh_r, pack_bwd, (hipStream_t)stream);
  return ws_check_launch("ws_lstm_pack_bwd_f8");
}

int ws_launch_lstm_pack_dx_f8(const float* wcat, float* pack, hipStream_t s);
extern "C" int ws_lstm_pack_dx_f8(const float* wcat, float* pack, void* stream) {
  WS_REQUIRE(wcat && pack, "ws_lstm_pack_dx_f8: null pointer");
  ws_launch_lstm_pack_dx_f8(wcat, pack, (hipStream_t)stream);
  return ws_check_launch("ws_lstm_pack_dx_f8");
}

extern "C" int ws_lstm_pack(const float* whh_f, const float* whh_r, float* pack_fwd,
                            float* pack_bwd, int mode, void* stream) {
  WS_REQUIRE(whh_f && whh_r && pack_fwd && pack_bwd, "ws_lstm_pack: null pointer");
  WS_REQUIRE(mode >= WS_LSTM_F32_MT1 && mode <= WS_LSTM_BF16X3_BLK16, "ws_lstm_pack: bad mode %d", mode);
  if (mode == WS_LSTM_BF16X3_BLK16) {
    ws_launch_lstm_pack_s16(whh_f, whh_r, pack_fwd, pack_bwd, (hipStream_t)stream);
    return ws_check_launch("ws_lstm_pack");
  }
  if (mode >= WS_LSTM_BF16X3) {
    ws_launch_lstm_pack_bf16(whh_f, whh_r, pack_fwd, pack_bwd, (hipStream_t)stream);
    return ws_check_launch("ws_lstm_pack");
  }
  hipLaunchKernelGGL(lstm_pack_kernel, dim3(512), dim3(256), 0, (hipStream_t)stream, whh_f, whh_r,
                     pack_fwd, pack_bwd);
  return ws_check_launch("ws_lstm_pack");
}

__global__ void lstm_cat_ih_kernel(const float* __restrict__ wih_f, const float* __restrict__ wih_r,
                                   const float* __restrict__ bih_f, const float* __restrict__ bhh_f,
                                   const float* __restrict__ bih_r, const float* __restrict__ bhh_r,
                                   int n_in, float* __restrict__ wcat, float* __restrict__ bcat) {
  const int per = LG * n_in;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < 2 * per; idx += gridDim.x * blockDim.x)
    wcat[idx] = idx < per ? wih_f[idx] : wih_r[idx - per];
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < 2 * LG; idx += gridDim.x * blockDim.x)
    bcat[idx] = idx < LG ? bih_f[idx] + bhh_f[idx] : bih_r[idx - LG] + bhh_r[idx - LG];
}

extern "C" int ws_lstm_cat_ih(const float* wih_f, const float* wih_r, const float* bih_f,
                              const float* bhh_f, const float* bih_r, const float* bhh_r, int n_in,
                              float* wcat, float* bcat, void* stream) {
  WS_REQUIRE(wih_f && wih_r && bih_f && bhh_f && bih_r && bhh_r && wcat && bcat && n_in > 0,
             "ws_lstm_cat_ih: bad args");
  hipLaunchKernelGGL(lstm_cat_ih_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, wih_f, wih_r,
                     bih_f, bhh_f, bih_r, bhh_r, n_in, wcat, bcat);
  return ws_check_launch("ws_lstm_cat_ih");
}

// ---------------------------------------------------------------------------------------------
// forward recurrence
// ---------------------------------------------------------------------------------------------
template <int MT>
__global__ __launch_bounds__(512) void lstm_fwd_kernel(const ws_lstm_args p) {
  __shared__ __attribute__((aligned(16))) float hl[2 * MT * 64 * HL_LD];
  const int d = blockIdx.y;
  const int seq0 = blockIdx.x * (16 * MT);
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, l15 = lane & 15, lq = lane >> 4;
  const int L = p.L;

  for (int i = tid; i < 2 * MT * 64 * HL_LD; i += 512) hl[i] = 0.f;

  long long rowbase[MT][4];
  bool valid[MT][4];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int s = seq0 + mt * 16 + 4 * lq + r;
      valid[mt][r] = s < p.nseq;
      const int ss = valid[mt][r] ? s : 0;
      rowbase[mt][r] = (long long)(ss / p.sq_div) * p.sq_s1 + (long long)(ss % p.sq_div) * p.sq_s2;
    }
  const int ubase = 32 * w + l15;  // unit of column tile s: ubase + 16*s
  float c[MT][2][4];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int r = 0; r < 4; ++r) c[mt][s][r] = 0.f;

  const f32x4* wp = reinterpret_cast<const f32x4*>(p.wpack) + (long long)(d * 8 + w) * (16 * 8 * 64) + lane;
  f32x4 b0[8];  // fragment block ks4 = 0: identical every step, kept resident
#pragma unroll
  for (int tile = 0; tile < 8; ++tile) b0[tile] = wp[tile * 64];

  f32x4 acc[MT][8];
  auto load_gx = [&](int t) {  // x-projection of step t straight into the accumulators
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long long row = rowbase[mt][r] + (long long)t * p.step_rows;
        const float* g = p.gates + (row * 2 + d) * LG + ubase;
#pragma unroll
        for (int tile = 0; tile < 8; ++tile) acc[mt][tile][r] = g[(tile >> 1) * 256 + 16 * (tile & 1)];
      }
  };
  load_gx(d == 0 ? 0 : L - 1);
  __syncthreads();

  for (int step = 0; step < L; ++step) {
    const int t = d == 0 ? step : L - 1 - step;
    const int cur = step & 1;
    const float* hcur = hl + cur * (MT * 64 * HL_LD) + (lq * 16 + l15) * HL_LD;
    f32x4 bcur[8], bnxt[8];
#pragma unroll
    for (int tile = 0; tile < 8; ++tile) bcur[tile] = b0[tile];
#pragma unroll 2
    for (int ks4 = 0; ks4 < 16; ++ks4) {
      if (ks4 + 1 < 16) {
#pragma unroll
        for (int tile = 0; tile < 8; ++tile) bnxt[tile] = wp[((ks4 + 1) * 8 + tile) * 64];
      }
      f32x4 a[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        a[mt] = *reinterpret_cast<const f32x4*>(hcur + mt * (64 * HL_LD) + 4 * ks4);
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int tile = 0; tile < 8; ++tile)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) acc[mt][tile] = mfma16(a[mt][q], bcur[tile][q], acc[mt][tile]);
#pragma unroll
      for (int tile = 0; tile < 8; ++tile) bcur[tile] = bnxt[tile];
    }

    // cell update into registers; accumulators are dead afterwards
    f32x4 gi[MT][2], gf[MT][2], gg[MT][2], go[MT][2], hv[MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float ig = fast_sigmoid(acc[mt][0 + s][r]);
          const float fg = fast_sigmoid(acc[mt][2 + s][r]);
          const float g_ = fast_tanh(acc[mt][4 + s][r]);
          const float og = fast_sigmoid(acc[mt][6 + s][r]);
          const float cn = fg * c[mt][s][r] + ig * g_;
          c[mt][s][r] = cn;
          gi[mt][s][r] = ig;
          gf[mt][s][r] = fg;
          gg[mt][s][r] = g_;
          go[mt][s][r] = og;
          hv[mt][s][r] = og * fast_tanh(cn);
        }
    // next step's x-projection: issued BEFORE this step's stores so its wait never covers them
    if (step + 1 < L) load_gx(d == 0 ? step + 1 : L - 2 - step);

    float* hnext = hl + (cur ^ 1) * (MT * 64 * HL_LD);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int u = ubase + 16 * s;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          hnext[((mt * 4 + (u & 3)) * 16 + 4 * lq + r) * HL_LD + (u >> 2)] = hv[mt][s][r];
          if (valid[mt][r]) {
            const long long row = rowbase[mt][r] + (long long)t * p.step_rows;
            float* g = p.gates + (row * 2 + d) * LG + u;
            g[0] = gi[mt][s][r];
            g[256] = gf[mt][s][r];
            g[512] = gg[mt][s][r];
            g[768] = go[mt][s][r];
            p.cbuf[row * (2 * LH) + d * LH + u] = c[mt][s][r];
            p.hcat[row * (2 * LH) + d * LH + u] = hv[mt][s][r];
          }
        }
      }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// backward recurrence (BPTT).  Walks the steps in the reverse of the forward order.
// ---------------------------------------------------------------------------------------------
template <int MT>
__global__ __launch_bounds__(512) void lstm_bwd_kernel(const ws_lstm_args p) {
  __shared__ __attribute__((aligned(16))) float dgl[MT * 64 * DG_LD];
  const int d = blockIdx.y;
  const int seq0 = blockIdx.x * (16 * MT);
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, l15 = lane & 15, lq = lane >> 4;
  const int L = p.L;

  long long rowbase[MT][4];
  bool valid[MT][4];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int s = seq0 + mt * 16 + 4 * lq + r;
      valid[mt][r] = s < p.nseq;
      const int ss = valid[mt][r] ? s : 0;
      rowbase[mt][r] = (long long)(ss / p.sq_div) * p.sq_s1 + (long long)(ss % p.sq_div) * p.sq_s2;
    }
  const int ubase = 32 * w + l15;
  const long long prev_rows = (d == 0 ? -1 : 1) * p.step_rows;  // row of the forward-previous step

  f32x4 dh[MT][2], dc[MT][2];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int r = 0; r < 4; ++r) dh[mt][s][r] = dc[mt][s][r] = 0.f;

  const f32x4* wp = reinterpret_cast<const f32x4*>(p.wpack) + (long long)(d * 8 + w) * (64 * 2 * 64) + lane;
  const f32x4 b00 = wp[0], b01 = wp[64];  // fragment block ks4 = 0, resident

  // prefetched step inputs
  f32x4 n_i[MT][2], n_f[MT][2], n_g[MT][2], n_o[MT][2], n_c[MT][2], n_cp[MT][2], n_dh[MT][2];
  auto load_step = [&](int t) {
    const bool has_prev = d == 0 ? (t > 0) : (t < L - 1);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int u = ubase + 16 * s;
          const long long row = rowbase[mt][r] + (long long)t * p.step_rows;
          const float* g = p.gates + (row * 2 + d) * LG + u;
          n_i[mt][s][r] = g[0];
          n_f[mt][s][r] = g[256];
          n_g[mt][s][r] = g[512];
          n_o[mt][s][r] = g[768];
          const long long hc = row * (2 * LH) + d * LH + u;
          n_c[mt][s][r] = p.cbuf[hc];
          n_dh[mt][s][r] = p.dhcat[hc];
          n_cp[mt][s][r] = has_prev ? p.cbuf[hc + prev_rows * (2 * LH)] : 0.f;
        }
  };
  load_step(d == 0 ? L - 1 : 0);

  for (int step = 0; step < L; ++step) {
    const int t = d == 0 ? L - 1 - step : step;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int u = ubase + 16 * s;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float ig = n_i[mt][s][r], fg = n_f[mt][s][r], gg = n_g[mt][s][r], og = n_o[mt][s][r];
          const float dhv = n_dh[mt][s][r] + dh[mt][s][r];
          const float tc = fast_tanh(n_c[mt][s][r]);
          const float dov = dhv * tc;
          const float dcv = dc[mt][s][r] + dhv * og * (1.f - tc * tc);
          dc[mt][s][r] = dcv * fg;
          float pi = dcv * gg * ig * (1.f - ig);
          float pf = dcv * n_cp[mt][s][r] * fg * (1.f - fg);
          float pg = dcv * ig * (1.f - gg * gg);
          float po = dov * og * (1.f - og);
          if (valid[mt][r]) {
            const long long row = rowbase[mt][r] + (long long)t * p.step_rows;
            float* g = p.gates + (row * 2 + d) * LG + u;
            g[0] = pi;
            g[256] = pf;
            g[512] = pg;
            g[768] = po;
          } else {
            pi = pf = pg = po = 0.f;
          }
          // A-fragment order: column col = gate*256 + u  ->  [col & 3][seq row][col >> 2]
          float* dst = dgl + ((mt * 4 + (u & 3)) * 16 + 4 * lq + r) * DG_LD + (u >> 2);
          dst[0] = pi;
          dst[64] = pf;
          dst[128] = pg;
          dst[192] = po;
        }
      }
    __syncthreads();
    if (step + 1 < L) load_step(d == 0 ? L - 2 - step : step + 1);

#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int r = 0; r < 4; ++r) dh[mt][s][r] = 0.f;
    const float* arow = dgl + (lq * 16 + l15) * DG_LD;
    f32x4 bcur[2], bnxt[2];
    bcur[0] = b00;
    bcur[1] = b01;
#pragma unroll 4
    for (int ks4 = 0; ks4 < 64; ++ks4) {
      if (ks4 + 1 < 64) {
        bnxt[0] = wp[((ks4 + 1) * 2 + 0) * 64];
        bnxt[1] = wp[((ks4 + 1) * 2 + 1) * 64];
      }
      f32x4 a[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        a[mt] = *reinterpret_cast<const f32x4*>(arow + mt * (64 * DG_LD) + 4 * ks4);
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) dh[mt][s] = mfma16(a[mt][q], bcur[s][q], dh[mt][s]);
      bcur[0] = bnxt[0];
      bcur[1] = bnxt[1];
    }
    __syncthreads();
  }
}

static int lstm_check(const ws_lstm_args* a, bool bwd, const char* who) {
  WS_REQUIRE(a && a->gates && a->cbuf && a->hcat && a->wpack, "%s: null pointer", who);
  WS_REQUIRE(!bwd || a->dhcat, "%s: null dhcat", who);
  WS_REQUIRE(a->nseq > 0 && a->L > 0 && a->sq_div > 0, "%s: bad nseq/L/sq_div", who);
  WS_REQUIRE((a->mode & 255) >= WS_LSTM_F32_MT1 && (a->mode & 255) <= WS_LSTM_BF16X3_BLK16, "%s: bad mode %d", who,
             a->mode);
  WS_REQUIRE(!a->run_if || (!bwd && (a->mode & 255) >= WS_LSTM_BF16X3) || (bwd && (a->mode & 255) >= WS_LSTM_BF16X3_BLK),
             "%s: run_if is honoured by the split-bf16 forward kernels and the blocked-layout BPTT kernels only", who);
  WS_REQUIRE(a->gfmt >= WS_GATES_F32 && a->gfmt <= WS_GATES_H2F, "%s: bad gfmt %d", who, a->gfmt);
  WS_REQUIRE(a->gfmt != WS_GATES_H2F || !bwd || a->amax, "%s: WS_GATES_H2F needs amax", who);
  WS_REQUIRE(a->gfmt == WS_GATES_F32 || (a->mode & 255) >= WS_LSTM_BF16X3_BLK,
             "%s: gfmt %d is a format of the blocked-layout modes", who, a->gfmt);
  WS_REQUIRE(a->gfmt == WS_GATES_F32 || bwd || a->gates_in, "%s: gfmt %d needs gates_in (the fp32 pre-activations)", who,
             a->gfmt);
  WS_REQUIRE(a->gfmt != WS_GATES_H2S || !bwd || a->dgates, "%s: WS_GATES_H2S needs dgates", who);
  WS_REQUIRE(a->gfmt == WS_GATES_F32 || ((a->mode >> 8) & 7) == 0, "%s: probe builds are WS_GATES_F32 only", who);
  WS_REQUIRE(a->rfmt == 0 || ((a->rfmt == 2 || a->rfmt == 3) && bwd && a->mode == WS_LSTM_BF16X3_BLK && a->gfmt == WS_GATES_H2F),
             "%s: rfmt %d (2 / 3 = fp16 recurrence on fp16 + FP8 weights, 3: the lo term on the FP8 MFMA: ws_lstm_bwd, "
             "WS_LSTM_BF16X3_BLK, WS_GATES_H2F only)", who, a->rfmt);
  WS_REQUIRE(!a->dxn || (bwd && a->rfmt == 2 && a->wxpack && a->dxn_dir_stride > 0 && a->dxn_dir_stride < (1ll << 29) && !a->run_if),
             "%s: dxn (d(xn) inside the BPTT, ABI v19) needs rfmt 2, wxpack (ws_lstm_pack_dx_f8), 0 < dxn_dir_stride < 2^29 floats "
             "(32-bit store offsets) and no run_if", who);
  return WS_OK;
}

extern "C" int ws_lstm_fwd(const ws_lstm_args* a, void* stream) {
  int rc = lstm_check(a, false, "ws_lstm_fwd");
  if (rc != WS_OK) return rc;
  hipStream_t s = (hipStream_t)stream;
  const int per = 16 * a->mode;
  dim3 grid((a->nseq + per - 1) / per, 2), block(512);
  const bool timed = a->run_if == nullptr;  // a predicated fall-back launch is normally empty: not a sample of this kind
  if (timed) ws_prof_begin(WS_PROF_LSTM_FWD, s);
  if ((a->mode & 255) == WS_LSTM_BF16X3_BLK16)
    ws_launch_lstm_fwd_s16(a, s);
  else if ((a->mode & 255) >= WS_LSTM_BF16X3)
    ws_launch_lstm_fwd_bf16(a, s);
  else if (a->mode == WS_LSTM_F32_MT1)
    hipLaunchKernelGGL((lstm_fwd_kernel<1>), grid, block, 0, s, *a);
  else
    hipLaunchKernelGGL((lstm_fwd_kernel<2>), grid, block, 0, s, *a);
  if (timed) ws_prof_end(WS_PROF_LSTM_FWD, s);
  return ws_check_launch("ws_lstm_fwd");
}

extern "C" int ws_lstm_bwd(const ws_lstm_args* a, void* stream) {
  int rc = lstm_check(a, true, "ws_lstm_bwd");
  if (rc != WS_OK) return rc;
  hipStream_t s = (hipStream_t)stream;
  const int per = 16 * a->mode;
  dim3 grid((a->nseq + per - 1) / per, 2), block(512);
  const bool timed = a->run_if == nullptr;  // a predicated fall-back launch is normally empty: not a sample of this kind
  if (timed) ws_prof_begin(WS_PROF_LSTM_BWD, s);
  if ((a->mode & 255) == WS_LSTM_BF16X3_BLK16)
    ws_launch_lstm_bwd_s16(a, s);
  else if ((a->mode & 255) >= WS_LSTM_BF16X3)
    ws_launch_lstm_bwd_bf16(a, s);
  else if (a->mode == WS_LSTM_F32_MT1)
    hipLaunchKernelGGL((lstm_bwd_kernel<1>), grid, block, 0, s, *a);
  else
    hipLaunchKernelGGL((lstm_bwd_kernel<2>), grid, block, 0, s, *a);
  if (timed) ws_prof_end(WS_PROF_LSTM_BWD, s);
  return ws_check_launch("ws_lstm_bwd");
}
