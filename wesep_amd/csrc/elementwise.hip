// HBM-bound elementwise / reduction kernels of the pBSRNN training step on gfx950:
//   * speaker fusion affine (SpeakerFuseLayer multiply/additive/FiLM, speaker.py:81-125,
//     norm.py:118-139) without the reference's [R,K,Tf,256] embedding expansion,
//   * SI-SDR loss and its closed-form gradient (auraloss SISDRLoss via losses.py:24-25),
//   * per-tensor gradient clip (funcs.py:79-88) + Adam with coupled L2 (train.py:237-238)
//     as ONE multi-tensor launch pair instead of ~640 host syncs per step.
// Coalesced 16-byte accesses where the layout allows, wave-shuffle + LDS block reductions.
#include "common.h"

// ------------------------------------------------------------------------------------------
// out[p][n] = z[p][n] * (a0 + a[r][n]) + b[r][n]
// ------------------------------------------------------------------------------------------
// z and out may alias (in-place use by the concat fuse), so neither is __restrict__.
__global__ void affine_fwd_kernel(const float* z, const float* __restrict__ a,
                                  const float* __restrict__ b, float a0, long long rows,
                                  int rows_per_r, int N, float* out) {
  const long long total4 = rows * N / 4;
  const int n4 = N / 4;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total4;
       i += (long long)gridDim.x * blockDim.x) {
    const long long row = i / n4;
    const int c = (int)(i - row * n4) * 4;
    const int r = (int)(row / rows_per_r);
    f32x4 v = *reinterpret_cast<const f32x4*>(z + i * 4);
    f32x4 av = {a0, a0, a0, a0}, bv = {0.f, 0.f, 0.f, 0.f};
    if (a) av += *reinterpret_cast<const f32x4*>(a + (long long)r * N + c);
    if (b) bv = *reinterpret_cast<const f32x4*>(b + (long long)r * N + c);
    *reinterpret_cast<f32x4*>(out + i * 4) = v * av + bv;
  }
}

extern "C" int ws_affine_fwd(const float* z, const float* a, const float* b, float a0,
                             long long rows, int rows_per_r, int N, float* out, void* stream) {
  WS_REQUIRE(z && out && rows > 0 && rows_per_r > 0 && N > 0 && N % 4 == 0, "ws_affine_fwd: bad args");
  long long blocks = (rows * N / 4 + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(affine_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     z, a, b, a0, rows, rows_per_r, N, out);
  return ws_check_launch("ws_affine_fwd");
}

// grid (R, nsplit); 256 threads = 128 columns x 2 row lanes (N <= 128).
__global__ __launch_bounds__(256) void affine_bwd_kernel(
    const float* __restrict__ dz, const float* __restrict__ z_in, const float* __restrict__ a,
    float a0, int rows_per_r, int N, int nsplit, float* __restrict__ dz_in,
    float* __restrict__ da_slab, float* __restrict__ db_slab, float* __restrict__ da, float* __restrict__ db,
    unsigned* counter) {
  __shared__ float sh[2][128];
  const int r = blockIdx.x, split = blockIdx.y, R = gridDim.x;
  const int col = threadIdx.x & 127, rl = threadIdx.x >> 7;
  float sa = 0.f, sb = 0.f;
  if (col < N) {
    const float scale = a0 + (a ? a[(long long)r * N + col] : 0.f);
    const int chunk = (rows_per_r + nsplit - 1) / nsplit;
    const int lo = split * chunk, hi = min(rows_per_r, lo + chunk);
    for (int j = lo + rl; j < hi; j += 2) {
      const long long o = ((long long)r * rows_per_r + j) * N + col;
      const float g = dz[o];
      if (da_slab) sa += g * z_in[o];
      sb += g;
      if (dz_in) dz_in[o] = g * scale;
    }
  }
  if (rl == 1) {
    sh[0][col] = sa;
    sh[1][col] = sb;
  }
  __syncthreads();
  if (rl == 0 && col < N) {
    const long long o = ((long long)split * R + r) * N + col;
    if (counter) {  // read by the row's last workgroup below: agent-scope stores (common.h ws_last_block)
      if (da_slab) ws_st_agent(da_slab + o, sa + sh[0][col]);
      if (db_slab) ws_st_agent(db_slab + o, sb + sh[1][col]);
    } else {
      if (da_slab) da_slab[o] = sa + sh[0][col];
      if (db_slab) db_slab[o] = sb + sh[1][col];
    }
  }
  // (optional) the last workgroup of every ROW r (counter word r) adds that row's splits up, in split order, into
  // da[r] / db[r]: threads 0..127 -> da, 128..255 -> db; no reduction launches
  if (counter && ws_last_block(counter + r, (unsigned)nsplit)) {
    float* dst = rl == 0 ? da : db;
    const float* src = rl == 0 ? da_slab : db_slab;
    if (dst && col < N) {
      float t = 0.f;
      for (int k = 0; k < nsplit; ++k) t += ws_ld_agent(src + ((long long)k * R + r) * N + col);
      dst[(long long)r * N + col] = t;
    }
  }
}

// N % 4 == 0: the same sums and the same output with 16-byte accesses -- 256 threads = 32 column quads x 8 row lanes, two rows per
// lane in flight (64 B of loads per thread and iteration instead of 8: the scalar kernel above moved 0.8 GB in 0.33 ms, six
// launches per pBSRNN step)
__global__ __launch_bounds__(256) void affine_bwd4_kernel(
    const float* __restrict__ dz, const float* __restrict__ z_in, const float* __restrict__ a,
    float a0, int rows_per_r, int N, int nsplit, float* __restrict__ dz_in,
    float* __restrict__ da_slab, float* __restrict__ db_slab, float* __restrict__ da, float* __restrict__ db,
    unsigned* counter) {
  __shared__ f32x4 sh[2][8][32];
  const int r = blockIdx.x, split = blockIdx.y, R = gridDim.x;
  const int c4 = threadIdx.x & 31, rl = threadIdx.x >> 5;
  f32x4 sa = {0.f, 0.f, 0.f, 0.f}, sb = {0.f, 0.f, 0.f, 0.f};
  const bool live = 4 * c4 < N;
  if (live) {
    f32x4 scale = {a0, a0, a0, a0};
    if (a) scale += *reinterpret_cast<const f32x4*>(a + (long long)r * N + 4 * c4);
    const int chunk = (rows_per_r + nsplit - 1) / nsplit;
    const int lo = split * chunk, hi = min(rows_per_r, lo + chunk);
    const long long base = (long long)r * rows_per_r * N + 4 * c4;
    for (int j = lo + rl; j < hi; j += 16) {
      const bool two = j + 8 < hi;
      const long long o0 = base + (long long)j * N, o1 = base + (long long)(two ? j + 8 : j) * N;
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(dz + o0);
      f32x4 g1 = *reinterpret_cast<const f32x4*>(dz + o1);
      if (da_slab) {
        const f32x4 z0 = *reinterpret_cast<const f32x4*>(z_in + o0);
        const f32x4 z1 = *reinterpret_cast<const f32x4*>(z_in + o1);
        sa += g0 * z0;
        if (two) sa += g1 * z1;
      }
      sb += g0;
      if (two) sb += g1;
      if (dz_in) {
        *reinterpret_cast<f32x4*>(dz_in + o0) = g0 * scale;
        if (two) *reinterpret_cast<f32x4*>(dz_in + o1) = g1 * scale;
      }
    }
  }
  sh[0][rl][c4] = sa;
  sh[1][rl][c4] = sb;
  __syncthreads();
  // threads 0..127: column tid of the da partial; 128..255: column tid - 128 of the db partial -- row lanes added in order
  {
    const int which = threadIdx.x >> 7, col = threadIdx.x & 127;
    float* slab = which == 0 ? da_slab : db_slab;
    if (slab && col < N) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) t += reinterpret_cast<const float*>(&sh[which][k][0])[col];
      const long long o = ((long long)split * R + r) * N + col;
      if (counter) ws_st_agent(slab + o, t);
      else slab[o] = t;
    }
  }
  if (counter && ws_last_block(counter + r, (unsigned)nsplit)) {
    const int which = threadIdx.x >> 7, col = threadIdx.x & 127;
    float* dst = which == 0 ? da : db;
    const float* src = which == 0 ? da_slab : db_slab;
    if (dst && col < N) {
      float t = 0.f;
      for (int k = 0; k < nsplit; ++k) t += ws_ld_agent(src + ((long long)k * R + r) * N + col);
      dst[(long long)r * N + col] = t;
    }
  }
}

extern "C" int ws_affine_bwd(const float* dz, const float* z_in, const float* a, float a0,
                             long long rows, int rows_per_r, int N, int nsplit, float* dz_in,
                             float* da_slab, float* db_slab, float* da, float* db, unsigned* counter, void* stream) {
  WS_REQUIRE(dz && rows > 0 && rows_per_r > 0 && N > 0 && N <= 128 && nsplit > 0,
             "ws_affine_bwd: bad args");
  WS_REQUIRE(rows % rows_per_r == 0, "ws_affine_bwd: rows %% rows_per_r != 0");
  WS_REQUIRE(!da_slab || z_in, "ws_affine_bwd: da needs z_in");
  WS_REQUIRE((!da && !db) || counter, "ws_affine_bwd: da / db need a counter word");
  WS_REQUIRE((!da || da_slab) && (!db || db_slab), "ws_affine_bwd: da / db are sums of their slabs");
  const int R = (int)(rows / rows_per_r);
  if (N % 4 == 0)
    hipLaunchKernelGGL(affine_bwd4_kernel, dim3(R, nsplit), dim3(256), 0, (hipStream_t)stream, dz, z_in,
                       a, a0, rows_per_r, N, nsplit, dz_in, da_slab, db_slab, da, db, (da || db) ? counter : nullptr);
  else
    hipLaunchKernelGGL(affine_bwd_kernel, dim3(R, nsplit), dim3(256), 0, (hipStream_t)stream, dz, z_in,
                       a, a0, rows_per_r, N, nsplit, dz_in, da_slab, db_slab, da, db, (da || db) ? counter : nullptr);
  return ws_check_launch("ws_affine_bwd");
}

// ------------------------------------------------------------------------------------------
// SI-SDR.  One workgroup (1024 threads) per row; the row (2 x 256 KB at 4 s) stays in L2 for
// the three passes: means, centred dot products, residual energy (no cancellation).
// rowstat[r] = (mean_x, mean_t, alpha, c1, c2, sisdr_dB, 0, 0):
//   d loss / d x[n] = gout/R * (c1 * tc[n] + c2 * res[n]),  tc = t - mean_t, res = xc - alpha tc
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void sisdr_fwd_kernel(const float* __restrict__ est,
                                                         const float* __restrict__ tgt, int T,
                                                         float eps, float* __restrict__ rowstat) {
  __shared__ float red[16];
  const int r = blockIdx.x;
  const float* x = est + (long long)r * T;
  const float* t = tgt + (long long)r * T;
  float sx = 0.f, st = 0.f;
  for (int i = threadIdx.x; i < T; i += 1024) {
    sx += x[i];
    st += t[i];
  }
  const float mx = ws_block_sum(sx, red) / (float)T;
  const float mt = ws_block_sum(st, red) / (float)T;
  float sxt = 0.f, stt = 0.f;
  for (int i = threadIdx.x; i < T; i += 1024) {
    const float xc = x[i] - mx, tc = t[i] - mt;
    sxt += xc * tc;
    stt += tc * tc;
  }
  sxt = ws_block_sum(sxt, red);
  stt = ws_block_sum(stt, red);
  const float alpha = sxt / (stt + eps);
  float srr = 0.f, srt = 0.f;
  for (int i = threadIdx.x; i < T; i += 1024) {
    const float tc = t[i] - mt;
    const float res = (x[i] - mx) - alpha * tc;
    srr += res * res;
    srt += res * tc;
  }
  srr = ws_block_sum(srr, red);
  srt = ws_block_sum(srt, red);
  if (threadIdx.x == 0) {
    const float A = alpha * alpha * stt;  // |alpha t|^2
    const float B = srr;
    const float ratio = A / (B + eps);
    const float val = 10.f * log10f(ratio + eps);
    // L_r = -val;  dL/dA, dL/dB
    const float k = 10.f / logf(10.f) / (ratio + eps);
    const float dLdA = -k / (B + eps);
    const float dLdB = k * A / ((B + eps) * (B + eps));
    const float dalpha = 1.f / (stt + eps);  // d alpha / d xc[n] = tc[n] * dalpha
    const float c1 = dLdA * 2.f * alpha * stt * dalpha - dLdB * 2.f * srt * dalpha;
    const float c2 = dLdB * 2.f;
    float* o = rowstat + (long long)r * 8;
    o[0] = mx;
    o[1] = mt;
    o[2] = alpha;
    o[3] = c1;
    o[4] = c2;
    o[5] = val;
    o[6] = 0.f;
    o[7] = 0.f;
  }
}

__global__ void sisdr_mean_kernel(const float* __restrict__ rowstat, int R, float* __restrict__ loss) {
  float s = 0.f;
  for (int r = threadIdx.x; r < R; r += 64) s += rowstat[(long long)r * 8 + 5];
  s = ws_wave_sum(s);
  if (threadIdx.x == 0) loss[0] = -s / (float)R;
}

extern "C" int ws_sisdr_fwd(const float* est, const float* tgt, int R, int T, float eps,
                            float* rowstat, float* loss, void* stream) {
  WS_REQUIRE(est && tgt && rowstat && loss && R > 0 && T > 0, "ws_sisdr_fwd: bad args");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(sisdr_fwd_kernel, dim3(R), dim3(1024), 0, s, est, tgt, T, eps, rowstat);
  hipLaunchKernelGGL(sisdr_mean_kernel, dim3(1), dim3(64), 0, s, rowstat, R, loss);
  return ws_check_launch("ws_sisdr_fwd");
}

__global__ void sisdr_bwd_kernel(const float* __restrict__ est, const float* __restrict__ tgt,
                                 const float* __restrict__ rowstat, const float* __restrict__ gout,
                                 int R, int T, float* __restrict__ dest) {
  const long long total = (long long)R * T;
  const float go = gout[0] / (float)R;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(i / T);
    const float* o = rowstat + (long long)r * 8;
    const float tc = tgt[i] - o[1];
    const float res = (est[i] - o[0]) - o[2] * tc;
    dest[i] = go * (o[3] * tc + o[4] * res);
  }
}

extern "C" int ws_sisdr_bwd(const float* est, const float* tgt, const float* rowstat,
                            const float* gout, int R, int T, float* dest, void* stream) {
  WS_REQUIRE(est && tgt && rowstat && gout && dest && R > 0 && T > 0, "ws_sisdr_bwd: bad args");
  long long blocks = ((long long)R * T + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(sisdr_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     est, tgt, rowstat, gout, R, T, dest);
  return ws_check_launch("ws_sisdr_bwd");
}

// ------------------------------------------------------------------------------------------
// multi-tensor per-tensor-norm clip + Adam (coupled L2)
// ------------------------------------------------------------------------------------------
#define MT_CHUNK 16384  // elements per workgroup pass

// one workgroup per tensor: norms[i] = ||grad_i||_2   (largest tensor 262144 elements)
// guard (optional, two words): both are set to 1 when a norm is not finite -- a NaN / Inf gradient anywhere in the tensor.
// The caller zeroes guard[0] before the launch and hands it to ws_clip_adam_step as a skip word, so a poisoned gradient
// (a BPTT launch whose bounded wait timed out, on this rank or -- through the all-reduce -- on any rank) never reaches
// the weights; guard[1] is sticky, for the host's asynchronous bookkeeping
__global__ __launch_bounds__(1024) void grad_norms_kernel(const ws_tensor_ref* __restrict__ tab,
                                                          float* __restrict__ norms, unsigned* __restrict__ guard) {
  __shared__ float red[16];
  const ws_tensor_ref t = tab[blockIdx.x];
  float s = 0.f;
  if (t.grad) {
    for (long long i = threadIdx.x; i < t.numel; i += 1024) {
      const float g = t.grad[i];
      s += g * g;
    }
  }
  s = ws_block_sum(s, red);
  if (threadIdx.x == 0) {
    const float n = sqrtf(s);
    norms[blockIdx.x] = n;
    if (guard && !(fabsf(n) <= 3.4028234e38f)) guard[0] = 1u;  // NaN or Inf (plain store: every writer writes 1)
  }
}

extern "C" int ws_grad_norms(const ws_tensor_ref* tab, int ntensors, float* norms, unsigned* guard, void* stream) {
  WS_REQUIRE(tab && norms && ntensors > 0, "ws_grad_norms: bad args");
  hipLaunchKernelGGL(grad_norms_kernel, dim3(ntensors), dim3(1024), 0, (hipStream_t)stream, tab, norms, guard);
  return ws_check_launch("ws_grad_norms");
}

// grid (ntensors, chunks); each workgroup grid-strides over its tensor in MT_CHUNK pieces.
__global__ __launch_bounds__(256) void clip_adam_kernel(const ws_tensor_ref* __restrict__ tab,
                                                        const float* __restrict__ norms, float clip,
                                                        float lr, float beta1, float beta2,
                                                        float eps, float wd, float bc1,
                                                        float bc2_sqrt, int clip_only,
                                                        const unsigned* __restrict__ skip0,
                                                        const unsigned* __restrict__ skip1,
                                                        const unsigned* __restrict__ step_lag, int step) {
  // skip words (optional device words, read at kernel start; uniform): the whole launch does nothing when one of them is
  // non-zero -- the non-finite-gradient guard of ws_grad_norms, the sticky status word of an in-place BPTT time-out
  if ((skip0 && *skip0 != 0u) || (skip1 && *skip1 != 0u)) return;
  const ws_tensor_ref t = tab[blockIdx.x];
  if (!t.grad) return;
  float coef = 1.f;
  if (clip > 0.f) {
    const float c = clip / (norms[blockIdx.x] + 1e-6f);
    if (c < 1.f) coef = c;
  }
  if (step_lag && !clip_only) {
    // bias corrections of the step count the DEVICE knows: attempted steps minus the skipped ones the host has not
    // subtracted yet (wesep_hip.h); uniform, in double like the host's
    // ALWAYS recomputed here when a lag word is given, also for lag 0 (round 6): under DDP the ranks learn of a
    // skipped step at different host polls, so in one step one rank could take the host's libm pow (lag already reconciled) and
    // another the device's -- one ulp apart is enough to break bit-identical replicas.  One place, one pow.
    const unsigned lag = *step_lag;
    const double eff = (double)max(step - (int)lag, 1);
    bc1 = (float)(1.0 - pow((double)beta1, eff));
    bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, eff));
  }
  const float step_size = lr / bc1;
  for (long long i = (long long)blockIdx.y * blockDim.x + threadIdx.x; i < t.numel;
       i += (long long)gridDim.y * blockDim.x) {
    float g = t.grad[i] * coef;
    if (coef != 1.f) t.grad[i] = g;
    if (clip_only) continue;
    const float p = t.param[i];
    g += wd * p;
    const float m = beta1 * t.exp_avg[i] + (1.f - beta1) * g;
    const float v = beta2 * t.exp_avg_sq[i] + (1.f - beta2) * g * g;
    t.exp_avg[i] = m;
    t.exp_avg_sq[i] = v;
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    t.param[i] = p - step_size * (m / denom);
  }
}

extern "C" int ws_clip_adam_step(const ws_tensor_ref* tab, int ntensors, const float* norms,
                                 float clip, float lr, float beta1, float beta2, float eps,
                                 float weight_decay, int step, int clip_only, const unsigned* skip0,
                                 const unsigned* skip1, const unsigned* step_lag, void* stream) {
  WS_REQUIRE(tab && ntensors > 0, "ws_clip_adam_step: bad args");
  WS_REQUIRE(clip <= 0.f || norms, "ws_clip_adam_step: clip needs norms");
  WS_REQUIRE(clip_only || step >= 1, "ws_clip_adam_step: step must be >= 1");
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  hipLaunchKernelGGL(clip_adam_kernel, dim3(ntensors, 16), dim3(256), 0, (hipStream_t)stream, tab,
                     norms, clip, lr, beta1, beta2, eps, weight_decay, (float)bc1,
                     (float)sqrt(bc2), clip_only, skip0, skip1, step_lag, step);
  return ws_check_launch("ws_clip_adam_step");
}

__global__ void guard_commit_kernel(unsigned* __restrict__ guard, const unsigned* __restrict__ skip1) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const bool skipped = guard[0] != 0u || (skip1 && *skip1 != 0u);
  if (skipped) {
    guard[1] += 1u;
    guard[2] += 1u;
    guard[3] += 1u;
  } else {
    guard[2] = 0u;
  }
}

extern "C" int ws_guard_commit(unsigned* guard, const unsigned* skip1, void* stream) {
  WS_REQUIRE(guard, "ws_guard_commit: null pointer");
  hipLaunchKernelGGL(guard_commit_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, guard, skip1);
  return ws_check_launch("ws_guard_commit");
}
