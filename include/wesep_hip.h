/* wesep_hip.h -- C ABI of libwesep_hip.so, the MI355X (gfx950) device library behind
 * wesep_amd's pBSRNN training path.
 *
 * wesep (the reference) has no FFI/operator interface of its own: its device boundary is
 * the set of stock ATen operators its Python modules call.  Each entry
 * point below replaces the ATen operator sequence of the reference lines it cites.  All
 * pointers are DEVICE pointers to fp32 unless stated; all tensors are caller-allocated;
 * `stream` is a hipStream_t (the caller's current stream); no call allocates, frees or
 * synchronises.  Return value: WS_OK or a negative WS_ERR_* code, message via
 * ws_last_error().  Host-side binding: wesep_amd/_lib.py (ctypes); see INTEGRATION.md.
 *
 * Activation layout ("Z layout"): [R][K][Tf][N] fp32, N = feature_dim contiguous, one
 * "position" p = (r*K + k)*Tf + t per row.  The reference's [B, K*N, T] / [B*K, N, T] /
 * [B*T, N, K] views (bsrnn.py:73-81) are strided row sets of this one buffer, so its two
 * permute().contiguous() copies per BSNet do not exist here.
 */
#ifndef WESEP_HIP_H_
#define WESEP_HIP_H_

#ifdef __cplusplus
extern "C" {
#endif

#define WS_ABI_VERSION 20
#define WS_OK 0
#define WS_ERR_INVALID (-1)
#define WS_ERR_LAUNCH (-2)

int ws_abi_version(void);
const char* ws_last_error(void);

/* ---- profiling: HIP-event brackets recorded on the launch stream ------------------- */
#define WS_PROF_LSTM_FWD 0
#define WS_PROF_LSTM_BWD 1
#define WS_PROF_GEMM_NT 2
#define WS_PROF_GEMM_TN 3
#define WS_PROF_NKINDS 4
int ws_prof_enable(int on);                       /* 1: bracket every launch of the kinds above */
int ws_prof_collect(int kind, double* total_ms, long long* launches); /* syncs the events; resets */
/* Test support: fills 60 KB of LDS per workgroup with `value` (NaN) on `stream` and keeps the workgroups alive for
 * `spins` LDS reads per thread; beside product kernels on another stream it exposes reads of LDS a kernel never wrote. */
int ws_debug_dirty_lds(float value, int nblocks, int spins, float* sink, void* stream);
/* Test support (ABI v15): holds `nblocks` compute units (112 KB of LDS each: no workgroup of the co-resident recurrences or
 * of the weight-gradient GEMM fits beside it) for `usec` microseconds of wall clock, or until *stop != 0 (optional
 * device word) -- a resident-collective-shaped occupant for the robustness tests.  Bounded: usec <= 2 000 000.        */
int ws_debug_occupy(int nblocks, int usec, const unsigned* stop, float* sink, void* stream);

/* ---- row addressing used by the GEMMs ---------------------------------------------------
 * row m of a matrix lives at  base + (m / div) * s1 + (m % div) * s2   (elements).       */

/* Per-group overrides for grouped (per-band) launches; array lives in DEVICE memory.      */
typedef struct ws_group_nt {
  const float* W;
  const float* bias;
  const float* gamma;
  const float* beta;
  long long a_off;    /* added to A                                     */
  long long c_off;    /* added to C (and to R, T)                       */
  long long st_base;  /* stat-index base                                */
  int K, N, ldw, pad_;
} ws_group_nt;

/* Implicit patch matrix (ABI v8): with conv.on != 0 the A operand of ws_gemm_nt / ws_gemm_tn (split-bf16 kernels
 * only) is the im2col matrix of a channels-last image x [R][H][W][C] that is never materialised -- A points to x and
 *   A[m][(ky*k + kx)*C + c],  m = (r*Ho + ho)*Wo + wo,  is
 *   mode 0 (convolution view):  x[r][ho*sh + ky*dil - p][wo*sw + kx*dil - p][c]
 *   mode 1 (transposed view):   x[r][(ho + p - ky*dil)/sh][(wo + p - kx*dil)/sw][c] where both divisions are exact,
 * and 0 outside the image (dil = 0 means 1).  A dilated Conv1d over [R][T][C] is the view H = 1, W = T, k x k taps,
 * p = dil * (k / 2): the rows ky != k/2 fall outside the one-row image (ECAPA-TDNN's Res2Net branches).  K = k*k*C, C % 4 == 0, a_div / a_s1 / a_s2 are ignored; mode 1 needs sh, sw in {1, 2}.
 * Conv2d = mode 0 on the input; its input gradient = mode 1 on the output gradient (K = k*k*Cout);
 * ConvTranspose2d = mode 1 on the input, its input gradient = mode 0 on the output gradient; the weight gradients
 * are ws_gemm_tn with the mode-0 view as A.  Replaces F.conv2d / F.conv_transpose2d of wesep/modules/dpccn/convs.py:28-110
 * and of the wespeaker ResNet (round 1 wrote the 9x larger patch matrix to HBM and read it back).          */
typedef struct ws_conv_view {
  int on, mode;
  int H, W, C;       /* the image A points to */
  int Ho, Wo;        /* patch grid (rows of the implicit matrix per image: Ho*Wo) */
  int k, sh, sw, p;
  int dil;           /* tap spacing; 0 = 1 */
  int ldp;           /* floats between consecutive pixels of the image (0 = C): the image may be the first C columns
                        of a wider channels-last tensor, e.g. the growing feature map of a dense block */
} ws_conv_view;

/* C[m][n] = epi( sum_k pro(A[m][k]) * W[n][k] )        (torch Linear / Conv1d(k=1) layout)
 *   pro : optional GroupNorm-on-load  a' = (a - mean[s]) * rstd[s] * gamma[k] + beta[k],
 *         s = (m / st_div1) * st_m1 + (m % st_div2) * st_m2 + st_base, stats = [S][2]
 *   epi : + bias[n]; act (0 none, 1 tanh, 2 ReLU); * (1 - T^2) if T (act 4: * (T > 0), the ReLU
 *         derivative from the saved output); + R if R                 (T, R addressed like C)
 * Replaces: F.group_norm + F.linear / Conv1d(k=1) (+tanh, +residual) at bsrnn.py:38-46,
 * 252-258, 271-282 and their autograd data-gradients.                                      */
typedef struct ws_gemm_nt_args {
  const float* A;
  const float* W;
  const float* bias;
  float* C;
  const float* R;
  const float* T;
  const float* stats;
  const float* gamma;
  const float* beta;
  const ws_group_nt* groups; /* NULL or device array [ngroups] */
  long long a_s1, a_s2, c_s1, c_s2, st_m1, st_m2, st_base;
  int a_div, c_div, st_div1, st_div2;
  int M, N, K, ldw;
  int act, ngroups, max_n, vec; /* max_n: max N over groups; vec bit0: A float4-loadable, bit1: W,
                                   bit2: split-bf16 (hi/lo, 3 MFMAs, fp32 accumulate) products;
                                   bit3 (round 6, with bits 0-2): W is stored TRANSPOSED, W'[n][k] = W[k * ldw + n] (per group:
                                   ws_group_nt.ldw >= N) -- C = A [M, K] x B [K, N] with B as it lies ("NN"): N % 4 == 0 and
                                   ldw % 4 == 0 (16-byte rows of W); no conv view, no norm-on-load                       */
  ws_conv_view conv;            /* conv.on: A is an implicit patch matrix (above), either mode */
} ws_gemm_nt_args;
int ws_gemm_nt(const ws_gemm_nt_args* a, void* stream);

typedef struct ws_group_tn {
  const float* gamma;
  const float* beta;
  long long g_off, a_off, st_base, out_off, bout_off;
  int Nn, Kk, pad0_, pad1_;
} ws_group_tn;

/* Weight gradients: slab[split][out_off + n*Kk + k] = sum_{m in split} G[m][n] * pro(A[m'][k]),
 * bslab[split][bout_off + n] = sum_m G[m][n]; the caller then sums the splits with
 * ws_reduce_slabs (deterministic, no atomics).  m' = m + shift_rows with the row zeroed
 * when the step index ((m / seq_div) % seq_len) +/- 1 leaves [0, seq_len) (h_{t-1} for dW_hh).
 * Replaces the autograd weight-gradient of the same reference lines as ws_gemm_nt.         */
typedef struct ws_gemm_tn_args {
  const float* G;
  const float* A;
  float* slab;
  float* bslab;              /* NULL: no bias gradient */
  const float* stats;
  const float* gamma;
  const float* beta;
  const ws_group_tn* groups; /* NULL or device array [ngroups] */
  long long g_s1, g_s2, a_s1, a_s2, st_m1, st_m2, st_base;
  long long slab_stride, bslab_stride, out_off, bout_off;
  int g_div, a_div, st_div1, st_div2;
  int M, Nn, Kk, rows_per_split, nsplit;
  int shift_rows, seq_div, seq_len;
  int ngroups, max_n, max_k, vec; /* vec bit0: A float4-loadable, bit2: split-bf16 products */
  ws_conv_view conv;              /* conv.on: A is an implicit patch matrix, mode 0 only (Kk = k*k*C) */
} ws_gemm_tn_args;
int ws_gemm_tn(const ws_gemm_tn_args* a, void* stream);

/* Weight gradient of a convolution with at most 32 output channels, one pass over the activation (conv_wgrad.hip):
 *   slab[split][n * Kk + kk] = sum_{m in split} G[m * ldg + n] * P[m][kk],   bslab[split][n] = sum_m G[m * ldg + n]
 * P = the mode-0 implicit patch matrix `conv` of the image X (Kk = k*k*C, C % 4 == 0, k <= 5; a workgroup owns up to 768
 * columns, wider matrices run one pass per 768-column chunk); a split is tiles_per_split tiles of 32 consecutive rows.  Same result as ws_gemm_tn with conv.on, which sends every 128-column
 * slice of P to another workgroup and so streams X k*k times; this one owns all of P's columns per tile.
 * Replaces autograd's weight gradient of F.conv2d / F.conv_transpose2d (wesep/modules/dpccn/convs.py:28-110).     */
typedef struct ws_conv_wgrad_args {
  const float* G;      /* [M][ldg] output gradient (conv2d) or the layer input (conv_transpose2d) */
  const float* X;      /* channels-last image the patch view is taken of */
  float* slab;
  float* bslab;        /* or NULL */
  long long ldg, slab_stride, bslab_stride;
  int M, Nn, nsplit, tiles_per_split;
  ws_conv_view conv;
} ws_conv_wgrad_args;
int ws_conv_wgrad(const ws_conv_wgrad_args* a, void* stream);

/* out[(i / w) * ldo + (i % w)] = sum_s slab[s * stride + i],  i < count                    */
int ws_reduce_slabs(const float* slab, int nsplit, long long stride, long long count,
                    float* out, int w, long long ldo, void* stream);

/* dst[c][r] = src[r][c]  (src [rows][cols] with leading dim lds)                           */
int ws_transpose(const float* src, int rows, int cols, long long lds, float* dst, void* stream);

/* ---- GroupNorm(1, C, eps) pieces (bsrnn.py:26,256,275; eps = FLT_EPSILON) ---------------
 * A "group" g covers L rows x W contiguous floats:
 *   base(g) = (g / gdiv) * gs1 + (g % gdiv) * gs2 (+ band_off[g % gdiv]),  row stride rs,
 *   W = band_w ? band_w[g % nbands] : W.                                                   */
typedef struct ws_groups_geom {
  const int* band_w;   /* device int[nbands], or NULL */
  const int* band_off; /* device int[nbands], or NULL */
  long long gs1, gs2, rs;
  int ngroups, gdiv, L, W; /* W: width (max width when band_w is given), <= 128 */
  int nbands, pad_;        /* band of group g = g % nbands (per-band widths / gammas) */
} ws_groups_geom;
int ws_group_stats(const float* x, const ws_groups_geom* geo, float eps, float* stats, void* stream);

/* GroupNorm backward, two passes over the same geometry:
 *   pass 1  ab[g] = (mean_g(dxn*gamma), mean_g(dxn*gamma*xhat))
 *   pass 2  dx = rstd*(dxn*gamma - ab0 - xhat*ab1) (+ res)          (dx may alias dxn)
 * gamma_tab: NULL -> `gamma` (per column); else device table of per-band pointers indexed
 * by g % nbands.                                                                            */
int ws_gn_bwd_reduce(const float* x, const float* dxn, const float* stats, const float* gamma,
                     const float* const* gamma_tab, const ws_groups_geom* geo, float* ab,
                     void* stream);
int ws_gn_bwd_apply(const float* x, const float* dxn, const float* stats, const float* ab,
                    const float* gamma, const float* const* gamma_tab, const float* res,
                    const ws_groups_geom* geo, float* dx, void* stream);
/* dgamma/dbeta partials: slab[split][band][2][W]; band b owns groups {g : g % nbands == b}
 * (nbands = 1: every group).  Caller reduces the splits.                                    */
int ws_gn_param_grad(const float* x, const float* dxn, const float* stats,
                     const ws_groups_geom* geo, int nsplit, float* slab, void* stream);

/* The three calls above in ONE pass for small single-band groups (the band view of ResRNN: L = K = 32 rows of
 * W = 128 floats per group; L even, <= 32): a wave owns a group in registers.
 *   dx = rstd * (dxn*gamma - mean(dxn*gamma) - xhat * mean(dxn*gamma*xhat)) (+ res);
 *   pslab[wg][0][c] / [wg][1][c] = this workgroup's share of dgamma[c] / dbeta[c] (sum over wg = the gradient).
 * Replaces autograd's GroupNorm backward of ResRNN.norm (bsrnn.py:26,39).                                  */
int ws_gn_bwd_fused(const float* x, const float* dxn, const float* stats, const float* gamma,
                    const float* res, const ws_groups_geom* geo, int nwg, float* dx, float* pslab,
                    float* pout, unsigned* counter, void* stream);
/* ABI v19: the same with d(xn) arriving as TWO addends (dxn + dxn2: the per-direction buffers ws_lstm_bwd writes through
 * ws_lstm_args.dxn); dxn2 = NULL is ws_gn_bwd_fused.                                                                */
int ws_gn_bwd_fused2(const float* x, const float* dxn, const float* dxn2, const float* stats, const float* gamma,
                     const float* res, const ws_groups_geom* geo, int nwg, float* dx, float* pslab,
                     float* pout, unsigned* counter, void* stream);
/* ABI v15: pout [2][128] (optional) = (dgamma, dbeta) summed over the workgroups BY the workgroups of the launch (the last
 * finisher of every 32 consecutive workgroups adds their shares up in index order, the last of those the group sums:
 * deterministic; no ws_reduce_slabs launch behind it).  pslab then needs nwg + ceil(nwg / 32) rows of [2][128];
 * `counter`: 1 + ceil(nwg / 32) device words that are 0 at launch (the kernel leaves them at 0).
 * ws_gn_bwd_apply_pg: pass 2 of the two-pass form for single-band groups of 128-float rows (the time view of ResRNN.norm)
 * WITH the parameter sums: one partial [2][128] per group to pslab [ngroups][2][128], summed into pout [2][128] by the last
 * workgroup -- replaces ws_gn_bwd_apply + ws_gn_param_grad + ws_reduce_slabs there.                                */
int ws_gn_bwd_apply_pg(const float* x, const float* dxn, const float* stats, const float* ab,
                       const float* gamma, const float* res, const ws_groups_geom* geo, float* dx,
                       float* pslab, float* pout, unsigned* counter, void* stream);

/* ---- bidirectional LSTM recurrence (nn.LSTM inside ResRNN, bsrnn.py:27-33,40) -----------
 * Sequence s, step t lives at row  p = (s / sq_div) * sq_s1 + (s % sq_div) * sq_s2 + t * step_rows.
 * gates [P][2][4H] holds x-projection + biases on entry (gate order i,f,g,o) and the
 * ACTIVATED gates on exit; cbuf/hcat [P][2H] (dir-major halves).  H = 256 only.            */
typedef struct ws_lstm_args {
  float* gates;
  float* cbuf;
  float* hcat;
  const float* dhcat;   /* bwd only: dL/dhcat [P][2H]                         */
  const float* wpack;   /* ws_lstm_pack output for this pass (fwd or bwd)     */
  long long sq_s1, sq_s2, step_rows;
  int nseq, sq_div, L, mode;   /* WS_LSTM_* below; must match the mode wpack was packed with */
  const unsigned* run_if;      /* split-bf16 ws_lstm_fwd and blocked-layout ws_lstm_bwd: optional device word; when given,
                                  the launch does nothing unless *run_if != 0 at kernel start (the predicated fall-back
                                  behind ws_lstm_fwd_cluster / ws_lstm_bwd_pair, see there)               */
  /* ABI v15, blocked-layout modes only: storage format of the saved recurrence state (WS_GATES_* below).  With
   * gfmt != 0 `gates` is a BLH(2 * 4H) buffer of 2-byte elements; the forward reads the fp32 pre-activations from
   * `gates_in` (BL(2 * 4H) fp32, not modified) instead of `gates`; the backward with WS_GATES_H2S writes d(gates)
   * to `dgates` (BL(2 * 4H), BLS elements) instead of in place; with WS_GATES_H2 `dgates` is optional: given, it
   * receives the bf16 d(gates) (BLH(2 * 4H)) and the saved gates stay intact -- what makes a BPTT launch repeatable
   * (the predicated fall-back behind ws_lstm_bwd_pair).                                                    */
  const float* gates_in;
  float* dgates;
  int gfmt;
  int rfmt;              /* ABI v18 (the former pad_: 0 = every earlier behaviour), ws_lstm_bwd with mode WS_LSTM_BF16X3_BLK and
                            WS_GATES_H2F only: 2 = the recurrent product d(h) = d(gates) W_hh on v_mfma_f32_32x32x16_f16 with the
                            STORED scaled-fp16 d(gates) as its one operand against W_hh as fp16 hi + scaled-FP8 lo of 256 w --
                            `wpack` from ws_lstm_pack_bwd_f8: two MFMAs per product instead of three, 96 instead of 128 KB of
                            weights streamed per wave and step (the arithmetic of ws_lstm_pair_args.rfmt = 2).  3 (ABI v20):
                            the same pack and stream with the lo term on v_mfma_scale_f32_32x32x64_f8f6f4 (the codes as A
                            operands of K = 64 against e4m3 of d(gates) / 256): four fp16 MFMAs + one FP8 MFMA per 64 gate
                            columns instead of eight fp16 MFMAs; not with `dxn`                                          */
  const unsigned* amax;  /* WS_GATES_H2F, backward: max |dhcat| of this launch as float bits (see WS_GATES_H2F) */
  /* ABI v19 (three trailing fields, zero = every earlier behaviour), ws_lstm_bwd with rfmt = 2 only: d(normalised input) of
   * the ResRNN computed INSIDE the BPTT from the d(gates) image its recurrent product reads anyway (autograd's d(input) of
   * nn.LSTM, bsrnn.py:40) -- ws_gemm_b2p(a_fmt = 2) over d(gates), a 2.1 GB read per band-view layer at R = 32, is then not
   * launched.  dxn: plain rows [P][128] of direction 0, direction 1 at dxn + dxn_dir_stride floats; the row of (sequence,
   * step) is the ws_seqmap formula on this struct's sq_s1 / sq_s2 / sq_div / step_rows (which the blocked-layout modes
   * otherwise ignore); padded slots (sequence >= nseq) store nothing.  The consumer adds the two directions
   * (ws_gn_bwd_fused2).  wxpack: ws_lstm_pack_dx_f8 output.                                                           */
  float* dxn;
  long long dxn_dir_stride;
  const float* wxpack;
} ws_lstm_args;
/* Storage format of the saved activated gates and of d(pre-activation gates) on the blocked layout (ABI v15).
 * BLH(C): the BL(C) index formula with 2-byte elements -- element (b, slot i, column c) at 2-byte index
 * b*32*C + ((c >> 2)*32 + i)*4 + (c & 3); a lane's 4-column cell is 8 bytes, 32 lanes 256 contiguous bytes.
 *   WS_GATES_F32 (0): gates fp32 BL, d(gates) as BLS pairs IN PLACE (ABI <= 14: one 16E-byte buffer, three lives).
 *   WS_GATES_H2  (1): activated gates as unorm16 in BLH -- i, f, o in (0, 1): u = floor(x * 65535 + 0.5);
 *                     g in (-1, 1): u = floor((x + 1) * 32767.5 + 0.5) -- absolute error <= 7.7e-6 / 1.6e-5 where fp16
 *                     would leave 2.4e-4 near saturation; d(gates) IN PLACE as bf16 = the hi term of the split pair
 *                     (relative error 2^-9: consumers ws_gemm_b2p a_fmt = 1, ws_gemm_tnb g_fmt = 1).  Halves the
 *                     largest buffer of the step and every pass over it.
 *   WS_GATES_H2S (2): activated gates as in H2; d(gates) as BLS pairs (full split precision) to a separate
 *                     BL(2 * 4H) buffer.                                                                      */
#define WS_GATES_F32 0
#define WS_GATES_H2 1
#define WS_GATES_H2S 2
/*   WS_GATES_H2F (3): activated gates as in H2; d(gates) as SCALED fp16, in place or to `dgates` like H2:
 *                     stored = fp16(x * S),  S = ws_dgates_scale(*amax) = the power of two that puts max |d(hcat)| of the
 *                     launch -- the word `amax` (float bits), raised by the ws_gemm_p2b launch that produced d(hcat)
 *                     (ws_gemm_p2b_args.amax; the caller zeroes it) -- into [2^WS_DGATES_EXP, 2^(WS_DGATES_EXP + 1)):
 *                     2^7 of headroom for what the BPTT accumulates on top of d(hcat), and everything down to 2^-22 of
 *                     the largest keeps fp16's 11 bits -- 8x finer than bf16, at the same 2 bytes.  NOT clamped (ABI v17;
 *                     v15 / v16 stored fmed3(x * S, +-65504) and put the maximum into [2^10, 2^11)): a value beyond fp16's
 *                     range is stored as +-Inf, a NaN as NaN -- the consumers turn them into non-finite gradients, which
 *                     ws_grad_norms' guard catches: the optimizer step is skipped and counted instead of taken on
 *                     silently clipped gradients.  Consumers (ws_gemm_b2p a_fmt = 2, ws_gemm_tnb g_fmt = 2) read `amax`
 *                     themselves, feed the fp16 values to v_mfma_f32_32x32x16_f16 as they are and undo S (exact) in their
 *                     epilogues.  The default of the Python path.                                                     */
#define WS_GATES_H2F 3
#define WS_DGATES_EXP 8
#define WS_LSTM_F32_MT1 1 /* exact-fp32 MFMA, 16 sequences per workgroup                       */
#define WS_LSTM_F32_MT2 2 /* exact-fp32 MFMA, 32 sequences per workgroup                       */
#define WS_LSTM_BF16X3 3  /* split-bf16 (hi/lo, 3 bf16 MFMAs per product, fp32 accumulate), 32 */
#define WS_LSTM_BF16X3_BLK 4 /* as 3, but gates / cbuf / hcat / dhcat are in the blocked layout BL
                                (below): block b = tile * L + step, tile = 32 consecutive sequences;
                                the sq_* / step_rows fields are ignored.  hcat (fwd) and the d(gates)
                                left in `gates` (bwd) are written in split-bf16 storage BLS (below)  */
#define WS_LSTM_BF16X3_BLK16 5 /* as 4 with 16-sequence workgroups (two per tile): for views with too few
                                  sequences to fill the chip with 32-sequence workgroups         */
#define WS_LSTM_H 256
#define WS_LSTM_PACK_FLOATS (2 * 4 * WS_LSTM_H * WS_LSTM_H) /* per pass, both directions */
/* Packs weight_hh_l0 / _reverse [4H][H] into MFMA fragment order for the fwd and bwd pass
 * (fp32 for WS_LSTM_F32_*, bf16 hi/lo pairs for WS_LSTM_BF16X3; same byte size).           */
int ws_lstm_pack(const float* whh_f, const float* whh_r, float* pack_fwd, float* pack_bwd,
                 int mode, void* stream);
/* ABI v18: the BPTT pack of ws_lstm_args.rfmt = 2 (WS_LSTM_PACK_FLOATS floats like the others; per (direction, wave) region
 * of 128 KB: 16 chunks of 6 KB = four fp16 hi fragments of 256 w + four fragments of e4m3 codes of the remainder over the
 * scale of their group of 8 k-steps; the eight scales as floats at byte 96 K).  |w| < 255.                       */
int ws_lstm_pack_bwd_f8(const float* whh_f, const float* whh_r, float* pack_bwd, void* stream);
/* ABI v19: W_ih^T of both directions for ws_lstm_args.dxn -- wcat [2][4H][128] from ws_lstm_cat_ih; pack: WS_LSTM_DX_PACK_FLOATS
 * floats (16 regions of 48 KB + 64 B: fp16 hi fragments of 256 w for v_mfma_f32_16x16x32_f16, e4m3 codes of the remainder,
 * eight group scales).                                                                                              */
#define WS_LSTM_DX_PACK_FLOATS (16 * (48 * 1024 + 64) / 4)
int ws_lstm_pack_dx_f8(const float* wcat, float* pack, void* stream);
int ws_lstm_fwd(const ws_lstm_args* a, void* stream);
/* On exit gates holds dL/d(pre-activation gates).                                           */
int ws_lstm_bwd(const ws_lstm_args* a, void* stream);
/* Weight-stationary forward recurrence over clusters of 8 co-resident workgroups (blocked layout,
 * same gates / cbuf / hcat contract as WS_LSTM_BF16X3_BLK): W_hh stays in registers, h_t is exchanged
 * through `xchg` each step.  nseq % 64 == 0 and (nseq / 32) * 8 <= CUs of the device (checked).
 * xchg: (nseq / 32) * 128 KB scratch; flags: (nseq / 32) * 8 + 8 words (zeroed by the call on `stream`).
 * Residency of the whole grid is a precondition the launcher can only check against the CU count, not
 * against CUs held by other streams / processes, so every wait is bounded.  If one times out the outputs
 * are NaN-poisoned AND the launch's own timeout word  flags[(nseq / 32) * 8]  is set to 1 (it is 0 after a
 * clean launch); `status` (optional, sticky, never cleared by the library) is set to 1 as well.  Callers
 * enqueue the streaming kernels behind this launch with  run_if = &flags[(nseq / 32) * 8]  (ws_gemm_p2b to
 * rebuild the pre-activations, then ws_lstm_fwd): they cost an empty launch after a clean run and redo the
 * layer after a timeout, without a host round trip.                                               */
typedef struct ws_lstm_cluster_args {
  float* gates;
  float* cbuf;
  float* hcat;          /* fwd: output */
  const float* dhcat;   /* bwd: dL/dhcat, BL(512) */
  const float* whh_f;   /* weight_hh_l0          [4H][H] fp32 */
  const float* whh_r;   /* weight_hh_l0_reverse  [4H][H] fp32 */
  void* xchg;
  unsigned* flags;
  unsigned* status;
  int nseq, L;
  int dbg, gfmt;        /* dbg: probes / tests only: 1 skip the flag wait, 2 skip the gather, 4 skip the publish,
                           8 force a timeout in workgroup 0 at step 2 (exercises the fall-back).
                           gfmt (ABI v15, forward only; the cluster BPTT is WS_GATES_F32 only): WS_GATES_*; != 0: the
                           pre-activations come from gates_in (fp32 BL), `gates` receives unorm16 BLH      */
  const float* gates_in;
} ws_lstm_cluster_args;
int ws_lstm_fwd_cluster(const ws_lstm_cluster_args* a, void* stream);
/* Second-generation cluster forward (lstm_cluster2.hip; ABI v17): the same clusters, outputs and residency rules, for
 * WS_GATES_H2-style storage only (`gates` = unorm16 BLH(2 * 4H) out, cbuf fp32 BL, hcat BLS), with
 *   - the x-projection computed in the kernel from the normalised input as split pairs (xn: BL(128) of BLS elements, what
 *     ws_gemm_p2b writes as A_bl) against wcat [2][4H][128] (ws_lstm_cat_ih) and bcat [2][4H], the full three-term split-bf16
 *     product: no pre-activation buffer, no x-projection GEMM (bsrnn.py:39-40 fused into the recurrence);
 *   - the recurrent product on v_mfma_f32_32x32x16_f16: h_t as one fp16 operand, W_hh as fp16 hi / lo of 256 w;
 *   - a data-tagged hand-off (bit 14 of every fp16 h carries the step's tag; no flags).
 * xchg: (nseq / 32) * 64 KB scratch (filled by the call on `stream`); tword: the launch's time-out word (zeroed by the
 * call; 0 after a clean launch; callers enqueue ws_gemm_p2b + ws_lstm_fwd with run_if = tword behind the launch);
 * status: optional, sticky.  dbg (probes / tests): 1 skip the wait, 4 skip the publish, 8 force a time-out in workgroup 0
 * at step 2, 2048 cycle stamps into dbg_buf.                                                                                                       */
typedef struct ws_lstm_cluster2_args {
  float* gates;
  float* cbuf;
  float* hcat;
  const float* xn;
  const float* wcat;
  const float* bcat;
  const float* whh_f;
  const float* whh_r;
  void* xchg;
  unsigned* tword;
  unsigned* status;
  int nseq, L;
  int dbg, rfmt;        /* rfmt (ABI v20, the former pad_: 0 = every earlier behaviour): 1 = the lo term of the recurrent product on
                           v_mfma_scale_f32_32x32x64_f8f6f4 (e4m3 codes of 256 w - hi, one exponent per wave, against e4m3 of h): four
                           fp16 MFMAs + one FP8 MFMA per 64 columns instead of eight fp16 MFMAs                        */
  float* dbg_buf;       /* dbg 2048: L * 2 * 8 64-bit cycle stamps of cluster 0 / member 0 (tools/r05_recur_probe.py) */
} ws_lstm_cluster2_args;
int ws_lstm_fwd_cluster2(const ws_lstm_cluster2_args* a, void* stream);
/* BPTT over the same clusters (reduce-scatter of partial dh each step): gates holds the activated
 * gates on entry and d(pre-activation gates) on exit; xchg: (nseq / 32) * 1 MB; flags / status as above
 * (in place on `gates`: there is no device-side fall-back, the caller checks the timeout word).  */
int ws_lstm_bwd_cluster(const ws_lstm_cluster_args* a, void* stream);
/* BPTT over PAIRS of workgroups (lstm_pair.hip; ABI v9): the two workgroups of a (32-sequence tile, direction) split
 * W_hh by gate rows -- each keeps the hi plane of its 512 rows in registers, streams only the lo plane, multiplies its
 * own d(gates) into a partial dh for all 256 units and hands the partner's half over (16 KB per step through `xchg`).
 * Same gates / cbuf / dhcat contract as ws_lstm_bwd(WS_LSTM_BF16X3_BLK): blocked layout, any nseq (padded slots),
 * gates = activated gates on entry, d(pre-activation gates) as BLS on exit -- replaces autograd through nn.LSTM for
 * the time view (bsrnn.py:38-46) on half of the chip's CUs.  wpack: ws_lstm_pack_pair output (WS_LSTM_PACK_FLOATS).
 * npair = 2 * ceil(nseq / 32); 2 * npair <= CUs of the device (checked).  xchg: npair * 64 KB scratch; flags:
 * npair * 8 + 8 words (zeroed by the call on `stream`).  Every wait is bounded: on a timeout d(gates) are NaN-poisoned,
 * flags[npair * 8] (0 after a clean launch) and *status (optional, sticky) are set to 1.  In place there is no
 * device-side repair; with `dgates` given (WS_GATES_H2 / H2S: the saved gates stay intact) callers enqueue the streaming
 * BPTT behind this launch with  run_if = &flags[npair * 8]  and the same `dgates`: an empty launch after a clean run,
 * the whole BPTT again after a timeout -- no NaN reaches a consumer, no host round trip (ABI v15).
 * dbg (probes / tests only): 1 skip the flag wait, 2 skip the exchange, 4 no weight reloads,
 * 8 force a timeout in pair 0 at step 2, 32 no wave priorities,
 * 64 full agent-scope release / acquire fences around the hand-off, 2048 (WS_GATES_F32 / H2F) cycle stamps of pair 0
 * into dbg_buf (tools/pair_diag.py --ts).                                                                         */
typedef struct ws_lstm_pair_args {
  float* gates;
  const float* cbuf;
  const float* dhcat;
  const float* wpack;
  void* xchg;
  unsigned* flags;
  unsigned* status;
  float* dbg_buf;       /* NULL, or npair * L * 2 * 2 * 4096 floats: per (pair, step, member) the partial it sent and
                           the partial it received (diagnosis only, tools/pair_diag.py)                       */
  int nseq, L;
  int dbg, gfmt;        /* gfmt (ABI v15): WS_GATES_*; H2 / H2F: unorm16 gates in, bf16 / scaled-fp16 d(gates) out -- to `dgates`
                           (BLH) when given, else in place; H2S: d(gates) as BLS pairs to `dgates`                     */
  float* dgates;
  const unsigned* amax; /* WS_GATES_H2F: max |dhcat| of this launch as float bits */
  int rfmt, pad_;       /* ABI v17: arithmetic of the recurrent product d(h) = d(gates) W_hh.  0: split-bf16, three
                           v_mfma_f32_32x32x16_bf16 per product (wpack from ws_lstm_pack_pair); 1 (WS_GATES_H2F only): the
                           STORED scaled-fp16 d(gates) as one operand of v_mfma_f32_32x32x16_f16 against W_hh as fp16 hi / lo
                           of 256 w (wpack from ws_lstm_pack_pair_f16): two MFMAs per product, what the kernel writes is
                           what its own recurrence and both consumers read.  2 (ABI v18; WS_GATES_H2F only): the same
                           product with the lo plane of W_hh as block-scaled FP8 (wpack from ws_lstm_pack_pair_f8): 16
                           instead of 22 significant bits of every weight, and the whole of W_hh stays on the compute
                           unit for the launch (hi plane in registers, lo plane in LDS) -- nothing of it is streamed.
                           3 (ABI v20; WS_GATES_H2F only): rfmt 2 with the lo term on v_mfma_scale_f32_32x32x64_f8f6f4 --
                           the same codes as A operands of K = 64 (wpack from ws_lstm_pack_pair_f8mx) against e4m3 of
                           d(gates) / 256 built in registers from the fp16 fragments: per 64 gate columns four fp16 MFMAs
                           + one FP8 MFMA at twice the rate instead of eight fp16 MFMAs                               */
} ws_lstm_pair_args;
int ws_lstm_pack_pair(const float* whh_f, const float* whh_r, float* pack, void* stream);
/* ABI v17: the pack of rfmt = 1 (same size and unit order, fp16 hi / lo of 256 w; |w| < 255) */
int ws_lstm_pack_pair_f16(const float* whh_f, const float* whh_r, float* pack, void* stream);
/* ABI v18: the pack of rfmt = 2 (same size: 32 blocks of 64 KB, one per (direction, half, wave); per block the fp16 hi plane
 * of 256 w exactly as ws_lstm_pack_pair_f16 writes it (32 KB), then the lo plane as OCP e4m3 codes of (256 w - hi) / S, 8
 * bytes per lane and k-step (16 KB), then S as one float at byte 48 K; S = 2^(e - 20) for the block's max |256 w| in
 * [2^(e-1), 2^e): the largest possible remainder maps to 256 -- e4m3 has no saturation, it overflows to NaN above 448.
 * |w| < 255.) */
int ws_lstm_pack_pair_f8(const float* whh_f, const float* whh_r, float* pack, void* stream);
/* ABI v20: the pack of rfmt = 3 -- hi plane, codes and S of ws_lstm_pack_pair_f8; the codes as eight 2 KB operand fragments
 * (K = 64) per block: a lane's 32 bytes = its four 8-byte units of k-steps 4 kb .. 4 kb + 3, as two 16-byte pieces 1 KB apart;
 * the E8M0 byte of S as an int at byte 48 K + 4 */
int ws_lstm_pack_pair_f8mx(const float* whh_f, const float* whh_r, float* pack, void* stream);
int ws_lstm_bwd_pair(const ws_lstm_pair_args* a, void* stream);
/* wcat[2][4H][N] <- (w_ih_f, w_ih_r);  bcat[2][4H] <- b_ih + b_hh per direction             */
int ws_lstm_cat_ih(const float* wih_f, const float* wih_r, const float* bih_f, const float* bhh_f,
                   const float* bih_r, const float* bhh_r, int n_in, float* wcat, float* bcat,
                   void* stream);

/* ---- blocked layout BL and the GEMMs around the recurrence (bsrnn.py:38-46) ----------------
 * BL(C): a [rows][C] matrix whose rows are grouped in blocks of 32; block b = tile * L + step
 * holds the 32 consecutive sequences of an LSTM workgroup (`tile`) at one `step`;
 *   element (b, slot i, column c)  at  b*32*C + ((c >> 2)*32 + i)*4 + (c & 3).
 * Slots whose sequence index tile*32 + i >= nseq are padding: producers write zeros there.
 * Split-bf16 storage BLS (ABI v8): the BL buffers that are only ever consumed as MFMA operands -- hcat (h of the
 * blocked-layout recurrences), the d(pre-activation gates) that BPTT leaves in `gates`, and A_bl of ws_gemm_p2b
 * (normalised input / incoming gradient) -- hold each 4-byte element as the two terms of the split product:
 *   bits 31..16 = bf16 hi = bf16(x),  bits 15..0 = bf16 lo = bf16(x - hi),  value = hi + lo  (|error| <= 2^-17 |x|).
 * Producers have both terms at hand (they feed them to their own MFMAs); consumers (ws_gemm_b2p's A, all operands
 * of ws_gemm_tnb, xn of ws_lstm_fwd_fused) rebuild fragments with byte permutes instead of converting again -- the
 * products are bit-identical to splitting an fp32 copy.  Pre-activations / activated gates, cbuf and dhcat stay fp32.
 * ws_seqmap maps a slot to its position (row) in the plain Z-layout tensors:
 *   pos = (seq / sq_div) * sq_s1 + (seq % sq_div) * sq_s2 + step * step_rows.              */
typedef struct ws_seqmap {
  long long sq_s1, sq_s2, step_rows;
  int nseq, sq_div, L;
  int nvalid;   /* ABI v15: 0 = every sequence < nseq maps to rows; else sequences >= nvalid are PADDING (zeros on the way
                   into BL, dropped on the way out): nseq then only fixes the number of 32-sequence tiles -- strided maps
                   (TF-GridNet's inter-frame path in place, no transposed copy) whose sequence count the cluster
                   recurrence wants rounded up to a multiple of 64                                                  */
} ws_seqmap;

/* out (bf16 pairs, N*K*4 bytes) <- W'[n][k] = trans ? W[k*ldw + n] : W[n*ldw + k], split into
 * bf16 hi/lo and ordered for ws_gemm_p2b (order 0) or ws_gemm_b2p (order 1).                */
int ws_pack_w(const float* W, int N, int K, long long ldw, int trans, int order, float* out,
              void* stream);
/* The same units with fp16 hi / lo of 256 * W' (ABI v15): the weight operand of ws_gemm_b2p with a_fmt = 2 (whose A
 * operand, the scaled-fp16 d(gates) of WS_GATES_H2F, then feeds v_mfma_f32_32x32x16_f16 without conversion).   */
int ws_pack_w_f16(const float* W, int N, int K, long long ldw, int trans, int order, float* out,
                  void* stream);
/* ABI v20: the weight operand of ws_gemm_b2p with a_fmt = 3 (N = 128, K % 64 == 0; fits the N*K*4 bytes of the other packs):
 * per stage of 64 k the sixteen fp16 hi fragments of ws_pack_w_f16, then per column tile one 2 KB operand fragment of
 * v_mfma_scale_f32_32x32x64_f8f6f4 with the e4m3 codes of the residuals (one exponent per fragment); the exponents (E8M0,
 * one dword per stage) behind the last stage                                                                          */
int ws_pack_w_f16f8(const float* W, int N, int K, long long ldw, int trans, float* out, void* stream);

/* plain -> BL:  C[(b,i)][n] = sum_k pro(A[pos(b,i)][k]) * W'[n][k] + bias[n]   (K = 128, N % 64 == 0)
 * pro = optional GroupNorm-on-load as in ws_gemm_nt (stat index computed from pos).  If A_bl is
 * given, the (normalised) operand is also written in BL(K), as BLS elements.  Replaces F.group_norm + the
 * nn.LSTM input projection (bsrnn.py:39-40) and autograd's d(hcat) of proj (bsrnn.py:42-44). */
typedef struct ws_gemm_p2b_args {
  const float* A;
  const float* Wpack;
  const float* bias;
  float* C;          /* BL(N) */
  float* A_bl;       /* BL(K) or NULL */
  const float* stats;
  const float* gamma;
  const float* beta;
  ws_seqmap sm;
  long long lda, st_m1, st_m2, st_base;
  int st_div1, st_div2, N, K;
  const unsigned* run_if;   /* optional device word: the launch does nothing unless *run_if != 0 */
  unsigned* amax;           /* optional device word (ABI v15): raised (atomic max on the float bits) to max |C| of this
                               launch; the caller zeroes it.  The scale source of WS_GATES_H2F                     */
  void* A_bl16;             /* optional (ABI v16): the (normalised) operand once more as fp16 elements in BLH(K) -- the
                               2-byte A operand of ws_gemm_tnb (a_fmt = 1)                                          */
} ws_gemm_p2b_args;
int ws_gemm_p2b(const ws_gemm_p2b_args* a, void* stream);

/* BL -> plain:  C[pos(b,i)][n] = sum_k A[(b,i)][k] * W'[n][k] + bias[n] + R[pos][n]   (N = 128, K % 64 == 0)
 * Replaces ResRNN.proj + residual (bsrnn.py:42-46) and autograd's d(normalised input).      */
typedef struct ws_gemm_b2p_args {
  const float* A;    /* BL(K), BLS elements (hcat of the recurrences / d(gates) of BPTT) */
  const float* Wpack;
  const float* bias; /* or NULL */
  const float* R;    /* plain, addressed like C, or NULL */
  float* C;          /* plain rows, leading dimension ldc */
  ws_seqmap sm;
  long long ldc;
  int N, K;
  int a_fmt, pad_;   /* ABI v15: 0 = A holds BLS pairs (BL(K)); 1 = A holds bf16 elements (BLH(K)): d(gates) of WS_GATES_H2;
                        2 = A holds scaled fp16 elements (BLH(K)): d(gates) of WS_GATES_H2F, scale from `amax`; Wpack
                        is then a ws_pack_w_f16 pack.  3 (ABI v20): as 2 with the lo term of the product on
                        v_mfma_scale_f32_32x32x64_f8f6f4 (Wpack from ws_pack_w_f16f8; e4m3 of A / 256 built in registers) */
  const unsigned* amax;
  void* a16_out;     /* optional (ABI v16, a_fmt 0 only): the A operand once more as fp16 elements in BLH(K) -- every block
                        is read by exactly one wave here, so the copy costs no extra read (hcat -> ws_gemm_tnb a_fmt = 1) */
} ws_gemm_b2p_args;
int ws_gemm_b2p(const ws_gemm_b2p_args* a, void* stream);

/* BL x BL -> weight gradients, one pass over G:
 *   slab[split][g][a] = sum_{b in split} sum_i G[(b,i)][g_off + g] * Acat[(b + shift,i)][a]
 *   bslab[split][g]   = sum G[(b,i)][g_off + g]                                  (if bslab)
 * Acat = columns [a0_off, a0_off + a0_cols) of A0 (BL(a0_width), shifted by a0_shift steps
 * inside the tile, zero outside [0, L)) followed by a1_cols columns of A1 likewise; Acat has 128
 * or 384 columns, g_cols is a multiple of 128.  aslab[split][a] (optional) = column sums of Acat.
 * G, A0 and A1 hold BLS elements; a0_shift must be 0 (only A1 is ever the step-shifted h).
 * Replaces autograd's dW_ih / dW_hh / db (nn.LSTM) and dW_proj / db_proj.                       */
typedef struct ws_gemm_tnb_args {
  const float* G;
  const float* A0;
  const float* A1;   /* or NULL */
  float* slab;
  float* bslab;      /* or NULL */
  float* aslab;      /* or NULL */
  long long slab_stride, bslab_stride, aslab_stride;
  int g_width, g_off, g_cols;
  int a0_width, a0_off, a0_cols, a0_shift;
  int a1_width, a1_off, a1_cols, a1_shift;
  int nblk, L, nsplit, blocks_per_split;
  int g_fmt;         /* ABI v15: 0 = G holds BLS pairs; 1 = G holds bf16 elements (BLH(g_width)): d(gates) of WS_GATES_H2;
                        2 = scaled fp16 elements: d(gates) of WS_GATES_H2F, scale from `amax` (slab / bslab come out unscaled) */
  const unsigned* amax;
  int a_fmt, pad_;   /* ABI v16: 0 = A0 / A1 hold BLS pairs; 1 = fp16 elements (BLH(a0_width) / BLH(a1_width)), g_fmt = 2 only:
                        scaled-fp16 G times fp16 A is ONE v_mfma_f32_32x32x16_f16 per product, and a workgroup loads 32 KB
                        per block instead of 56 (the kernel is bound by the CU's global-load issue rate)                */
} ws_gemm_tnb_args;
int ws_gemm_tnb(const ws_gemm_tnb_args* a, void* stream);

/* ---- STFT / iSTFT (torch.stft / torch.istft at bsrnn.py:309-316, 382-389) ---------------
 * n_fft = 512, hop = 128, periodic Hann, center + reflect pad.  Band-split spectrogram
 * layout xbs [R*Tf][2*F]: band g (first bin f0, width bw) occupies columns
 * [2*f0, 2*f0 + 2*bw) as [re(bw) | im(bw)]  == the reference's subband_spec (bsrnn.py:319-328).
 * band_of_bin: device int[F] -> band id; band_f0/band_bw: device int[nband].               */
typedef struct ws_bands {
  const int* band_of_bin;
  const int* band_f0;
  const int* band_bw;
  int nband, nbins;
} ws_bands;
int ws_stft_bandsplit(const float* wav, int R, int T, const ws_bands* b, float* xbs, void* stream);
/* mask3 [R*Tf][4*F]: band g occupies columns [4*f0, 4*f0+4*bw) in the reference's channel
 * order c = glu*2*bw + ri*bw + f (bsrnn.py:366-370).  frames [R*Tf][512] = windowed irfft of
 * (mask * X)  (bsrnn.py:371-381 + the first half of istft).                                */
int ws_mask_istft_frames(const float* xbs, const float* mask3, int R, int Tf, const ws_bands* b,
                         float* frames, void* stream);
/* overlap-add + window-envelope normalisation + centre trim -> wav [R][T]                  */
int ws_istft_ola(const float* frames, int R, int Tf, int T, float* wav, void* stream);
/* backward of the two calls above: dwav [R][T] -> dmask3 [R*Tf][4*F]                        */
int ws_mask_istft_bwd(const float* dwav, const float* xbs, const float* mask3, int R, int Tf,
                      int T, const ws_bands* b, float* dmask3, void* stream);

/* ---- speaker fusion (speaker.py:81-125, norm.py:118-139) ---------------------------------
 * out[p][n] = z[p][n] * (a0 + a[r][n]) + b[r][n],  r = p / rows_per_r; a or b may be NULL. */
int ws_affine_fwd(const float* z, const float* a, const float* b, float a0, long long rows,
                  int rows_per_r, int N, float* out, void* stream);
/* dz_in = dz*(a0+a) (dz_in may be NULL); da_slab[split][r][n] = sum dz*z_in; db_slab likewise sum dz            */
int ws_affine_bwd(const float* dz, const float* z_in, const float* a, float a0, long long rows,
                  int rows_per_r, int N, int nsplit, float* dz_in, float* da_slab, float* db_slab,
                  float* da, float* db, unsigned* counter, void* stream);
/* ABI v15: da / db [R][N] (optional) = the slabs summed over the splits, in split order, by the last workgroup of each
 * row r to finish; `counter`: R device words that are 0 at launch (left at 0).                                    */

/* ---- SI-SDR loss (auraloss.time.SISDRLoss via wesep/utils/losses.py:24-25) ---------------
 * loss = -mean_r 10 log10(|a t|^2 / (|x - a t|^2 + eps) + eps), zero-mean, eps = 1e-8.
 * rowstat [R][8] keeps (mean_x, mean_t, alpha, c1, c2, sisdr_dB, 0, 0) for the backward.   */
int ws_sisdr_fwd(const float* est, const float* tgt, int R, int T, float eps, float* rowstat,
                 float* loss, void* stream);
int ws_sisdr_bwd(const float* est, const float* tgt, const float* rowstat, const float* gout,
                 int R, int T, float* dest, void* stream);

/* ---- per-tensor clip (funcs.py:79-88) + Adam with coupled L2 (train.py:237-238) ----------
 * tab: device array of ws_tensor_ref, one per parameter tensor.                            */
typedef struct ws_tensor_ref {
  float* param;
  float* grad;
  float* exp_avg;
  float* exp_avg_sq;
  long long numel;
} ws_tensor_ref;
/* guard (optional device words; ABI v17: FOUR words, see ws_guard_commit): guard[0] is set to 1 when any tensor's norm is
 * NaN / Inf.  The caller zeroes guard[0] before the first launch of a step (the skip word of that step's
 * ws_clip_adam_step launches).                                                                                       */
int ws_grad_norms(const ws_tensor_ref* tab, int ntensors, float* norms, unsigned* guard, void* stream);
/* ABI v17: closes an optimizer step's book-keeping ON THE DEVICE, after the step's last ws_clip_adam_step launch.  With
 * skipped := guard[0] != 0 || (skip1 && *skip1 != 0):  skipped -> ++guard[1] (steps skipped so far: exact, monotonic),
 * ++guard[2] (consecutive skipped steps), ++guard[3] (the bias-correction lag, below); else guard[2] = 0.  One thread.  */
int ws_guard_commit(unsigned* guard, const unsigned* skip1, void* stream);
/* if clip > 0: coef = clip / (norm + 1e-6); grad *= coef when coef < 1  (per tensor)
 * skip0 / skip1 (optional device words, ABI v15): the launch does NOTHING when one of them is non-zero at kernel start --
 * the guard word of ws_grad_norms (a non-finite gradient: a BPTT launch that timed out, on any rank of a data-parallel
 * job once the gradients are all-reduced) and / or the sticky status word of an in-place BPTT launch.  The reference
 * would apply the NaN to every weight (funcs.py:79-88: NaN comparisons are false, Adam follows).
 * step_lag (optional device word, ABI v17; pass &guard[3]): the bias corrections of Adam use  step - *step_lag  (>= 1), read
 * at kernel start: the host counts every ATTEMPTED step without waiting for the device, the device knows which of them
 * were skipped -- a skipped step must not advance the bias correction (torch.optim.Adam under a GradScaler does not call
 * step() at all).  The host subtracts what it has learnt asynchronously from both sides (FusedClipAdam._poll_guard).  */
int ws_clip_adam_step(const ws_tensor_ref* tab, int ntensors, const float* norms, float clip,
                      float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                      int clip_only, const unsigned* skip0, const unsigned* skip1, const unsigned* step_lag,
                      void* stream);

/* ---- Conv-TasNet / SpEx+, channels-last activations [R*T'][C] ----------
 * Everything that is not a 1x1 / framing convolution (those are ws_gemm_nt / ws_gemm_tn on row views).
 * Replaces, with their autograd: nn.PReLU (convs.py:60,73,123,137), GlobalChannelLayerNorm /
 * ChannelWiseLayerNorm (norm.py:7-59), the depthwise dilated Conv1d (convs.py:63-70,125-133), the
 * concatConv speaker fusion's broadcast half (convs.py:143-148), the decoder's mask product and
 * ConvTranspose1d overlap-add (decoder.py:92-114).                                                   */

/* mean / rstd of `ngroups` contiguous groups of n_per_group floats (gLN: one row's T'*C), chunked over
 * nchunk workgroups per group; scratch [ngroups][nchunk][4]; stats [ngroups][2] = (mean, 1/sqrt(var+eps)) */
int ws_flat_stats(const float* x, int ngroups, long long n_per_group, float eps, int nchunk,
                  float* scratch, float* stats, void* stream);
/* x += rb[m / rows_per_r] (rb may be NULL; x is updated in place = the saved pre-activation);
 * y = x > 0 ? x : a[0] * x                                                                         */
int ws_prelu_fwd(float* x, const float* rb, const float* a, long long rows, int C, int rows_per_r,
                 float* y, void* stream);
/* dx = dy * (pre > 0 ? 1 : a[0]) (dx may alias dy); slab[i < nslab] partial sums of dy * min(pre, 0)  */
int ws_prelu_bwd(const float* pre, const float* dy, const float* a, long long n, float* dx, float* slab,
                 int nslab, void* stream);
/* y = b + depthwise_conv(norm(x)), norm applied on load with stats index m / st_div, zero "same" padding,
 * w [C][P] (odd P <= 7), dilation dil                                                               */
int ws_dwconv_fwd(const float* x, const float* stats, const float* gamma, const float* beta,
                  const float* w, const float* b, int R, int Tp, int C, int P, int dil, int st_div,
                  float* y, void* stream);
/* dxn = d(norm(x)); slab[split][P + 1][C]: rows 0..P-1 = dw[.][p] partials, row P = db partials       */
int ws_dwconv_bwd(const float* dy, const float* x, const float* stats, const float* gamma,
                  const float* beta, const float* w, int R, int Tp, int C, int P, int dil, int st_div,
                  float* dxn, int nsplit, int rows_per_split, float* slab, void* stream);
/* The same pair with a `causal` flag: non-zero puts every tap at or before t (taps t - (P-1-p)*dil), the causal
 * Conv1DBlock of wesep/modules/tasnet/convs.py:61-62,91-92 (padding dil*(P-1), last dil*(P-1) outputs cut)      */
int ws_dwconv_ex_fwd(const float* x, const float* stats, const float* gamma, const float* beta,
                     const float* w, const float* b, int R, int Tp, int C, int P, int dil, int st_div, int causal,
                     float* y, void* stream);
int ws_dwconv_ex_bwd(const float* dy, const float* x, const float* stats, const float* gamma,
                     const float* beta, const float* w, int R, int Tp, int C, int P, int dil, int st_div, int causal,
                     float* dxn, int nsplit, int rows_per_split, float* slab, void* stream);
/* slab[split * ngroups + grp][2][C]: per-channel sums of g and of g * xhat over the rows of group grp
 * (rows_per_group consecutive rows) that fall into the split; xhat = (x - mean_s) * rstd_s, s = m / st_div
 * (stats NULL: xhat = x; x NULL: second row zero)                                                    */
int ws_chan_sums(const float* g, const float* x, const float* stats, int st_div, int rows_per_group,
                 int ngroups, int nsplit, int C, float* slab, void* stream);
/* ab[grp] = (sum_c gamma[c] * S0[grp][c], sum_c gamma[c] * S1[grp][c]) / n_per_group, sums [ngroups][2][C] */
int ws_norm_ab(const float* sums, const float* gamma, int ngroups, int C, long long n_per_group,
               float* ab, void* stream);
/* dx = rstd_s * (dxn * gamma - ab0_s - xhat * ab1_s) (+ res), s = m / st_div (dx may alias dxn)        */
int ws_norm_bwd_apply_cl(const float* x, const float* dxn, const float* stats, const float* ab,
                         const float* gamma, const float* res, long long rows, int C, int st_div,
                         float* dx, void* stream);
/* s = w * m (w with leading dimension ldw); backward: dw = ds * m (leading dimension ld_dw),
 * dm = ds * w * (m > 0)  (m is the ReLU output of the mask GEMM)                                     */
int ws_maskmul_fwd(const float* w, long long ldw, const float* m, long long rows, int N, float* s,
                   void* stream);
int ws_maskmul_bwd(const float* ds, const float* w, long long ldw, const float* m, long long rows, int N,
                   float* dw, long long ld_dw, float* dm, void* stream);
/* d *= (y > 0), in place                                                                            */
int ws_relu_mask(float* d, const float* y, long long n, void* stream);
/* est[r][j] = bias[0] + sum_t frames[r*Tp + t][j - hop*t], j < Tout <= (Tp-1)*hop + L; and the adjoint
 * gather dframes[m][k] = hop*t + k < Tout ? dest[r][hop*t + k] : 0                                     */
int ws_ola_fwd(const float* frames, const float* bias, int R, int Tp, int L, int hop, int Tout,
               float* est, void* stream);
int ws_ola_bwd(const float* dest, int R, int Tp, int L, int hop, int Tout, float* dframes, void* stream);
/* slab[i < nslab] = partial sums of x[0..n)                                                          */
int ws_sum_partial(const float* x, long long n, float* slab, int nslab, void* stream);

/* ---- SpEx+ speaker encoder (wesep/modules/tasnet/speaker.py:7-64), channels-last [M][C] ------------------
 * nn.BatchNorm1d in training mode: stats [2][C] = (batch mean, 1/sqrt(biased var + eps)), two passes;
 * running_mean / running_var (both or neither) are updated with `momentum` (unbiased variance tracked).
 * scratch: nsplit * C floats.                                                                          */
int ws_bn_stats(const float* x, long long M, int C, float eps, float momentum, float* running_mean,
                float* running_var, int nsplit, float* scratch, float* stats, void* stream);
/* u = gamma * (x - mean_c) * rstd_c + beta (+ res);  y = PReLU(u, a[0])                                  */
int ws_bn_prelu_fwd(const float* x, const float* stats, const float* gamma, const float* beta,
                    const float* res, const float* a, long long M, int C, float* u, float* y, void* stream);
/* BatchNorm backward from du: sums [2][C] = (sum du, sum du * xhat) = (dbeta, dgamma);
 * dx = gamma * rstd * (du - sums0/M - xhat * sums1/M) (dx may alias du); slab: nsplit * 2 * C floats     */
int ws_bn_bwd(const float* x, const float* du, const float* stats, const float* gamma, long long M, int C,
              int nsplit, float* slab, float* sums, float* dx, void* stream);
/* nn.MaxPool1d(3) over time on [R][T][C] -> [R][T/3][C]; backward to the first maximal position        */
int ws_maxpool3_fwd(const float* x, int R, int T, int C, float* y, void* stream);
int ws_maxpool3_bwd(const float* x, const float* dy, int R, int T, int C, float* dx, void* stream);
/* out[m][c] = scale * src[m / rows_per_r][c]                                                            */
int ws_bcast_rows(const float* src, float scale, int rows_per_r, long long M, int C, float* out, void* stream);
/* nn.CrossEntropyLoss (mean) on [R][S] logits, int64 labels: loss[0] and dlogits = d loss / d logits     */
int ws_cross_entropy(const float* logits, const long long* label, int R, int S, float* loss, float* dlogits,
                     void* stream);

/* ---- forward recurrence with the input projection fused in (blocked layout, 32-sequence workgroups) --------
 * Replaces ws_gemm_p2b(x-projection) + ws_lstm_fwd(WS_LSTM_BF16X3_BLK) for views with many sequences:
 * xn is the normalised ResRNN input in BL(128); gates (BL(2048)) receives the ACTIVATED gates, cbuf / hcat
 * (BL(512)) as in ws_lstm_fwd; bias [2][4H] = b_ih + b_hh per direction.  wpack: ws_lstm_pack_fused output,
 * WS_LSTM_FUSED_PACK_FLOATS floats ([W_ih | W_hh] as one K = 384 stream of bf16 hi/lo MFMA fragments).       */
typedef struct ws_lstm_fused_args {
  float* gates;
  float* cbuf;
  float* hcat;
  const float* xn;
  const float* wpack;
  const float* bias;
  int nseq, L;
  int gfmt;             /* ABI v15: WS_GATES_* (!= 0: `gates` receives unorm16 BLH) */
  int hfmt;             /* ABI v19 (the former pad_: 0 = every earlier behaviour): 1 = the recurrent part of the stream on
                           v_mfma_f32_32x32x16_f16 with h as ONE fp16 operand and W_hh as fp16 hi / lo of 256 w (two MFMAs per
                           product instead of three; the arithmetic of ws_lstm_fwd_cluster2) -- wpack from
                           ws_lstm_pack_fused_h16, 2-byte gate formats only.  Bit 1 (value
                           2, measurement): the 64-sequence kernels drain every store of a step before the next one starts
                           (the wait of rounds 3-5).  Bit 2 (ABI v20; with bit 0: hfmt = 5): the lo term of that product on
                           v_mfma_scale_f32_32x32x64_f8f6f4 -- the residuals 256 w - hi as e4m3 codes with one exponent per
                           fragment against an e4m3 image of h: four fp16 MFMAs + one FP8 MFMA (K = 64, twice the rate) per
                           recurrent k-step instead of eight; wpack from ws_lstm_pack_fused_h8 (64- and 32-sequence kernels)  */
} ws_lstm_fused_args;
#define WS_LSTM_FUSED_PACK_FLOATS (2 * 8 * 24 * 4 * 2 * 64 * 4)
int ws_lstm_pack_fused(const float* wih_f, const float* wih_r, const float* whh_f, const float* whh_r,
                       float* pack, void* stream);
/* ABI v19: the pack of hfmt = 1 -- same size and unit order; both parts hold 256 w: W_ih as bf16 hi / lo, W_hh as fp16 hi / lo */
int ws_lstm_pack_fused_h16(const float* wih_f, const float* wih_r, const float* whh_f, const float* whh_r,
                           float* pack, void* stream);
/* ABI v20: the pack of hfmt = 5 (fits the same WS_LSTM_FUSED_PACK_FLOATS): per (direction, wave) the W_ih k-steps of the h16
 * pack, then 16 recurrent k-steps of four fp16 hi fragments + one 2 KB FP8 fragment, then the 16 fragment exponents (E8M0)  */
int ws_lstm_pack_fused_h8(const float* wih_f, const float* wih_r, const float* whh_f, const float* whh_r,
                          float* pack, void* stream);
int ws_lstm_fwd_fused(const ws_lstm_fused_args* a, void* stream);

/* ---- wespeaker ResNet speaker encoder (third-party model, call sites
 * wesep/models/bsrnn.py:9,217,352-356), channels-last [R][H][W][C] ------------------------------------------
 * conv2d(k x k, stride s, padding p, bias-free) = ws_im2col + ws_gemm_nt (K = k*k*C); input gradient =
 * ws_gemm_nt + ws_col2im (gather, deterministic); weight gradient = ws_gemm_tn on the patch matrix.
 * patches [R*Ho*Wo][ldp], column (ky*k + kx)*C + c; Ho = (H + 2p - k)/s + 1.  C == 1 or C % 4 == 0
 * (then ldp == k*k*C).  BatchNorm2d / ReLU / residual are the channels-last ws_bn_* / ws_prelu_* entry points. */
int ws_im2col(const float* x, int R, int H, int W, int C, int k, int s, int p, long long ldp, float* patches,
              void* stream);
int ws_col2im(const float* dpatches, int R, int H, int W, int C, int k, int s, int p, float* dx, void* stream);
/* TSTP pooling: x [R][F][T][C] -> stats [R][2][C*F] = (mean over T, sqrt(unbiased var + eps)), feature c*F + f */
int ws_tstp_fwd(const float* x, int R, int F, int T, int C, float eps, float* stats, void* stream);
int ws_tstp_bwd(const float* x, const float* stats, const float* dstats, int R, int F, int T, int C, float* dx,
                void* stream);
/* ASTP (attentive statistics pooling, wespeaker ECAPA-TDNN; wesep/models/bsrnn.py:217,352-356 via bsrnn.yaml:66-71) on
 * channels-last x, logits [R][T][C]: alpha = softmax_T(logits), out [R][2C] = sum alpha x || sqrt(max(sum alpha x^2 -
 * mean^2, floor)), aux [R][4][C] = (max logit, sum exp, mean, sum alpha x^2) saved for ws_astp_bwd.               */
int ws_astp_fwd(const float* x, const float* logits, int R, int T, int C, float floor_, float* out, float* aux,
                void* stream);
int ws_astp_bwd(const float* x, const float* logits, const float* out, const float* aux, const float* dout, int R, int T,
                int C, float floor_, float* dx, float* dlogits, void* stream);
/* y = act(x + rb[row / rows_per_r]) on [rows][C] (act 1 tanh, 3 sigmoid; rb NULL or [rows / rows_per_r][C]) and
 * dx = dy * act'(y) from the saved output: ECAPA's attention bottleneck (tanh) and SE gate (sigmoid).            */
int ws_rowbias_act_fwd(const float* x, const float* rb, long long rows, int C, int rows_per_r, int act, float* y,
                       void* stream);
int ws_act_bwd(const float* y, const float* dy, long long n, int act, float* dx, void* stream);

/* Segment pooling of the CAM++ context-aware mask (wespeaker `CAMPPlus`, the recipe's alternative speaker encoder:
 * wesep/models/bsrnn.py:217 via examples/librimix/tse/v2/confs/bsrnn.yaml:66-74; F.avg_pool1d(seg_len, ceil_mode) expanded
 * back over the frames) on channels-last [R][T][C], nseg = ceil(T / seg_len), last segment of an utterance shorter:
 *   ws_seg_sums : out[r][s][c] = sum_{t in segment s} a[r][t][c] (* b[r][t][c] when b)
 *   ws_seg_scale: out[r][t][c] = (x ? x[r][t][c] : 1) * m[r][t / seg_len][c]            (out may alias x)          */
int ws_seg_sums(const float* a, const float* b, int R, int T, int C, int seg_len, float* out, void* stream);
int ws_seg_scale(const float* x, const float* m, int R, int T, int C, int seg_len, float* out, void* stream);

/* ---- in-model enrollment front-end (bsrnn.py:231-242,343-350) -------------------
 * out[r][j] = y[reflect(j - pad)], y = pre-emphasis of x (speaker.py:10-23), row stride ldo >= T + 2*pad:
 * the centred, reflect-padded signal whose hop-strided row views are the STFT frames.                     */
int ws_preemph_pad(const float* x, int R, int T, int pad, int ldo, float coef, float* out, void* stream);
/* p[m][f] = re^2 + im^2 of interleaved spectra [M][lds] (nf bins); columns nf..ldp-1 zeroed               */
int ws_power_spec(const float* spec, long long M, int nf, int lds, int ldp, float* p, void* stream);
/* x = log(x + eps) in place                                                                               */
int ws_log_eps(float* x, long long n, float eps, void* stream);

/* ---- DPCCN pieces (wesep/modules/dpccn/convs.py, wesep/models/dpccn.py) -----------
 * channels-last [B][H][W][C], H = frame, W = frequency bin.  Conv2d / ConvTranspose2d with per-axis strides:
 * ws_im2col_hw + GEMM, and GEMM + ws_col2im_hw (the transposed convolution is the col2im gather of a GEMM output).*/
int ws_im2col_hw(const float* x, int R, int H, int W, int C, int k, int sh, int sw, int p, long long ldp,
                 float* patches, void* stream);
int ws_col2im_hw(const float* dpatches, int R, int H, int W, int C, int k, int sh, int sw, int p, float* dx,
                 void* stream);
/* nn.ELU: y = x > 0 ? x : expm1(x);  dx = dy * (x > 0 ? 1 : exp(x)) (dx may alias dy)                        */
int ws_elu_fwd(const float* x, long long n, float* y, void* stream);
int ws_elu_bwd(const float* x, const float* dy, long long n, float* dx, void* stream);
/* InstanceNorm{1,2}d without affine over the P positions of each of G batch rows ([G*P][C] channels-last):
 * sums [G][2][C] = ws_chan_sums(g = x, x = x) -> stats [G][2][C] = (mean, rstd); y = (x - mean) * rstd;
 * backward with sums = ws_chan_sums(g = dy, x = y): dx = rstd * (dy - S0/P - y * S1/P)                       */
int ws_inorm_finalize(const float* sums, int G, int C, long long P, float eps, float* stats, void* stream);
int ws_inorm_apply(const float* x, const float* stats, long long rows, int P, int C, float* y, void* stream);
int ws_inorm_bwd_apply(const float* y, const float* dy, const float* stats, const float* sums, long long rows, int P,
                       int C, float* dx, void* stream);
/* 3 x 3, stride-1, padding-1 convolution of a channels-last image through an LDS halo tile (conv3x3.hip): every input
 * pixel crosses the L2 -> CU path ~1.2 times instead of the 9 of the implicit patch matrix.  Replaces F.conv2d of the
 * dense blocks (wesep/modules/dpccn/convs.py:80-112) and, with the flipped / channel-swapped weights, its input gradient:
 *   Y[m][n] = bias[n] + R[m][n] + sum_{ky,kx,c} X[pixel(m) + (ky-1, kx-1)][c] * W[n][(ky*3 + kx)*Cin + c],  m = (b*H + h)*Wd + w
 * X: pixel stride ldx >= Cin; Y and R (bias / R may be NULL; R may alias Y): row stride ldy >= Cout.
 * W: the weights PACKED as bf16 hi / lo MFMA fragments, ceil(Cin / 16) * 9 * NTP * 2 units of 64 lanes x 8 bf16, NTP =
 * ceil(Cout / 32) rounded up to even when above 2 (wesep_amd.dev.conv3x3_pack writes the layout: unit
 * (((chunk*9 + tap)*NTP + t)*2 + part)*64 + lane = W[t*32 + (lane & 31)][tap][16 chunk + 8 (lane >> 5) + j], zero beyond
 * Cout / Cin; part 0 = bf16(w), part 1 = bf16(w - hi)); ldw is ignored.  Cin % 4 == 0, Cout % 4 == 0, Cout <= 1024.
 * Split-bf16 products, fp32 accumulation. */
typedef struct ws_conv3x3_args {
  const float* X;
  const float* W;
  const float* bias;
  const float* R;
  float* Y;
  long long ldx, ldw, ldy;
  int B, H, Wd, Cin, Cout, pad_;
} ws_conv3x3_args;
int ws_conv3x3(const ws_conv3x3_args* a, void* stream);
/* ABI v19: the packed weights of ws_conv3x3 in ONE launch from up to WS_C3_NSRC strided views of weight tensors -- the logical
 * W[n][tap][c] (n < Cout rows, c < Cin columns, zero-padded to the fragment grid) takes column c from the source k whose range
 * [col_off, col_off + cols) holds it:  W = w_k[n * s_row + (c - col_off) * s_col + (flip ? 8 - tap : tap) * s_tap].
 * A layer's forward: one source, nn.Conv2d weight [co][ci][3][3] -> (s_row, s_col, s_tap) = (9 Ci, 9, 1), flip 0.  The input
 * gradient of a channel block [lo, hi) of a dense block (convs.py:80-112): sources = the weights of the layers that read the
 * block, pointers advanced to input channel lo, rows = the block's channels (s_row = 9), columns = the layers' output channels
 * side by side (s_col = 9 Ci_k), flip 1.  out: ceil(Cin / 16) * 9 * NTP * 2 * 64 * 8 bf16 (= half as many floats).        */
#define WS_C3_NSRC 5
typedef struct ws_conv3x3_pack_src {
  const float* w;
  long long s_row, s_col, s_tap;
  int col_off, cols;
} ws_conv3x3_pack_src;
typedef struct ws_conv3x3_pack_args {
  ws_conv3x3_pack_src src[WS_C3_NSRC];
  float* out;
  int Cin, Cout, nsrc, flip;
} ws_conv3x3_pack_args;
int ws_conv3x3_pack(const ws_conv3x3_pack_args* a, void* stream);
/* Weight (and bias) gradient of a 3 x 3 convolution with padding 1, stride 1 along h and sw = 1 or 2 along w, one pass
 * over the image (conv3x3.hip; replaces ws_conv_wgrad / the implicit TN GEMM for these shapes: the dense blocks, the
 * (1, 2)-strided encoder convolutions and -- with image = dy, G = x -- the decoder's transposed convolutions of
 * wesep/modules/dpccn/convs.py, and the 3 x 3 convolutions of the ResNet speaker encoder):
 *   slab[split][n][(ky*3 + kx)*Cin + c] = sum over the split's pixels m = (b, h, w) of G[m][n] * X[b][h + ky - 1][sw*w + kx - 1][c]
 *   bslab[split][n]                    = sum over the split's pixels of G[m][n]                       (bslab may be NULL)
 * G [B*H*Wd rows, stride ldg >= Nn]; X [B][H][Wx] pixels of stride ldx >= Cin, Wd = (Wx - 1) / sw + 1.  The gradient grid
 * is cut into tiles of 30 rows x 4 columns (B * ceil(H / 30) * ceil(Wd / 4) of them, column-fastest); split s owns tiles
 * [s, s + 1) * tiles_per_split.  The caller sums the nsplit slabs (ws_reduce_slabs: deterministic, no atomics).
 * Cin % 4 == 0, Nn % 4 == 0. */
typedef struct ws_conv3x3_wgrad_args {
  const float* G;
  const float* X;
  float* slab;
  float* bslab;
  long long ldg, ldx, slab_stride, bslab_stride;
  int B, H, Wd, Wx, sw, Cin, Nn, nsplit, tiles_per_split, pad_;
} ws_conv3x3_wgrad_args;
int ws_conv3x3_wgrad(const ws_conv3x3_wgrad_args* a, void* stream);

/* InstanceNorm fused with its neighbouring ELU (convs.py:28-77: conv - ELU - IN; convs.py:115-152: IN - ELU - conv):
 *   flags bit 0: y = IN(ELU(x));  bit 1: y = ELU(IN(x));  statistics [G][2][C] as ws_inorm_finalize writes them.
 * ws_in_act_sums: slab[nsplit][G][2][C] partial sums over the P rows of each group -- forward (dy NULL): (sum u, sum u^2)
 * of u = pre(x), to be reduced and finalised by ws_reduce_slabs + ws_inorm_finalize; backward: (sum d, sum d * n) with
 * n = (u - mean) * rstd and d = dy * (bit 1 ? ELU'(n) : 1).  ws_in_act_apply: y from x and the statistics (one pass).
 * ws_in_act_bwd_apply: dx = pre'(x) * rstd * (d - S0/P - n * S1/P) from x, dy, the statistics and the reduced sums
 * (dx may alias dy when both strides agree).  Only the pre-activation x has to be kept for the backward.  x is dense
 * [rows][C]; y (apply) has row stride ldy, dy (sums, bwd_apply) row stride ldd and dx row stride lddx -- 0 = C, else >= C and % 4: the
 * dense blocks write y into / read dy from a column range of their one wide feature map (no torch.cat, no slices). */
int ws_in_act_sums(const float* x, const float* dy, long long ldd, const float* stats, int P, int G, int nsplit, int C,
                   int flags, float* slab, void* stream);
int ws_in_act_apply(const float* x, const float* stats, long long rows, int P, int C, int flags, float* y, long long ldy,
                    void* stream);
int ws_in_act_bwd_apply(const float* x, const float* dy, long long ldd, const float* stats, const float* sums, long long rows,
                        int P, int C, int flags, float* dx, long long lddx, void* stream);
/* nn.AvgPool2d(sz) and its adjoint; nn.Upsample(size = (H, W), mode = "bilinear") (align_corners False) and its
 * adjoint (a gather over destination pixels)                                                                  */
int ws_avgpool_fwd(const float* x, int B, int H, int W, int C, int sz, float* y, void* stream);
int ws_avgpool_bwd(const float* dy, int B, int H, int W, int C, int sz, float* dx, void* stream);
int ws_bilinear_fwd(const float* x, int B, int h, int w, int H, int W, int C, float* y, void* stream);
/* tmp: scratch of B * H * w * C floats (the adjoint runs as two separable passes: destination columns, then rows)   */
int ws_bilinear_bwd(const float* dy, int B, int h, int w, int H, int W, int C, float* tmp, float* dx, void* stream);
/* SpeakerFuseLayer multiply (mode 0) / additive (mode 1) on [B][T][F][C] with s [B][F] (speaker.py:102-121);
 * backward: dx, and ds [B][F] = sum over (t, c) of dy * x (mode 0) or dy (mode 1)                             */
int ws_scale_bf_fwd(const float* x, const float* s, int B, int T, int F, int C, int mode, float* y, void* stream);
int ws_scale_bf_bwd(const float* x, const float* dy, const float* s, int B, int T, int F, int C, int mode, float* dx,
                    float* ds, void* stream);
/* ABI v17: SpeakerFuseLayer 'concat' (speaker.py:95-101) on [B][T][F][C]: the Linear over the frequency axis of cat[x, e] --
 * y[b][t][f'][c] = sum_f W[f' * ldw + f] x[b][t][f][c] + rb[b][f'];  W: the first F columns of fc.linear.weight [F][F + E]
 * (ldw = F + E), rb [B][F] = We e + bias (ws_gemm_nt).  Exact fp32; F * C <= 16384.  Forward only: the native runtime's
 * form of the fusion (runtime/engine.cc); training composes it from GEMMs on a transposed view.                    */
int ws_freq_linear_fwd(const float* x, const float* W, long long ldw, const float* rb, int B, int T, int F, int C, float* y,
                       void* stream);

/* ---- TF-GridNet: everything but this row softmax is composed from the entry points
 * above (gridnet_block.py:212-213): y = softmax(scale * x) per row of n; dx = scale * y * (dy - sum(dy * y))   */
int ws_softmax_rows_fwd(const float* x, long long rows, int n, float scale, float* y, void* stream);
int ws_softmax_rows_bwd(const float* y, const float* dy, long long rows, int n, float scale, float* dx, void* stream);

/* nn.LayerNorm over short rows (gridnet_block.py:139-160 `intra_norm` / `inter_norm`, width = emb_dim; also any
 * [M][W] with W <= 256, W % 4 == 0): one pass forward (y and the (mean, rstd) pairs stats[M][2]), one pass backward:
 *   dx = rstd * (gamma*dy - mean_row(gamma*dy) - xhat * mean_row(gamma*dy*xhat)) (+ res; dx may alias dy)
 *   slab[ws_rowln_grid(M, W)][2][W]: per-workgroup partial sums of d(beta) (row 0) and d(gamma) (row 1); the caller
 *   reduces them (ws_reduce_slabs) -- the grid is a function of (M, W) only, so the sums are reproducible.      */
int ws_rowln_grid(long long M, int W);
int ws_rowln_fwd(const float* x, const float* gamma, const float* beta, long long M, int W, float eps, float* y,
                 float* stats, void* stream);
int ws_rowln_bwd(const float* x, const float* dy, const float* stats, const float* gamma, const float* res, long long M,
                 int W, float* dx, float* slab, void* stream);

/* The attention heads of a TF-GridNet block in one pass (heads.hip; gridnet_block.py:176-199: per head Conv2d(1x1) ->
 * PReLU -> LayerNormalization4DCF over (E, F), heads concatenated along the batch, `.transpose(1, 2).flatten(2)`):
 *   y[(h*B + b)*Tp + t][q*ch + e] = gamma[h][q*ch + e] * (u - mean) * rstd + beta[h][q*ch + e],
 *   u = PReLU(x[(b*T + t)*Q + q][h*ch + e]; slope[h]),  mean / biased variance over the Q*ch elements of (b, t, h)
 * x: the projection's columns (pointer at the first of them) in rows of stride ldx -- Q, K and V are one GEMM with
 * concatenated weights; y rows t in [T, Tp) are written as zeros (keys / values padded to 16-byte rows of the logits);
 * stats[h][b*T + t][2] = (mean, rstd).  ws_heads_bwd: dx (layout of x, row stride lddx) from x, dy (layout of y), the
 * statistics; slab[nwg][2*W + 8], W = Q*nh*ch: per-workgroup partial sums of d(gamma) (elements [0, W) in the order
 * (q, h, e)), d(beta) ([W, 2W)) and d(slope) ([2W, 2W + nh)) for ws_reduce_slabs; nwg = the launch grid (any > 0: the
 * sums are reproducible for a given nwg).  ch % 4 == 0, nh <= 8, W <= 9216. */
typedef struct ws_heads_args {
  const float* x;
  const float* dy;
  const float* slope;
  const float* gamma;
  const float* beta;
  float* y;
  float* stats;
  float* dx;
  float* slab;
  long long ldx, lddx;
  int B, T, Tp, Q, nh, ch, nwg;
  float eps;
} ws_heads_args;
int ws_heads_fwd(const ws_heads_args* a, void* stream);
int ws_heads_bwd(const ws_heads_args* a, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* WESEP_HIP_H_ */
