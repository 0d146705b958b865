// libwesep_engine.so -- native inference runtime of wesep_amd (include/wesep_engine.h).
//
// MI355X counterpart of the reference's C++ runtime (runtime/separate/separate_engine.{h,cc}: a TorchScript module on
// LibTorch-CPU plus a host kaldi fbank).  Here the pBSRNN forward (wesep/models/bsrnn.py:300-394) is a fixed launch
// plan over the device library's C ABI (include/wesep_hip.h):
//   * load: the weight container is read, uploaded once, and everything that the Python training path re-derives
//     per step is derived once -- [W_ih_f | W_ih_r] concatenation, MFMA-fragment packs of W_ih / W_hh / proj for the
//     blocked-layout GEMMs and recurrences, BatchNorm folded to (running mean, rstd), conv kernels permuted to the
//     im2col column order, the folded kaldi fbank basis (see wesep_amd/utils/funcs.py);
//   * forward: activations come from one grow-only device arena with stack discipline (per-layer scratch is released
//     when the layer ends, so the peak is one ResRNN's working set, not the sum); the per-band grouped GEMMs get their
//     descriptor tables rebuilt only when the frame count changes.
// Host code only: no kernels in this file.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../include/wesep_engine.h"
#include "../include/wesep_hip.h"

namespace {

thread_local char g_err[768] = "";

void set_err(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

constexpr int kN = 128;                 // feature_dim
constexpr int kH = 256;                 // LSTM hidden size
constexpr int kG4 = 4 * kH;             // gate rows per direction
constexpr int kNBin = 257;              // n_fft / 2 + 1
constexpr int kHop = 128;
constexpr int kBig = 1 << 30;           // row divisor meaning "never wraps"
constexpr float kGnEps = 1.1920928955078125e-07f;   // torch.finfo(float32).eps, bsrnn.py:23
constexpr float kBnEps = 1e-5f;
constexpr float kTstpEps = 1e-7f;
constexpr float kAstpFloor = 1e-7f;

struct Tensor {
  std::vector<int64_t> dims;
  size_t off = 0;   // floats into the weight blob
  size_t n = 0;
};

// ---- device memory: chunked bump allocator with stack discipline --------------------------------------------
struct Arena {
  struct Chunk {
    char* base;
    size_t cap;
  };
  std::vector<Chunk> chunks;
  size_t cur = 0, top = 0;      // current chunk and offset inside it
  size_t live_bytes = 0, peak_bytes = 0;
  bool dry = false;
  bool poison = getenv("WS_ENGINE_POISON") != nullptr;

  struct Mark {
    size_t cur, top, live;
  };

  static size_t round_up(size_t b) { return (b + 255) & ~size_t(255); }

  bool add_chunk(size_t bytes) {
    Chunk c{nullptr, bytes};
    if (dry) {
      c.base = static_cast<char*>(malloc(bytes));
    } else if (hipMalloc(reinterpret_cast<void**>(&c.base), bytes) != hipSuccess) {
      c.base = nullptr;
    }
    if (!c.base) return false;
    chunks.push_back(c);
    return true;
  }

  float* alloc(size_t nfloats) {
    const size_t bytes = round_up(nfloats * 4 + 4);
    while (true) {
      if (cur < chunks.size() && top + bytes <= chunks[cur].cap) break;
      if (cur + 1 < chunks.size()) {           // move on to the next existing chunk
        ++cur;
        top = 0;
        continue;
      }
      const size_t want = bytes > (size_t(256) << 20) ? bytes : (size_t(256) << 20);
      if (!add_chunk(want)) {
        set_err("engine: device allocation of %zu bytes failed", want);
        return nullptr;
      }
      cur = chunks.size() - 1;
      top = 0;
    }
    float* p = reinterpret_cast<float*>(chunks[cur].base + top);
    // WS_ENGINE_POISON=1 (tests): every allocation starts as NaN (0xFF bytes), so a launch plan that reads memory no
    // kernel has written shows up as NaN output instead of depending on what the arena held before (device-wide
    // syncs around it: the engine's stream is non-blocking).
    if (poison && !dry) {
      (void)hipDeviceSynchronize();
      (void)hipMemset(p, 0xFF, bytes);
      (void)hipDeviceSynchronize();
    }
    top += bytes;
    live_bytes += bytes;
    if (live_bytes > peak_bytes) peak_bytes = live_bytes;
    return p;
  }

  Mark mark() const { return Mark{cur, top, live_bytes}; }
  void release(const Mark& m) {
    cur = m.cur;
    top = m.top;
    live_bytes = m.live;
  }
  void reset() {
    cur = 0;
    top = 0;
    live_bytes = 0;
  }
  // after a forward that had to add chunks: one chunk of the peak size for the next call
  void consolidate() {
    if (chunks.size() <= 1 || live_bytes != 0) return;
    const size_t want = round_up(peak_bytes + (size_t(16) << 20));
    free_all();
    add_chunk(want);
  }
  void free_all() {
    for (auto& c : chunks) {
      if (dry)
        free(c.base);
      else
        (void)hipFree(c.base);
    }
    chunks.clear();
    reset();
  }
};

struct RnnPrep {            // one ResRNN (bsrnn.py:26-46), everything the forward needs, device pointers
  const float *norm_w, *norm_b, *whf, *whr, *proj_b;
  float *bcat, *wih_pack, *proj_pack, *fpack, *pack16, *pack32;
};

struct ConvPrep {           // conv (bias-free) + BatchNorm(eval) (+ ReLU) of the speaker encoder
  int cin, cout, k, stride, ldp;
  int sw = 0;               // stride along W when it differs from `stride` (CAM++'s FCM head strides the mel axis only); 0: same
  bool relu;
  const float *gamma, *beta;
  float *w2, *st;           // [cout][ldp] in im2col column order; [2][cout] = (running mean, rstd)
};

struct BlockPrep {          // BasicBlock: c1 (3x3, stride) c2 (3x3); Bottleneck: c1 (1x1) c2 (3x3, stride) c3 (1x1, x4)
  ConvPrep c1, c2, c3, sc;
  bool has_sc;
};

struct TdnnPrep {           // Conv1d (bias) -> ReLU -> BatchNorm1d(eval): wespeaker ECAPA-TDNN's Conv1dReluBn
  int cin, cout, k, dil;
  const float *w, *bias, *gamma, *beta;   // w: k == 1 the checkpoint's [cout][cin]; else the one-row-image view weight
  float* st;                              // [2][cout] = (running mean, rstd)
};

struct SeRes2Prep {         // SE_Res2Block: 1x1 TDNN, Res2Net branches, 1x1 TDNN, squeeze-excitation, + input
  TdnnPrep in, out;
  std::vector<TdnnPrep> branch;
  std::string se;           // "...se_res2block.3." (linear1 / linear2)
};

// ---- wespeaker CAM++ (spk_kind 2; wesep_amd/models/campplus.py) ----
struct CamBn {              // BatchNorm1d(eval) of a pre-activation D-TDNN layer: (mean, rstd) + affine operands (or ones / zeros)
  int c;
  float* st;
  const float *gamma, *beta;
};
struct CamLayer {           // CAMDenseTDNNLayer: BN-ReLU, 1x1 to 128, BN-ReLU, dilated k = 3 conv to 32, context-aware mask
  int cin, dil;
  CamBn bn1, bn2;
  const float *w1, *wloc;   // linear1 [128][cin]; linear_local as the 3 x 3 view of the one-row image [32][9 * 128]
  const float *l1w, *l1b, *l2w, *l2b;
};
struct CamTransit {         // BN-ReLU + 1x1 (bias-free) to half the channels
  int cin, cout;
  CamBn bn;
  const float* w;
};

}  // namespace

static std::mutex g_device_mutex[16];   // see ws_engine_separate

namespace {
struct GridNet;                          // TF-GridNet plan state (arch 3), defined with the plan
void grid_free(GridNet* g);
}

struct ws_engine {
  GridNet* grid = nullptr;
  bool dry = false;
  int device = 0, cu_count = 0;
  hipStream_t stream = nullptr;
  std::map<std::string, int64_t> meta;
  std::map<std::string, Tensor> tensors;
  std::vector<float> hw;          // host copy of the weight blob
  float* dw = nullptr;            // device copy
  Arena persist, work;
  long long n_launches = 0;
  long long cluster_fallbacks = 0;   // forwards in which a cluster recurrence timed out and the streaming kernels took over
  unsigned* cl_status = nullptr;     // sticky device word set by ws_lstm_fwd_cluster on a timeout
  // configuration
  int sr = 16000, num_repeat = 6, E = 256, fuse = 2, multi_fuse = 0, use_xform = 0, joint = 0, feat_dim = 80;
  int blocks[4] = {0, 0, 0, 0};
  // band tables (bsrnn.py:190-209)
  std::vector<int> bw, f0;
  int K = 0;
  int *d_band_of_bin = nullptr, *d_f0 = nullptr, *d_bw = nullptr, *d_bw2 = nullptr, *d_off2 = nullptr;
  // prepared weights
  std::vector<RnnPrep> rnn;       // 2 per BSNet: band_rnn (time view), band_comm (band view)
  std::vector<int> sep_kind;      // per entry of separator.separation: 0 fuse layer, 1 BSNet
  ConvPrep stem;
  std::vector<BlockPrep> res_blocks;
  // ECAPA-TDNN speaker encoder (spk_kind 1; wesep_amd/models/ecapa_tdnn.py)
  int spk_kind = 0, spk_channels = 512, spk_glob = 0, spk_emb_bn = 0;
  int spk_bottleneck = 0, spk_two_emb = 0;        // wespeaker ResNet50 / 101 / 152 blocks; seg_1 -> ReLU -> BN -> seg_2
  float* seg_bn_st = nullptr;
  TdnnPrep tdnn1;
  std::vector<SeRes2Prep> se_blocks;
  // CAM++ speaker encoder (spk_kind 2; wesep_amd/models/campplus.py)
  std::vector<ConvPrep> cam_fcm;      // conv1, then per BasicResBlock (conv1, [shortcut], conv2), then conv2
  std::vector<int> cam_fcm_kind;      // 0 plain, 1 block conv1, 2 shortcut, 3 block conv2 (+ residual)
  ConvPrep cam_dummy;
  const float* cam_tdnn_w = nullptr;  // xvector.tdnn as the 5 x 5 view weight
  CamBn cam_tdnn_bn, cam_out_bn, cam_dense_bn;
  std::vector<std::vector<CamLayer>> cam_blocks;
  std::vector<CamTransit> cam_transit;
  int cam_init = 128, cam_growth = 32, cam_bn = 128;
  float *cam_one = nullptr, *cam_zero = nullptr, *cam_id_st = nullptr;   // ones / zeros / (0 x C | 1 x C), C = 1024
  float *id_st = nullptr, *id_one = nullptr, *id_zero = nullptr;   // identity BatchNorm operands: y = x + res
  float *pool_bn_st = nullptr, *emb_bn_st = nullptr;
  float *slope0 = nullptr, *slope1 = nullptr;     // PReLU slopes 0 (ReLU) and 1 (identity)
  float *fb_basis = nullptr, *fb_bank = nullptr, *fb_floor = nullptr;
  int fb_win = 400, fb_shift = 160, fb_padded = 512;
  // in-model front-end of spk_feat = False models (bsrnn.py:231-242,343-350): PreEmphasis + MelSpectrogram
  int spk_feat = 1;
  float *mel_basis = nullptr, *mel_fbt = nullptr, mel_coef = 0.97f;
  int mel_lds = 516, mel_ldp = 260;
  // Conv-TasNet / SpEx+ (arch 1; wesep/models/convtasnet.py): geometry and prepared operands
  int arch = 0;                  // 0 pBSRNN, 1 Conv-TasNet (Multi encoder / decoder, gLN, concatConv fusion)
  int tN = 512, tL = 16, tB = 128, tH = 512, tP = 3, tX = 8, tR = 3;
  float* tas_dec_wt = nullptr;   // decoder_1d_1 weight transposed to [L][N]
  float* tas_bn_st[3][2] = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};  // eval BN (mean, rstd) [2][C]
  // DPCCN (arch 2; wesep/models/dpccn.py): prepared operands, in the order the forward consumes them
  int dp_fuse = 2, dp_causal = 0, dp_tcn_blocks = 10, dp_tcn_layers = 2;
  float *dp_ana4 = nullptr, *dp_syn4 = nullptr;       // analysis basis [4 * 257][512] (re, im, 0, 0 per bin), synthesis [512][4 * 257]
  float *dp_w_in = nullptr;                           // conv2d (2 -> 16) as [16][9 * 4] on the (re, im, 0, 0) pixels
  float *dp_w_out = nullptr, *dp_b_out = nullptr;     // deconv2d (32 -> 2) as [4][9 * 32] + bias[4]: (re, im, 0, 0) per bin
  float *dp_ident = nullptr, *dp_ones = nullptr, *dp_zeros = nullptr;   // (mean 0, rstd 1) rows, gamma 1, beta 0 of the plain depthwise conv
  std::map<std::string, float*> dp_w;                 // per-layer GEMM weights / conv3x3 packs, keyed by the layer's state_dict prefix
  // grouped-GEMM descriptor tables, rebuilt when (R, Tf) changes
  int desc_R = -1, desc_Tf = -1;
  ws_group_nt *d_bn = nullptr, *d_l1 = nullptr, *d_l2 = nullptr, *d_l3 = nullptr;

  const Tensor* find(const std::string& name) const {
    auto it = tensors.find(name);
    return it == tensors.end() ? nullptr : &it->second;
  }
  const float* dev(const std::string& name) const {
    const Tensor* t = find(name);
    return t ? dw + t->off : nullptr;
  }
  const float* host(const std::string& name) const {
    const Tensor* t = find(name);
    return t ? hw.data() + t->off : nullptr;
  }
};

namespace {

// A launch "passes" when it succeeded, or -- in a dry run -- when it failed for any reason other than its
// argument validation (there is no device to launch on).
bool passes(ws_engine* e, int rc, const char* what) {
  ++e->n_launches;
  if (rc == WS_OK) return true;
  if (e->dry && rc != WS_ERR_INVALID) return true;
  set_err("engine: %s failed (rc=%d): %s", what, rc, ws_last_error());
  return false;
}

#define WS_RUN(e, call)                              \
  do {                                               \
    const int rc__ = (call);                         \
    if (!passes((e), rc__, #call)) return rc__ ? rc__ : WS_ERR_LAUNCH; \
  } while (0)

#define WS_PTR(p)                  \
  do {                             \
    if (!(p)) return WS_ERR_LAUNCH; \
  } while (0)

int to_device(ws_engine* e, void* dst, const void* src, size_t bytes) {
  if (e->dry) {
    memcpy(dst, src, bytes);
    return WS_OK;
  }
  if (hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, e->stream) != hipSuccess ||
      hipStreamSynchronize(e->stream) != hipSuccess) {
    set_err("engine: host-to-device copy of %zu bytes failed", bytes);
    return WS_ERR_LAUNCH;
  }
  return WS_OK;
}

int to_host(ws_engine* e, void* dst, const void* src, size_t bytes) {
  if (e->dry) return WS_OK;    // nothing was computed: leave the caller's buffer untouched
  if (hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, e->stream) != hipSuccess ||
      hipStreamSynchronize(e->stream) != hipSuccess) {
    set_err("engine: device-to-host copy of %zu bytes failed", bytes);
    return WS_ERR_LAUNCH;
  }
  return WS_OK;
}

int zero_device(ws_engine* e, void* p, size_t bytes) {
  if (e->dry) {
    memset(p, 0, bytes);
    return WS_OK;
  }
  if (hipMemsetAsync(p, 0, bytes, e->stream) != hipSuccess) {
    set_err("engine: memset failed");
    return WS_ERR_LAUNCH;
  }
  return WS_OK;
}

float* upload(ws_engine* e, Arena& a, const float* src, size_t n) {
  float* d = a.alloc(n);
  if (!d) return nullptr;
  if (to_device(e, d, src, n * 4) != WS_OK) return nullptr;
  return d;
}

int* upload_ints(ws_engine* e, Arena& a, const std::vector<int>& v) {
  return reinterpret_cast<int*>(upload(e, a, reinterpret_cast<const float*>(v.data()), v.size()));
}

// ---- weight container (written by wesep_amd/bin/export_engine.py) ---------------------------------------------
//   char magic[8] = "WSEPW001"
//   u32 n_meta;    n_meta    x { char key[32]; i64 value }
//   u32 n_tensors; n_tensors x { u32 name_len; char name[name_len]; u32 ndim; i64 dims[ndim]; u64 offset_floats }
//   u64 n_floats;  float data[n_floats]
bool read_exact(FILE* f, void* p, size_t n) { return fread(p, 1, n, f) == n; }

int load_container(ws_engine* e, const char* path) {
  FILE* f = fopen(path, "rb");
  if (!f) {
    set_err("engine: cannot open %s", path);
    return WS_ERR_INVALID;
  }
  char magic[8];
  uint32_t n_meta = 0, n_tensors = 0;
  bool ok = read_exact(f, magic, 8) && memcmp(magic, "WSEPW001", 8) == 0 && read_exact(f, &n_meta, 4) &&
            n_meta < 4096;
  for (uint32_t i = 0; ok && i < n_meta; ++i) {
    char key[33] = {0};
    int64_t v = 0;
    ok = read_exact(f, key, 32) && read_exact(f, &v, 8);
    if (ok) e->meta[key] = v;
  }
  ok = ok && read_exact(f, &n_tensors, 4) && n_tensors < (1u << 20);
  for (uint32_t i = 0; ok && i < n_tensors; ++i) {
    uint32_t name_len = 0, ndim = 0;
    ok = read_exact(f, &name_len, 4) && name_len < 1024;
    std::string name(name_len, '\0');
    ok = ok && read_exact(f, &name[0], name_len) && read_exact(f, &ndim, 4) && ndim <= 8;
    Tensor t;
    t.n = 1;
    for (uint32_t d = 0; ok && d < ndim; ++d) {
      int64_t v = 0;
      ok = read_exact(f, &v, 8) && v >= 0;
      t.dims.push_back(v);
      // overflow-checked product: a crafted container must not wrap n (and pass the bounds check below)
      if (ok && v != 0 && t.n > (uint64_t(1) << 40) / static_cast<uint64_t>(v)) ok = false;
      t.n *= static_cast<size_t>(v);
    }
    uint64_t off = 0;
    ok = ok && read_exact(f, &off, 8);
    t.off = off;
    if (ok) e->tensors[name] = t;
  }
  uint64_t n_floats = 0;
  ok = ok && read_exact(f, &n_floats, 8) && n_floats < (uint64_t(1) << 34);
  if (ok) {
    e->hw.resize(n_floats);
    ok = read_exact(f, e->hw.data(), n_floats * 4);
  }
  fclose(f);
  if (!ok) {
    set_err("engine: %s is not a valid wesep_amd weight container", path);
    return WS_ERR_INVALID;
  }
  for (auto& kv : e->tensors) {
    if (kv.second.n > e->hw.size() || kv.second.off > e->hw.size() - kv.second.n) {   // no wrap-around
      set_err("engine: tensor %s exceeds the data section", kv.first.c_str());
      return WS_ERR_INVALID;
    }
  }
  return WS_OK;
}

int64_t meta_or(const ws_engine* e, const char* key, int64_t dflt) {
  auto it = e->meta.find(key);
  return it == e->meta.end() ? dflt : it->second;
}

bool require(ws_engine* e, const std::string& name, std::initializer_list<int64_t> dims) {
  const Tensor* t = e->find(name);
  if (!t) {
    set_err("engine: tensor %s is missing from the container", name.c_str());
    return false;
  }
  std::vector<int64_t> want(dims);
  size_t n = 1;
  for (auto d : want) n *= static_cast<size_t>(d);
  if (t->n != n) {
    set_err("engine: tensor %s has %zu elements, expected %zu", name.c_str(), t->n, n);
    return false;
  }
  return true;
}

// ---- load-time preparation -----------------------------------------------------------------------------------
void band_table(ws_engine* e) {         // bsrnn.py:190-209
  const double nyq = e->sr / 2.0;
  auto bwid = [&](double hz) { return static_cast<int>(floor(hz / nyq * kNBin)); };
  e->bw.clear();
  for (int i = 0; i < 15; ++i) e->bw.push_back(bwid(100));
  for (int i = 0; i < 10; ++i) e->bw.push_back(bwid(200));
  for (int i = 0; i < 5; ++i) e->bw.push_back(bwid(500));
  e->bw.push_back(bwid(2000));
  int sum = 0;
  for (int b : e->bw) sum += b;
  e->bw.push_back(kNBin - sum);
  e->K = static_cast<int>(e->bw.size());
  e->f0.assign(e->K, 0);
  for (int g = 1; g < e->K; ++g) e->f0[g] = e->f0[g - 1] + e->bw[g - 1];
}

int prep_rnn(ws_engine* e, const std::string& pre, RnnPrep* r) {
  static const char* names[] = {"rnn.weight_ih_l0", "rnn.weight_hh_l0", "rnn.bias_ih_l0", "rnn.bias_hh_l0",
                                "rnn.weight_ih_l0_reverse", "rnn.weight_hh_l0_reverse", "rnn.bias_ih_l0_reverse",
                                "rnn.bias_hh_l0_reverse"};
  const int64_t shapes[][2] = {{kG4, kN}, {kG4, kH}, {kG4, 1}, {kG4, 1}, {kG4, kN}, {kG4, kH}, {kG4, 1}, {kG4, 1}};
  for (int i = 0; i < 8; ++i)
    if (!require(e, pre + names[i], {shapes[i][0], shapes[i][1]})) return WS_ERR_INVALID;
  if (!require(e, pre + "norm.weight", {kN}) || !require(e, pre + "norm.bias", {kN}) ||
      !require(e, pre + "proj.weight", {kN, 2 * kH}) || !require(e, pre + "proj.bias", {kN}))
    return WS_ERR_INVALID;
  const float* wih_f = e->dev(pre + names[0]);
  const float* wih_r = e->dev(pre + names[4]);
  r->whf = e->dev(pre + names[1]);
  r->whr = e->dev(pre + names[5]);
  r->norm_w = e->dev(pre + "norm.weight");
  r->norm_b = e->dev(pre + "norm.bias");
  r->proj_b = e->dev(pre + "proj.bias");
  Arena& a = e->persist;
  float* wcat = a.alloc(size_t(2) * kG4 * kN);
  r->bcat = a.alloc(2 * kG4);
  r->wih_pack = a.alloc(size_t(2) * kG4 * kN);
  r->proj_pack = a.alloc(size_t(kN) * 2 * kH);
  r->fpack = a.alloc(WS_LSTM_FUSED_PACK_FLOATS);
  r->pack16 = a.alloc(WS_LSTM_PACK_FLOATS);
  r->pack32 = a.alloc(WS_LSTM_PACK_FLOATS);
  float* bwd_scratch = a.alloc(WS_LSTM_PACK_FLOATS);   // the backward-pass pack is produced too; unused here
  WS_PTR(wcat && r->bcat && r->wih_pack && r->proj_pack && r->fpack && r->pack16 && r->pack32 && bwd_scratch);
  void* s = e->stream;
  WS_RUN(e, ws_lstm_cat_ih(wih_f, wih_r, e->dev(pre + names[2]), e->dev(pre + names[3]), e->dev(pre + names[6]),
                           e->dev(pre + names[7]), kN, wcat, r->bcat, s));
  WS_RUN(e, ws_pack_w(wcat, 2 * kG4, kN, kN, 0, 0, r->wih_pack, s));
  WS_RUN(e, ws_pack_w(e->dev(pre + "proj.weight"), kN, 2 * kH, 2 * kH, 0, 1, r->proj_pack, s));
  WS_RUN(e, ws_lstm_pack_fused(wih_f, wih_r, r->whf, r->whr, r->fpack, s));
  WS_RUN(e, ws_lstm_pack(r->whf, r->whr, r->pack16, bwd_scratch, WS_LSTM_BF16X3_BLK16, s));
  WS_RUN(e, ws_lstm_pack(r->whf, r->whr, r->pack32, bwd_scratch, WS_LSTM_BF16X3_BLK, s));
  return WS_OK;
}

int prep_conv(ws_engine* e, const std::string& conv, const std::string& bn, int cin, int cout, int k, int stride,
              bool relu, ConvPrep* c) {
  if (!require(e, conv + ".weight", {cout, cin, k, k}) || !require(e, bn + ".weight", {cout}) ||
      !require(e, bn + ".bias", {cout}) || !require(e, bn + ".running_mean", {cout}) ||
      !require(e, bn + ".running_var", {cout}))
    return WS_ERR_INVALID;
  c->cin = cin;
  c->cout = cout;
  c->k = k;
  c->stride = stride;
  c->relu = relu;
  const int kk = k * k * cin;
  c->ldp = (kk + 3) / 4 * 4;
  // [cout][cin][ky][kx] -> [cout][(ky*k + kx)*cin + c], zero-padded to ldp columns (functional_resnet.py:29-31)
  const float* w = e->host(conv + ".weight");
  std::vector<float> w2(size_t(cout) * c->ldp, 0.f);
  for (int o = 0; o < cout; ++o)
    for (int ci = 0; ci < cin; ++ci)
      for (int t = 0; t < k * k; ++t) w2[size_t(o) * c->ldp + size_t(t) * cin + ci] = w[(size_t(o) * cin + ci) * k * k + t];
  const float* rm = e->host(bn + ".running_mean");
  const float* rv = e->host(bn + ".running_var");
  std::vector<float> st(2 * size_t(cout));
  for (int o = 0; o < cout; ++o) {
    st[o] = rm[o];
    st[cout + o] = 1.0f / sqrtf(rv[o] + kBnEps);
  }
  c->w2 = upload(e, e->persist, w2.data(), w2.size());
  c->st = upload(e, e->persist, st.data(), st.size());
  c->gamma = e->dev(bn + ".weight");
  c->beta = e->dev(bn + ".bias");
  WS_PTR(c->w2 && c->st);
  return WS_OK;
}

float* bn_eval_stats(ws_engine* e, const std::string& bn, int c);
int prep_campplus(ws_engine* e);
int campplus_embed(ws_engine* e, const float* fbank, int R, int Te, float* emb);

int prep_resnet(ws_engine* e) {
  const int m = 32, ex = e->spk_bottleneck ? 4 : 1;
  const std::string p = "spk_model.";
  int rc = prep_conv(e, p + "conv1", p + "bn1", 1, m, 3, 1, true, &e->stem);
  if (rc != WS_OK) return rc;
  int inp = m;
  for (int li = 0; li < 4; ++li) {
    const int planes = m << li, first_stride = li == 0 ? 1 : 2;
    for (int bi = 0; bi < e->blocks[li]; ++bi) {
      const std::string q = p + "layer" + std::to_string(li + 1) + "." + std::to_string(bi) + ".";
      const int stride = bi == 0 ? first_stride : 1;
      BlockPrep b;
      b.has_sc = stride != 1 || inp != ex * planes;
      if (e->spk_bottleneck) {
        if ((rc = prep_conv(e, q + "conv1", q + "bn1", inp, planes, 1, 1, true, &b.c1)) != WS_OK) return rc;
        if ((rc = prep_conv(e, q + "conv2", q + "bn2", planes, planes, 3, stride, true, &b.c2)) != WS_OK) return rc;
        if ((rc = prep_conv(e, q + "conv3", q + "bn3", planes, ex * planes, 1, 1, true, &b.c3)) != WS_OK) return rc;
      } else {
        if ((rc = prep_conv(e, q + "conv1", q + "bn1", inp, planes, 3, stride, true, &b.c1)) != WS_OK) return rc;
        if ((rc = prep_conv(e, q + "conv2", q + "bn2", planes, planes, 3, 1, true, &b.c2)) != WS_OK) return rc;
      }
      if (b.has_sc &&
          (rc = prep_conv(e, q + "shortcut.0", q + "shortcut.1", inp, ex * planes, 1, stride, false, &b.sc)) != WS_OK)
        return rc;
      e->res_blocks.push_back(b);
      inp = ex * planes;
    }
  }
  const int stats_dim = (e->feat_dim / 8) * m * 8 * ex;
  if (!require(e, p + "seg_1.weight", {e->E, 2 * stats_dim}) || !require(e, p + "seg_1.bias", {e->E})) return WS_ERR_INVALID;
  const float s0 = 0.f, s1 = 1.f;
  e->slope0 = upload(e, e->persist, &s0, 1);
  e->slope1 = upload(e, e->persist, &s1, 1);
  WS_PTR(e->slope0 && e->slope1);
  if (e->spk_two_emb) {        // embed_b = seg_2(BatchNorm1d(affine = False)(relu(seg_1(stats)))): the separator takes it
    if (!require(e, p + "seg_bn_1.running_mean", {e->E}) || !require(e, p + "seg_bn_1.running_var", {e->E}) ||
        !require(e, p + "seg_2.weight", {e->E, e->E}) || !require(e, p + "seg_2.bias", {e->E}))
      return WS_ERR_INVALID;
    e->seg_bn_st = bn_eval_stats(e, p + "seg_bn_1", e->E);
    std::vector<float> one(e->E, 1.f), zero(e->E, 0.f);
    e->id_one = upload(e, e->persist, one.data(), one.size());
    e->id_zero = upload(e, e->persist, zero.data(), zero.size());
    WS_PTR(e->seg_bn_st && e->id_one && e->id_zero);
  }
  return WS_OK;
}

float* bn_eval_stats(ws_engine* e, const std::string& bn, int c) {
  const float* rm = e->host(bn + ".running_mean");
  const float* rv = e->host(bn + ".running_var");
  std::vector<float> st(2 * size_t(c));
  for (int o = 0; o < c; ++o) {
    st[o] = rm[o];
    st[c + o] = 1.0f / sqrtf(rv[o] + kBnEps);
  }
  return upload(e, e->persist, st.data(), st.size());
}

// Conv1d [cout][cin][k] (dilation dil, 'same' padding) + BatchNorm1d of one Conv1dReluBn.  k > 1: the convolution runs
// as the k x k implicit-patch view of the one-row image [R][1][T][cin] (include/wesep_hip.h, ws_conv_view), whose
// weight is zero outside the middle kernel row (functional_ecapa.py:34-38).
int prep_tdnn(ws_engine* e, const std::string& conv, const std::string& bn, int cin, int cout, int k, int dil, TdnnPrep* t) {
  if (!require(e, conv + ".weight", {cout, cin, k}) || !require(e, conv + ".bias", {cout}) ||
      !require(e, bn + ".weight", {cout}) || !require(e, bn + ".bias", {cout}) ||
      !require(e, bn + ".running_mean", {cout}) || !require(e, bn + ".running_var", {cout}))
    return WS_ERR_INVALID;
  if (cin % 4 || cout % 4) {
    set_err("engine: ECAPA-TDNN channel counts must be multiples of 4 (%s: %d -> %d)", conv.c_str(), cin, cout);
    return WS_ERR_INVALID;
  }
  t->cin = cin, t->cout = cout, t->k = k, t->dil = dil;
  t->bias = e->dev(conv + ".bias");
  t->gamma = e->dev(bn + ".weight");
  t->beta = e->dev(bn + ".bias");
  if (k == 1) {
    t->w = e->dev(conv + ".weight");
  } else {
    const float* w = e->host(conv + ".weight");
    std::vector<float> w2(size_t(cout) * k * k * cin, 0.f);
    for (int o = 0; o < cout; ++o)
      for (int ci = 0; ci < cin; ++ci)
        for (int kx = 0; kx < k; ++kx)
          w2[(size_t(o) * k * k + size_t(k / 2) * k + kx) * cin + ci] = w[(size_t(o) * cin + ci) * k + kx];
    t->w = upload(e, e->persist, w2.data(), w2.size());
  }
  t->st = bn_eval_stats(e, bn, cout);
  WS_PTR(t->w && t->st);
  return WS_OK;
}

// wespeaker ECAPA_TDNN(_GLOB)_c512 / _c1024 (models/ecapa_tdnn.py): shapes checked against the container, conv-view
// weights and BatchNorm(eval) statistics prepared once
int prep_ecapa(ws_engine* e) {
  const std::string p = "spk_model.";
  const int C = e->spk_channels, F = e->feat_dim, scale = 8, width = C / scale, P = 1536, B = 128;
  if (C % (4 * scale)) {
    set_err("engine: ECAPA-TDNN channels %d: a multiple of 32 is required", C);
    return WS_ERR_INVALID;
  }
  int rc = prep_tdnn(e, p + "layer1.conv", p + "layer1.bn", F, C, 5, 1, &e->tdnn1);
  if (rc != WS_OK) return rc;
  for (int li = 0; li < 3; ++li) {
    const std::string q = p + "layer" + std::to_string(li + 2) + ".se_res2block.";
    SeRes2Prep b;
    if ((rc = prep_tdnn(e, q + "0.conv", q + "0.bn", C, C, 1, 1, &b.in)) != WS_OK) return rc;
    for (int i = 0; i < scale - 1; ++i) {
      TdnnPrep t;
      if ((rc = prep_tdnn(e, q + "1.convs." + std::to_string(i), q + "1.bns." + std::to_string(i), width, width, 3, li + 2,
                          &t)) != WS_OK)
        return rc;
      b.branch.push_back(t);
    }
    if ((rc = prep_tdnn(e, q + "2.conv", q + "2.bn", C, C, 1, 1, &b.out)) != WS_OK) return rc;
    b.se = q + "3.";
    if (!require(e, b.se + "linear1.weight", {B, C}) || !require(e, b.se + "linear1.bias", {B}) ||
        !require(e, b.se + "linear2.weight", {C, B}) || !require(e, b.se + "linear2.bias", {C}))
      return WS_ERR_INVALID;
    e->se_blocks.push_back(b);
  }
  if (!require(e, p + "conv.weight", {P, 3 * C, 1}) || !require(e, p + "conv.bias", {P}) ||
      !require(e, p + "pool.linear1.weight", {B, e->spk_glob ? 3 * P : P, 1}) || !require(e, p + "pool.linear1.bias", {B}) ||
      !require(e, p + "pool.linear2.weight", {P, B, 1}) || !require(e, p + "pool.linear2.bias", {P}) ||
      !require(e, p + "bn.weight", {2 * P}) || !require(e, p + "bn.bias", {2 * P}) ||
      !require(e, p + "bn.running_mean", {2 * P}) || !require(e, p + "bn.running_var", {2 * P}) ||
      !require(e, p + "linear.weight", {e->E, 2 * P}) || !require(e, p + "linear.bias", {e->E}))
    return WS_ERR_INVALID;
  e->pool_bn_st = bn_eval_stats(e, p + "bn", 2 * P);
  WS_PTR(e->pool_bn_st);
  if (e->spk_emb_bn) {
    if (!require(e, p + "bn2.weight", {e->E}) || !require(e, p + "bn2.bias", {e->E}) ||
        !require(e, p + "bn2.running_mean", {e->E}) || !require(e, p + "bn2.running_var", {e->E}))
      return WS_ERR_INVALID;
    e->emb_bn_st = bn_eval_stats(e, p + "bn2", e->E);
    WS_PTR(e->emb_bn_st);
  }
  // identity BatchNorm operands (mean 0, rstd 1, gamma 1, beta 0): ws_bn_prelu_fwd then computes y = x + res
  std::vector<float> st(2 * size_t(C), 0.f), one(C, 1.f), zero(C, 0.f);
  for (int c = 0; c < C; ++c) st[C + c] = 1.f;
  e->id_one = upload(e, e->persist, one.data(), one.size());
  e->id_zero = upload(e, e->persist, zero.data(), zero.size());
  const float s0 = 0.f, s1 = 1.f;
  e->slope0 = upload(e, e->persist, &s0, 1);
  e->slope1 = upload(e, e->persist, &s1, 1);
  WS_PTR(e->id_one && e->id_zero && e->slope0 && e->slope1);
  // (0 x C | 1 x C): the [2][c] statistics of any width c <= C start at id_st + C - c
  e->id_st = upload(e, e->persist, st.data(), st.size());
  WS_PTR(e->id_st);
  return WS_OK;
}

// wespeaker CAMPPlus (models/campplus.py, the recipe's alternative speaker encoder: bsrnn.yaml:66-74): FCM head (2-D
// convolutions that stride the mel axis only), D-TDNN backbone of three CAM-dense-TDNN blocks (12 / 24 / 16 layers, growth 32,
// bottleneck 128, kernel 3, dilations 1 / 2 / 2) with transit layers, BN-ReLU, TSTP, dense embedding layer (BatchNorm without
// affine).  Shapes checked against the container; view weights and BatchNorm(eval) statistics prepared once.
int cam_bn_prep(ws_engine* e, const std::string& bn, int c, bool affine, CamBn* b) {
  if (!require(e, bn + ".running_mean", {c}) || !require(e, bn + ".running_var", {c}) ||
      (affine && (!require(e, bn + ".weight", {c}) || !require(e, bn + ".bias", {c}))))
    return WS_ERR_INVALID;
  b->c = c;
  b->st = bn_eval_stats(e, bn, c);
  WS_PTR(b->st);
  b->gamma = affine ? e->dev(bn + ".weight") : e->cam_one;
  b->beta = affine ? e->dev(bn + ".bias") : e->cam_zero;
  return WS_OK;
}

// Conv1d [cout][cin][k] (bias-free, 'same' padding) as the k x k view of the one-row image (prep_tdnn)
const float* cam_view_weight(ws_engine* e, const std::string& conv, int cout, int cin, int k) {
  const float* w = e->host(conv + ".weight");
  std::vector<float> w2(size_t(cout) * k * k * cin, 0.f);
  for (int o = 0; o < cout; ++o)
    for (int ci = 0; ci < cin; ++ci)
      for (int kx = 0; kx < k; ++kx) w2[(size_t(o) * k * k + size_t(k / 2) * k + kx) * cin + ci] = w[(size_t(o) * cin + ci) * k + kx];
  return upload(e, e->persist, w2.data(), w2.size());
}

int prep_campplus(ws_engine* e) {
  const std::string p = "spk_model.";
  const int F = e->feat_dim, mc = 32;
  if (F % 8 || e->E % 4) {
    set_err("engine: CAM++ needs feat_dim %% 8 == 0 and an embedding size %% 4 == 0 (got %d, %d)", F, e->E);
    return WS_ERR_INVALID;
  }
  int rc;
  {   // ones / zeros / identity statistics for widths up to 1024 (affine-free BatchNorm, residual-free adds)
    const int C = 1024;
    std::vector<float> one(C, 1.f), zero(C, 0.f), st(2 * size_t(C), 0.f);
    for (int c = 0; c < C; ++c) st[C + c] = 1.f;
    e->cam_one = upload(e, e->persist, one.data(), one.size());
    e->cam_zero = upload(e, e->persist, zero.data(), zero.size());
    e->cam_id_st = upload(e, e->persist, st.data(), st.size());
    const float s0 = 0.f, s1 = 1.f;
    e->slope0 = upload(e, e->persist, &s0, 1);
    e->slope1 = upload(e, e->persist, &s1, 1);
    WS_PTR(e->cam_one && e->cam_zero && e->cam_id_st && e->slope0 && e->slope1);
  }
  // ---- FCM head ----
  auto add = [&](const std::string& conv, const std::string& bn, int cin, int k, int sh, bool relu, int kind) -> int {
    ConvPrep c;
    const int r = prep_conv(e, p + conv, p + bn, cin, mc, k, sh, relu, &c);
    if (r != WS_OK) return r;
    c.sw = 1;
    e->cam_fcm.push_back(c);
    e->cam_fcm_kind.push_back(kind);
    return WS_OK;
  };
  if ((rc = add("head.conv1", "head.bn1", 1, 3, 1, true, 0)) != WS_OK) return rc;
  for (int L = 1; L <= 2; ++L)
    for (int b = 0; b < 2; ++b) {
      const std::string q = "head.layer" + std::to_string(L) + "." + std::to_string(b) + ".";
      const int sh = b == 0 ? 2 : 1;
      if ((rc = add(q + "conv1", q + "bn1", mc, 3, sh, true, 1)) != WS_OK) return rc;
      if (b == 0 && (rc = add(q + "shortcut.0", q + "shortcut.1", mc, 1, sh, false, 2)) != WS_OK) return rc;
      if ((rc = add(q + "conv2", q + "bn2", mc, 3, 1, true, 3)) != WS_OK) return rc;
    }
  if ((rc = add("head.conv2", "head.bn2", mc, 3, 2, true, 0)) != WS_OK) return rc;
  // ---- D-TDNN backbone ----
  const int cin0 = mc * (F / 8), init = e->cam_init, growth = e->cam_growth, bnc = e->cam_bn;
  const std::string x = p + "xvector.";
  if (!require(e, x + "tdnn.linear.weight", {init, cin0, 5})) return WS_ERR_INVALID;
  e->cam_tdnn_w = cam_view_weight(e, x + "tdnn.linear", init, cin0, 5);
  WS_PTR(e->cam_tdnn_w);
  if ((rc = cam_bn_prep(e, x + "tdnn.nonlinear.batchnorm", init, true, &e->cam_tdnn_bn)) != WS_OK) return rc;
  const int nlayers[3] = {12, 24, 16}, dils[3] = {1, 2, 2};
  int ch = init;
  for (int bi = 0; bi < 3; ++bi) {
    std::vector<CamLayer> layers;
    for (int i = 0; i < nlayers[bi]; ++i) {
      const std::string q = x + "block" + std::to_string(bi + 1) + ".tdnnd" + std::to_string(i + 1) + ".";
      CamLayer l;
      l.cin = ch + i * growth, l.dil = dils[bi];
      if ((rc = cam_bn_prep(e, q + "nonlinear1.batchnorm", l.cin, true, &l.bn1)) != WS_OK) return rc;
      if (!require(e, q + "linear1.weight", {bnc, l.cin, 1})) return WS_ERR_INVALID;
      l.w1 = e->dev(q + "linear1.weight");
      if ((rc = cam_bn_prep(e, q + "nonlinear2.batchnorm", bnc, true, &l.bn2)) != WS_OK) return rc;
      if (!require(e, q + "cam_layer.linear_local.weight", {growth, bnc, 3}) ||
          !require(e, q + "cam_layer.linear1.weight", {bnc / 2, bnc, 1}) || !require(e, q + "cam_layer.linear1.bias", {bnc / 2}) ||
          !require(e, q + "cam_layer.linear2.weight", {growth, bnc / 2, 1}) || !require(e, q + "cam_layer.linear2.bias", {growth}))
        return WS_ERR_INVALID;
      l.wloc = cam_view_weight(e, q + "cam_layer.linear_local", growth, bnc, 3);
      WS_PTR(l.wloc);
      l.l1w = e->dev(q + "cam_layer.linear1.weight"), l.l1b = e->dev(q + "cam_layer.linear1.bias");
      l.l2w = e->dev(q + "cam_layer.linear2.weight"), l.l2b = e->dev(q + "cam_layer.linear2.bias");
      layers.push_back(l);
    }
    e->cam_blocks.push_back(layers);
    ch += nlayers[bi] * growth;
    CamTransit t;
    t.cin = ch, t.cout = ch / 2;
    const std::string q = x + "transit" + std::to_string(bi + 1) + ".";
    if ((rc = cam_bn_prep(e, q + "nonlinear.batchnorm", ch, true, &t.bn)) != WS_OK) return rc;
    if (!require(e, q + "linear.weight", {t.cout, ch, 1})) return WS_ERR_INVALID;
    t.w = e->dev(q + "linear.weight");
    e->cam_transit.push_back(t);
    ch /= 2;
  }
  if (ch > 512) {
    set_err("engine: CAM++ backbone width %d exceeds the plan's buffers", ch);
    return WS_ERR_INVALID;
  }
  if ((rc = cam_bn_prep(e, x + "out_nonlinear.batchnorm", ch, true, &e->cam_out_bn)) != WS_OK) return rc;
  if (!require(e, x + "dense.linear.weight", {e->E, 2 * ch, 1})) return WS_ERR_INVALID;
  if ((rc = cam_bn_prep(e, x + "dense.nonlinear.batchnorm", e->E, false, &e->cam_dense_bn)) != WS_OK) return rc;
  return WS_OK;
}

// kaldi fbank as two GEMMs: every per-frame step before the power spectrum (2^15 scaling, DC removal, 0.97
// pre-emphasis with the first sample replicated, symmetric Hamming window, zero padding, real DFT) folded into one
// [2 * padded/2][win] basis; triangular mel bank [feat_dim][padded/2]   (wesep_amd/utils/funcs.py, DESIGN 11a;
// reference: runtime/frontend/fbank.h:31-222, wesep/utils/funcs.py:91-116)
int prep_fbank(ws_engine* e) {
  const int win = e->sr / 40, shift = e->sr / 100;
  int padded = 1;
  while (padded < win) padded <<= 1;
  const int nf = padded / 2, nb = e->feat_dim;
  e->fb_win = win;
  e->fb_shift = shift;
  e->fb_padded = padded;
  // B = (DFT * window) P D  with P = pre-emphasis, D = I - 11^T/win, applied column by column in double
  std::vector<double> bw_(size_t(2) * nf * win);
  for (int k = 0; k < nf; ++k)
    for (int n = 0; n < win; ++n) {
      const double w = 0.54 - 0.46 * cos(2.0 * M_PI * n / (win - 1));
      const double ang = 2.0 * M_PI * double(k) * n / padded;
      bw_[(size_t(2) * k) * win + n] = cos(ang) * w;
      bw_[(size_t(2) * k + 1) * win + n] = -sin(ang) * w;
    }
  std::vector<float> basis(size_t(2) * nf * win);
  std::vector<double> row(win);
  for (int r = 0; r < 2 * nf; ++r) {
    const double* b = &bw_[size_t(r) * win];
    // (b P)[j] = b[j] - 0.97 b[j+1]  (+ for j = 0: - 0.97 b[0], the replicated first sample)
    for (int j = 0; j < win; ++j) row[j] = b[j] - (j + 1 < win ? 0.97 * b[j + 1] : 0.0);
    row[0] -= 0.97 * b[0];
    double mean = 0.0;
    for (int j = 0; j < win; ++j) mean += row[j];
    mean /= win;
    for (int j = 0; j < win; ++j) basis[size_t(r) * win + j] = static_cast<float>((row[j] - mean) * 32768.0);
  }
  auto mel = [](double f) { return 1127.0 * log(1.0 + f / 700.0); };
  const double lo = mel(20.0), hi = mel(0.5 * e->sr), delta = (hi - lo) / (nb + 1);
  std::vector<float> bank(size_t(nb) * nf, 0.f);
  for (int b = 0; b < nb; ++b) {
    const double left = lo + b * delta, center = left + delta, right = center + delta;
    for (int i = 0; i < nf; ++i) {
      const double m = mel(double(e->sr) / padded * i);
      const double up = (m - left) / (center - left), down = (right - m) / (right - center);
      const double v = up < down ? up : down;
      bank[size_t(b) * nf + i] = v > 0.0 ? static_cast<float>(v) : 0.f;
    }
  }
  std::vector<float> floor_row(nb, -kGnEps);
  e->fb_basis = upload(e, e->persist, basis.data(), basis.size());
  e->fb_bank = upload(e, e->persist, bank.data(), bank.size());
  e->fb_floor = upload(e, e->persist, floor_row.data(), floor_row.size());
  WS_PTR(e->fb_basis && e->fb_bank && e->fb_floor);
  return WS_OK;
}

// PreEmphasis (speaker.py:10-23) + torchaudio MelSpectrogram(n_fft = win_length = 512, hop 128, hamming window buffer,
// HTK filterbank buffer) of spk_feat = False models, as in modules/common/frontend.py: windowed DFT basis
// [2 * 257 (padded to 516)][512] and fb^T [n_mels][257 (padded to 260)] built from the model's own buffers
int prep_mel_frontend(ws_engine* e) {
  const int n = 512, nf = n / 2 + 1, nm = e->feat_dim;
  if (!require(e, "spk_encoder.spectrogram.window", {n}) || !require(e, "spk_encoder.mel_scale.fb", {nf, nm}) ||
      !require(e, "preEmphasis.flipped_filter", {2}))
    return WS_ERR_INVALID;
  const float* win = e->host("spk_encoder.spectrogram.window");
  const float* fb = e->host("spk_encoder.mel_scale.fb");
  e->mel_coef = -e->host("preEmphasis.flipped_filter")[0];
  e->mel_lds = (2 * nf + 3) / 4 * 4;
  e->mel_ldp = (nf + 3) / 4 * 4;
  std::vector<float> basis(size_t(e->mel_lds) * n, 0.f), fbt(size_t(nm) * e->mel_ldp, 0.f);
  for (int k = 0; k < nf; ++k)
    for (int j = 0; j < n; ++j) {
      const double ang = 2.0 * M_PI * double(k) * j / n;
      basis[(size_t(2) * k) * n + j] = static_cast<float>(cos(ang) * win[j]);
      basis[(size_t(2) * k + 1) * n + j] = static_cast<float>(-sin(ang) * win[j]);
    }
  for (int m = 0; m < nm; ++m)
    for (int k = 0; k < nf; ++k) fbt[size_t(m) * e->mel_ldp + k] = fb[size_t(k) * nm + m];
  e->mel_basis = upload(e, e->persist, basis.data(), basis.size());
  e->mel_fbt = upload(e, e->persist, fbt.data(), fbt.size());
  WS_PTR(e->mel_basis && e->mel_fbt);
  return WS_OK;
}

int prepare_tasnet(ws_engine* e);
int prepare_dpccn(ws_engine* e);
int prepare_gridnet(ws_engine* e);

// the speaker-encoder part of the container's meta block (shared by the pBSRNN and DPCCN plans)
int read_speaker_meta(ws_engine* e) {
  e->spk_feat = static_cast<int>(meta_or(e, "spk_feat", 1));
  e->E = static_cast<int>(meta_or(e, "spk_emb_dim", 256));
  e->use_xform = static_cast<int>(meta_or(e, "use_spk_transform", 0));
  e->joint = static_cast<int>(meta_or(e, "joint_training", 0));
  e->feat_dim = static_cast<int>(meta_or(e, "feat_dim", 80));
  for (int i = 0; i < 4; ++i) e->blocks[i] = static_cast<int>(meta_or(e, ("spk_blocks" + std::to_string(i)).c_str(), 0));
  e->spk_kind = static_cast<int>(meta_or(e, "spk_kind", 0));          // 0 wespeaker ResNet, 1 ECAPA-TDNN
  e->spk_channels = static_cast<int>(meta_or(e, "spk_channels", 512));
  e->spk_glob = static_cast<int>(meta_or(e, "spk_glob", 0));
  e->spk_emb_bn = static_cast<int>(meta_or(e, "spk_emb_bn", 0));
  e->spk_bottleneck = static_cast<int>(meta_or(e, "spk_bottleneck", 0));
  e->spk_two_emb = static_cast<int>(meta_or(e, "spk_two_emb", 0));
  if (e->spk_kind < 0 || e->spk_kind > 2) {
    set_err("engine: speaker encoder kind %d is not built (0 ResNet, 1 ECAPA-TDNN, 2 CAM++)", e->spk_kind);
    return WS_ERR_INVALID;
  }
  return WS_OK;
}

int prepare(ws_engine* e) {
  e->arch = static_cast<int>(meta_or(e, "arch", 0));
  if (e->arch == 1) return prepare_tasnet(e);
  if (e->arch == 2) return prepare_dpccn(e);
  if (e->arch == 3) return prepare_gridnet(e);
  if (e->arch != 0) {
    set_err("engine: architecture %d has no launch plan (0 pBSRNN, 1 Conv-TasNet, 2 DPCCN, 3 TF-GridNet)", e->arch);
    return WS_ERR_INVALID;
  }
  e->sr = static_cast<int>(meta_or(e, "sample_rate", 16000));
  e->spk_feat = static_cast<int>(meta_or(e, "spk_feat", 1));
  e->num_repeat = static_cast<int>(meta_or(e, "num_repeat", 6));
  e->E = static_cast<int>(meta_or(e, "spk_emb_dim", 256));
  e->fuse = static_cast<int>(meta_or(e, "spk_fuse_type", 2));
  e->multi_fuse = static_cast<int>(meta_or(e, "multi_fuse", 0));
  e->use_xform = static_cast<int>(meta_or(e, "use_spk_transform", 0));
  e->joint = static_cast<int>(meta_or(e, "joint_training", 0));
  e->feat_dim = static_cast<int>(meta_or(e, "feat_dim", 80));
  for (int i = 0; i < 4; ++i) e->blocks[i] = static_cast<int>(meta_or(e, ("spk_blocks" + std::to_string(i)).c_str(), 0));
  e->spk_kind = static_cast<int>(meta_or(e, "spk_kind", 0));          // 0 wespeaker ResNet, 1 ECAPA-TDNN
  e->spk_channels = static_cast<int>(meta_or(e, "spk_channels", 512));
  e->spk_glob = static_cast<int>(meta_or(e, "spk_glob", 0));
  e->spk_emb_bn = static_cast<int>(meta_or(e, "spk_emb_bn", 0));
  e->spk_bottleneck = static_cast<int>(meta_or(e, "spk_bottleneck", 0));
  e->spk_two_emb = static_cast<int>(meta_or(e, "spk_two_emb", 0));
  if (e->spk_kind < 0 || e->spk_kind > 2) {
    set_err("engine: speaker encoder kind %d is not built (0 ResNet, 1 ECAPA-TDNN, 2 CAM++)", e->spk_kind);
    return WS_ERR_INVALID;
  }
  if (meta_or(e, "win", 512) != 512 || meta_or(e, "stride", 128) != kHop || meta_or(e, "feature_dim", kN) != kN) {
    set_err("engine: built for win 512, stride 128, feature_dim 128");
    return WS_ERR_INVALID;
  }
  if (e->fuse < 0 || e->fuse > 3 || e->num_repeat < 1 || e->E % 4 || e->feat_dim % 8) {
    set_err("engine: unsupported configuration (fuse %d, num_repeat %d, spk_emb_dim %d, feat_dim %d)", e->fuse,
            e->num_repeat, e->E, e->feat_dim);
    return WS_ERR_INVALID;
  }
  band_table(e);
  // weights to the device, once
  e->dw = e->persist.alloc(e->hw.size());
  WS_PTR(e->dw);
  int rc = to_device(e, e->dw, e->hw.data(), e->hw.size() * 4);
  if (rc != WS_OK) return rc;
  std::vector<int> bob, bw2, off2;
  for (int g = 0; g < e->K; ++g) {
    for (int i = 0; i < e->bw[g]; ++i) bob.push_back(g);
    bw2.push_back(2 * e->bw[g]);
    off2.push_back(2 * e->f0[g]);
  }
  e->d_band_of_bin = upload_ints(e, e->persist, bob);
  e->d_f0 = upload_ints(e, e->persist, e->f0);
  e->d_bw = upload_ints(e, e->persist, e->bw);
  e->d_bw2 = upload_ints(e, e->persist, bw2);
  e->d_off2 = upload_ints(e, e->persist, off2);
  WS_PTR(e->d_band_of_bin && e->d_f0 && e->d_bw && e->d_bw2 && e->d_off2);
  // per-band BN / mask parameters
  for (int g = 0; g < e->K; ++g) {
    const std::string b = "BN." + std::to_string(g) + ".", m = "mask." + std::to_string(g) + ".";
    const int bw = e->bw[g];
    if (!require(e, b + "0.weight", {2 * bw}) || !require(e, b + "0.bias", {2 * bw}) ||
        !require(e, b + "1.weight", {kN, 2 * bw}) || !require(e, b + "1.bias", {kN}) ||
        !require(e, m + "0.weight", {kN}) || !require(e, m + "0.bias", {kN}) ||
        !require(e, m + "1.weight", {4 * kN, kN}) || !require(e, m + "1.bias", {4 * kN}) ||
        !require(e, m + "3.weight", {4 * kN, 4 * kN}) || !require(e, m + "3.bias", {4 * kN}) ||
        !require(e, m + "5.weight", {4 * bw, 4 * kN}) || !require(e, m + "5.bias", {4 * bw}))
      return WS_ERR_INVALID;
  }
  // separator.separation layout (bsrnn.py:106-125)
  e->sep_kind.clear();
  if (e->multi_fuse) {
    for (int r = 0; r < e->num_repeat; ++r) {
      e->sep_kind.push_back(0);
      e->sep_kind.push_back(1);
    }
  } else {
    e->sep_kind.push_back(0);
    for (int r = 0; r < e->num_repeat; ++r) e->sep_kind.push_back(1);
  }
  for (size_t i = 0; i < e->sep_kind.size(); ++i) {
    const std::string pre = "separator.separation." + std::to_string(i) + ".";
    if (e->sep_kind[i] == 1) {
      RnnPrep t, b;
      if ((rc = prep_rnn(e, pre + "band_rnn.", &t)) != WS_OK) return rc;
      if ((rc = prep_rnn(e, pre + "band_comm.", &b)) != WS_OK) return rc;
      e->rnn.push_back(t);
      e->rnn.push_back(b);
    } else if (e->fuse == 3) {
      if (!require(e, pre + "fc.gamma_fcs.0.weight", {kN, e->E}) || !require(e, pre + "fc.gamma_fcs.0.bias", {kN}) ||
          !require(e, pre + "fc.beta_fcs.0.weight", {kN, e->E}) || !require(e, pre + "fc.beta_fcs.0.bias", {kN}))
        return WS_ERR_INVALID;
    } else {
      const int in = e->fuse == 0 ? kN + e->E : e->E;
      if (!require(e, pre + "fc.linear.weight", {kN, in}) || !require(e, pre + "fc.linear.bias", {kN})) return WS_ERR_INVALID;
    }
  }
  if (e->use_xform) {
    const Tensor* t0 = e->find("spk_transform.transforms.0.weight");
    if (!t0 || t0->dims.size() < 2 || t0->dims[1] != e->E || !e->find("spk_transform.transforms.1.weight") ||
        !e->find("spk_transform.transforms.3.weight")) {
      set_err("engine: spk_transform tensors missing or mis-shaped");
      return WS_ERR_INVALID;
    }
  }
  if (e->joint) {
    if ((rc = e->spk_kind == 2 ? prep_campplus(e) : e->spk_kind == 1 ? prep_ecapa(e) : prep_resnet(e)) != WS_OK) return rc;
    if ((rc = e->spk_feat ? prep_fbank(e) : prep_mel_frontend(e)) != WS_OK) return rc;
  }
  if (!e->dry && hipStreamSynchronize(e->stream) != hipSuccess) {
    set_err("engine: weight preparation failed on the device");
    return WS_ERR_LAUNCH;
  }
  return WS_OK;
}

// ---- forward pieces ---------------------------------------------------------------------------------------------
int vec_bits(std::initializer_list<long long> dims, int base = 3) {
  for (long long d : dims)
    if (d % 4) return 4;                 // scalar loads, split-bf16 bit kept (falls back to the fp32 kernels)
  return base | 4;
}

// y[M][nout] = act(x[M][k] W[nout][k]^T + bias)      (functional._lin_fwd)
int linear(ws_engine* e, const float* x, int M, int k, const float* W, long long ldw, int nout, const float* bias,
           int act, float* y) {
  ws_gemm_nt_args a = {};
  a.A = x;
  a.W = W;
  a.bias = bias;
  a.C = y;
  a.a_div = kBig;
  a.a_s2 = k;
  a.c_div = kBig;
  a.c_s2 = nout;
  a.st_div1 = 1;
  a.st_div2 = 1;
  a.M = M;
  a.N = nout;
  a.K = k;
  a.ldw = static_cast<int>(ldw);
  a.act = act;
  a.vec = vec_bits({k, ldw});
  WS_RUN(e, ws_gemm_nt(&a, e->stream));
  return WS_OK;
}

int build_descriptors(ws_engine* e, int R, int Tf) {
  if (e->desc_R == R && e->desc_Tf == Tf && e->d_bn) return WS_OK;
  const int K = e->K, H1 = 4 * kN;
  const long long M = (long long)R * Tf;
  std::vector<ws_group_nt> bn(K), l1(K), l2(K), l3(K);
  for (int g = 0; g < K; ++g) {
    const std::string b = "BN." + std::to_string(g) + ".", m = "mask." + std::to_string(g) + ".";
    const int bw = e->bw[g];
    const long long zoff = (long long)g * Tf * kN, hoff = (long long)g * M * H1;
    bn[g] = ws_group_nt{e->dev(b + "1.weight"), e->dev(b + "1.bias"), e->dev(b + "0.weight"), e->dev(b + "0.bias"),
                        2LL * e->f0[g], zoff, g, 2 * bw, kN, 2 * bw, 0};
    l1[g] = ws_group_nt{e->dev(m + "1.weight"), e->dev(m + "1.bias"), e->dev(m + "0.weight"), e->dev(m + "0.bias"),
                        zoff, hoff, g, kN, H1, kN, 0};
    l2[g] = ws_group_nt{e->dev(m + "3.weight"), e->dev(m + "3.bias"), nullptr, nullptr, hoff, hoff, 0, H1, H1, H1, 0};
    l3[g] = ws_group_nt{e->dev(m + "5.weight"), e->dev(m + "5.bias"), nullptr, nullptr, hoff, 4LL * e->f0[g], 0, H1,
                        4 * bw, H1, 0};
  }
  if (!e->d_bn) {
    const size_t nf = (sizeof(ws_group_nt) * K + 3) / 4;
    e->d_bn = reinterpret_cast<ws_group_nt*>(e->persist.alloc(nf));
    e->d_l1 = reinterpret_cast<ws_group_nt*>(e->persist.alloc(nf));
    e->d_l2 = reinterpret_cast<ws_group_nt*>(e->persist.alloc(nf));
    e->d_l3 = reinterpret_cast<ws_group_nt*>(e->persist.alloc(nf));
    WS_PTR(e->d_bn && e->d_l1 && e->d_l2 && e->d_l3);
  }
  const size_t bytes = sizeof(ws_group_nt) * K;
  int rc;
  if ((rc = to_device(e, e->d_bn, bn.data(), bytes)) != WS_OK || (rc = to_device(e, e->d_l1, l1.data(), bytes)) != WS_OK ||
      (rc = to_device(e, e->d_l2, l2.data(), bytes)) != WS_OK || (rc = to_device(e, e->d_l3, l3.data(), bytes)) != WS_OK)
    return rc;
  e->desc_R = R;
  e->desc_Tf = Tf;
  return WS_OK;
}

// ResRNN (bsrnn.py:38-46) on the blocked layout; mirrors functional.ResRNNBlkFn.forward with the packs precomputed
int resrnn(ws_engine* e, const RnnPrep& w, bool time_view, const float* z, int R, int Tf, float* out) {
  const int K = e->K;
  ws_groups_geom geo = {};
  ws_seqmap sm = {};
  long long st_m1, st_m2;
  int st_div1, st_div2;
  if (time_view) {                       // band_rnn: sequences (r, k), steps over t
    geo.ngroups = R * K, geo.gdiv = 1, geo.gs1 = (long long)Tf * kN, geo.gs2 = 0, geo.rs = kN, geo.L = Tf;
    st_div1 = Tf, st_m1 = 1, st_div2 = 1, st_m2 = 0;
    sm.nseq = R * K, sm.sq_div = kBig, sm.sq_s1 = 0, sm.sq_s2 = Tf, sm.step_rows = 1, sm.L = Tf;
  } else {                               // band_comm: sequences (r, t), steps over k
    geo.ngroups = R * Tf, geo.gdiv = Tf, geo.gs1 = (long long)K * Tf * kN, geo.gs2 = kN, geo.rs = (long long)Tf * kN,
    geo.L = K;
    st_div1 = K * Tf, st_m1 = Tf, st_div2 = Tf, st_m2 = 1;
    sm.nseq = R * Tf, sm.sq_div = Tf, sm.sq_s1 = (long long)K * Tf, sm.sq_s2 = 1, sm.step_rows = Tf, sm.L = K;
  }
  geo.W = kN, geo.nbands = 1;
  const int ntile = (sm.nseq + 31) / 32;
  const size_t nb = size_t(ntile) * sm.L;
  const int lmode = 2 * ntile <= 128 ? WS_LSTM_BF16X3_BLK16 : WS_LSTM_BF16X3_BLK;
  static const bool no_cluster = getenv("WS_ENGINE_NO_CLUSTER") != nullptr;   // diagnostics: streaming kernels only
  const bool cluster = !no_cluster && sm.nseq % 64 == 0 && (sm.nseq / 32) * 8 <= e->cu_count && sm.L >= 64;
  const bool fused = !cluster && lmode == WS_LSTM_BF16X3_BLK;
  void* s = e->stream;
  Arena& a = e->work;
  const Arena::Mark mk = a.mark();
  float* stats = a.alloc(size_t(geo.ngroups) * 2);
  float* gates = a.alloc(nb * 32 * 2 * kG4);
  float* cbuf = a.alloc(nb * 32 * 2 * kH);
  float* hcat = a.alloc(nb * 32 * 2 * kH);
  float* xn = a.alloc(nb * 32 * kN);      // normalised input in BL(128): operand of the fused recurrence
  WS_PTR(stats && gates && cbuf && hcat && xn);
  WS_RUN(e, ws_group_stats(z, &geo, kGnEps, stats, s));
  ws_gemm_p2b_args p = {};
  p.A = z;
  p.stats = stats;
  p.gamma = w.norm_w;
  p.beta = w.norm_b;
  p.sm = sm;
  p.lda = kN;
  p.st_div1 = st_div1, p.st_m1 = st_m1, p.st_div2 = st_div2, p.st_m2 = st_m2, p.st_base = 0;
  p.K = kN;
  p.A_bl = xn;
  if (fused) {          // the recurrence computes x W_ih^T itself from the normalised input in BL(128)
    p.N = 0;
    WS_RUN(e, ws_gemm_p2b(&p, s));
    ws_lstm_fused_args f = {};
    f.gates = gates, f.cbuf = cbuf, f.hcat = hcat, f.xn = xn, f.wpack = w.fpack, f.bias = w.bcat;
    f.nseq = sm.nseq, f.L = sm.L;
    WS_RUN(e, ws_lstm_fwd_fused(&f, s));
  } else {
    p.Wpack = w.wih_pack;
    p.bias = w.bcat;
    p.C = gates;
    p.N = 2 * kG4;
    WS_RUN(e, ws_gemm_p2b(&p, s));
    if (cluster) {
      const int ncl = sm.nseq / 32;
      float* xchg = a.alloc(size_t(ncl) * 2 * 8 * 8192 / 4);
      unsigned* flags = reinterpret_cast<unsigned*>(a.alloc(size_t(ncl) * 8 + 8));
      WS_PTR(xchg && flags);
      if (!e->cl_status) {
        e->cl_status = reinterpret_cast<unsigned*>(e->persist.alloc(2));
        WS_PTR(e->cl_status);
        if (zero_device(e, e->cl_status, 8) != WS_OK) return WS_ERR_LAUNCH;
      }
      ws_lstm_cluster_args c = {};
      c.gates = gates, c.cbuf = cbuf, c.hcat = hcat, c.whh_f = w.whf, c.whh_r = w.whr;
      c.xchg = xchg, c.flags = flags, c.nseq = sm.nseq, c.L = sm.L;
      c.status = e->cl_status;
      WS_RUN(e, ws_lstm_fwd_cluster(&c, s));
      // Several engines may share one GPU (separate_main --jobs): the cluster's workgroups are then not guaranteed to
      // be co-resident and a bounded wait can time out.  The streaming pair below is predicated on this launch's
      // timeout word: empty launches after a clean run, the whole layer again after a timeout -- never NaN.
      p.run_if = flags + size_t(ncl) * 8;
      WS_RUN(e, ws_gemm_p2b(&p, s));
      ws_lstm_args l = {};
      l.gates = gates, l.cbuf = cbuf, l.hcat = hcat;
      l.wpack = lmode == WS_LSTM_BF16X3_BLK16 ? w.pack16 : w.pack32;
      l.sq_s1 = sm.sq_s1, l.sq_s2 = sm.sq_s2, l.step_rows = sm.step_rows;
      l.nseq = sm.nseq, l.sq_div = sm.sq_div, l.L = sm.L, l.mode = lmode;
      l.run_if = p.run_if;
      WS_RUN(e, ws_lstm_fwd(&l, s));
    } else {
      ws_lstm_args l = {};
      l.gates = gates, l.cbuf = cbuf, l.hcat = hcat;
      l.wpack = lmode == WS_LSTM_BF16X3_BLK16 ? w.pack16 : w.pack32;
      l.sq_s1 = sm.sq_s1, l.sq_s2 = sm.sq_s2, l.step_rows = sm.step_rows;
      l.nseq = sm.nseq, l.sq_div = sm.sq_div, l.L = sm.L, l.mode = lmode;
      WS_RUN(e, ws_lstm_fwd(&l, s));
    }
  }
  ws_gemm_b2p_args b = {};
  b.A = hcat, b.Wpack = w.proj_pack, b.bias = w.proj_b, b.R = z, b.C = out, b.sm = sm, b.ldc = kN, b.N = kN, b.K = 2 * kH;
  WS_RUN(e, ws_gemm_b2p(&b, s));
  a.release(mk);
  return WS_OK;
}

// speaker fusion on Z (speaker.py:81-125, norm.py:118-139), in place
int fuse_layer(ws_engine* e, const std::string& pre, float* z, const float* emb, int R, int Tf) {
  const int K = e->K, E = e->E;
  const long long P = (long long)R * K * Tf;
  void* s = e->stream;
  Arena& a = e->work;
  const Arena::Mark mk = a.mark();
  float* v = a.alloc(size_t(R) * kN);
  float* v2 = a.alloc(size_t(R) * kN);
  WS_PTR(v && v2);
  int rc;
  if (e->fuse == 3) {          // FiLM: (1 + gamma(e)) z + beta(e)
    if ((rc = linear(e, emb, R, E, e->dev(pre + "fc.gamma_fcs.0.weight"), E, kN, e->dev(pre + "fc.gamma_fcs.0.bias"), 0, v)) != WS_OK ||
        (rc = linear(e, emb, R, E, e->dev(pre + "fc.beta_fcs.0.weight"), E, kN, e->dev(pre + "fc.beta_fcs.0.bias"), 0, v2)) != WS_OK)
      return rc;
    WS_RUN(e, ws_affine_fwd(z, v, v2, 1.0f, P, K * Tf, kN, z, s));
  } else if (e->fuse == 0) {   // concat: Linear(cat[z, e]) = z Wz^T + (e We^T + b)
    const float* W = e->dev(pre + "fc.linear.weight");
    if ((rc = linear(e, emb, R, E, W + kN, kN + E, kN, e->dev(pre + "fc.linear.bias"), 0, v)) != WS_OK) return rc;
    float* t = a.alloc(size_t(P) * kN);
    WS_PTR(t);
    ws_gemm_nt_args g = {};
    g.A = z, g.W = W, g.C = t;
    g.a_div = kBig, g.a_s2 = kN, g.c_div = kBig, g.c_s2 = kN, g.st_div1 = 1, g.st_div2 = 1;
    g.M = static_cast<int>(P), g.N = kN, g.K = kN, g.ldw = kN + E, g.vec = 3 | 4;
    WS_RUN(e, ws_gemm_nt(&g, s));
    WS_RUN(e, ws_affine_fwd(t, nullptr, v, 1.0f, P, K * Tf, kN, z, s));
  } else {
    if ((rc = linear(e, emb, R, E, e->dev(pre + "fc.linear.weight"), E, kN, e->dev(pre + "fc.linear.bias"), 0, v)) != WS_OK)
      return rc;
    if (e->fuse == 2)
      WS_RUN(e, ws_affine_fwd(z, v, nullptr, 0.0f, P, K * Tf, kN, z, s));     // multiply
    else
      WS_RUN(e, ws_affine_fwd(z, nullptr, v, 1.0f, P, K * Tf, kN, z, s));     // additive
  }
  a.release(mk);
  return WS_OK;
}

// conv + BatchNorm(eval) + ReLU/identity (+ residual), channels-last (functional_resnet.py:15-46).  The convolution is
// one GEMM on the implicit patch matrix of x (ws_conv_view, nothing materialised); the 1-channel stem, whose patch
// rows are not float4-addressable, writes its 9-column patch matrix with ws_im2col first.
int conv_bn_act(ws_engine* e, const ConvPrep& c, const float* x, const float* res, int R, int H, int W, float* y,
                int* Ho_out, int* Wo_out) {
  const int pad = c.k / 2, sw = c.sw ? c.sw : c.stride;
  const int Ho = (H + 2 * pad - c.k) / c.stride + 1, Wo = (W + 2 * pad - c.k) / sw + 1;
  const long long M = (long long)R * Ho * Wo;
  const bool implicit = c.cin % 4 == 0 && (long long)H * W * c.cin < 0x7fffffffLL;
  void* s = e->stream;
  Arena& a = e->work;
  const Arena::Mark mk = a.mark();
  float* conv = a.alloc(size_t(M) * c.cout);
  float* u = a.alloc(size_t(M) * c.cout);
  WS_PTR(conv && u);
  ws_gemm_nt_args g = {};
  g.W = c.w2, g.C = conv;
  g.a_div = kBig, g.a_s2 = c.ldp, g.c_div = kBig, g.c_s2 = c.cout, g.st_div1 = 1, g.st_div2 = 1;
  g.M = static_cast<int>(M), g.N = c.cout, g.K = c.ldp, g.ldw = c.ldp, g.vec = 3 | 4;
  if (implicit) {
    g.A = x;
    g.conv.on = 1, g.conv.mode = 0, g.conv.H = H, g.conv.W = W, g.conv.C = c.cin, g.conv.Ho = Ho, g.conv.Wo = Wo;
    g.conv.k = c.k, g.conv.sh = c.stride, g.conv.sw = sw, g.conv.p = pad, g.conv.dil = 1;
  } else {
    if (sw != c.stride) {
      set_err("engine: a convolution with different strides along H and W needs cin %% 4 == 0 (the implicit-patch view)");
      return WS_ERR_INVALID;
    }
    float* patches = a.alloc(size_t(M) * c.ldp);
    WS_PTR(patches);
    if (c.ldp != c.k * c.k * c.cin) {
      const int rc = zero_device(e, patches, size_t(M) * c.ldp * 4);
      if (rc != WS_OK) return rc;
    }
    WS_RUN(e, ws_im2col(x, R, H, W, c.cin, c.k, c.stride, pad, c.ldp, patches, s));
    g.A = patches;
  }
  WS_RUN(e, ws_gemm_nt(&g, s));
  WS_RUN(e, ws_bn_prelu_fwd(conv, c.st, c.gamma, c.beta, res, c.relu ? e->slope0 : e->slope1, M, c.cout, u, y, s));
  a.release(mk);
  *Ho_out = Ho;
  *Wo_out = Wo;
  return WS_OK;
}

// fbank [R][Te][F] (device) -> embedding [R][E]   (wespeaker ResNet, eval mode; models/resnet.py: BasicBlock and
// Bottleneck stacks, TSTP, one or two embedding layers)
int resnet_embed(ws_engine* e, const float* fbank, int R, int Te, float* emb) {
  const int F = e->feat_dim, ex = e->spk_bottleneck ? 4 : 1;
  void* s = e->stream;
  Arena& a = e->work;
  const Arena::Mark mk = a.mark();
  // [R][Te][F] -> [R][F][Te][1]
  float* x = a.alloc(size_t(R) * F * Te);
  WS_PTR(x);
  for (int r = 0; r < R; ++r)
    WS_RUN(e, ws_transpose(fbank + size_t(r) * Te * F, Te, F, F, x + size_t(r) * F * Te, s));
  int H = F, W = Te, Ho, Wo, rc;
  // rotating activation buffers sized for the largest activation (the first stage's output: 32 * ex channels)
  const size_t act = size_t(R) * H * W * 32 * ex;
  float* bufs[4] = {a.alloc(act), a.alloc(act), a.alloc(act), nullptr};
  bufs[3] = e->spk_bottleneck ? a.alloc(act) : bufs[0];        // BasicBlock stacks rotate through three
  WS_PTR(bufs[0] && bufs[1] && bufs[2] && bufs[3]);
  if ((rc = conv_bn_act(e, e->stem, x, nullptr, R, H, W, bufs[0], &Ho, &Wo)) != WS_OK) return rc;
  int cur = 0, C = 32;
  for (const BlockPrep& b : e->res_blocks) {
    float* y = bufs[cur];
    float* t1 = bufs[(cur + 1) % 4];
    float* t2 = bufs[(cur + 2) % 4];
    float* t3 = bufs[(cur + 3) % 4];
    int H1, W1, H2, W2, Hs, Ws;
    const float* shortcut = y;
    if (e->spk_bottleneck) {
      if ((rc = conv_bn_act(e, b.c1, y, nullptr, R, H, W, t1, &H1, &W1)) != WS_OK) return rc;
      if ((rc = conv_bn_act(e, b.c2, t1, nullptr, R, H1, W1, t2, &H2, &W2)) != WS_OK) return rc;
      if (b.has_sc) {
        if ((rc = conv_bn_act(e, b.sc, y, nullptr, R, H, W, t1, &Hs, &Ws)) != WS_OK) return rc;
        shortcut = t1;
      }
      if ((rc = conv_bn_act(e, b.c3, t2, shortcut, R, H2, W2, t3, &H2, &W2)) != WS_OK) return rc;
      cur = (cur + 3) % 4;
      C = b.c3.cout;
    } else {                     // three of the buffers: conv2 writes over the block input unless that is the shortcut
      float* o = bufs[(cur + 1) % 3];
      float* sc = bufs[(cur + 2) % 3];
      if ((rc = conv_bn_act(e, b.c1, y, nullptr, R, H, W, o, &H1, &W1)) != WS_OK) return rc;
      if (b.has_sc) {
        if ((rc = conv_bn_act(e, b.sc, y, nullptr, R, H, W, sc, &Hs, &Ws)) != WS_OK) return rc;
        shortcut = sc;
      }
      float* dst = b.has_sc ? y : sc;
      if ((rc = conv_bn_act(e, b.c2, o, shortcut, R, H1, W1, dst, &H2, &W2)) != WS_OK) return rc;
      cur = b.has_sc ? cur : (cur + 2) % 3;
      C = b.c2.cout;
    }
    H = H2, W = W2;
  }
  float* stats = a.alloc(size_t(R) * 2 * C * H);
  WS_PTR(stats);
  WS_RUN(e, ws_tstp_fwd(bufs[cur], R, H, W, C, kTstpEps, stats, s));
  const std::string p = "spk_model.";
  if (!e->spk_two_emb) {
    rc = linear(e, stats, R, 2 * C * H, e->dev(p + "seg_1.weight"), 2 * C * H, e->E, e->dev(p + "seg_1.bias"), 0, emb);
  } else {
    float* t = a.alloc(size_t(R) * e->E);
    float* u = a.alloc(size_t(R) * e->E);
    float* v = a.alloc(size_t(R) * e->E);
    WS_PTR(t && u && v);
    if ((rc = linear(e, stats, R, 2 * C * H, e->dev(p + "seg_1.weight"), 2 * C * H, e->E, e->dev(p + "seg_1.bias"), 2, t)) != WS_OK)
      return rc;
    WS_RUN(e, ws_bn_prelu_fwd(t, e->seg_bn_st, e->id_one, e->id_zero, nullptr, e->slope1, R, e->E, u, v, s));
    rc = linear(e, v, R, e->E, e->dev(p + "seg_2.weight"), e->E, e->E, e->dev(p + "seg_2.bias"), 0, emb);
  }
  a.release(mk);
  return rc;
}

// dst[m][0:width] = src[m][0:width] for `rows` rows with row strides ldd / lds (floats): channel slices of
// channels-last activations (torch.split / torch.cat of the Res2Net branches and the layer aggregation)
int copy_cols(ws_engine* e, float* dst, long long ldd, const float* src, long long lds, int width, long long rows) {
  ++e->n_launches;
  if (e->dry) return WS_OK;
  if (hipMemcpy2DAsync(dst, size_t(ldd) * 4, src, size_t(lds) * 4, size_t(width) * 4, size_t(rows), hipMemcpyDeviceToDevice,
                       e->stream) != hipSuccess) {
    set_err("engine: strided device copy failed");
    return WS_ERR_LAUNCH;
  }
  return WS_OK;
}

// y = x + res on [M][c] (c <= spk_channels): the BatchNorm kernel with identity operands
int add_rows(ws_engine* e, const float* x, const float* res, long long M, int c, float* scratch, float* y) {
  WS_RUN(e, ws_bn_prelu_fwd(x, e->id_st + (e->spk_channels - c), e->id_one, e->id_zero, res, e->slope1, M, c, scratch, y,
                            e->stream));
  return WS_OK;
}

// y [M][cout] = BN(ReLU(conv1d(x [R][T][cin]))), x rows lda apart (k == 1) or dense (k > 1)   (functional_ecapa.py:23-51)
int tdnn(ws_engine* e, const TdnnPrep& t, const float* x, long long lda, int R, int T, float* y) {
  const long long M = (long long)R * T;
  void* s = e->stream;
  Arena& a = e->work;
  const Arena::Mark mk = a.mark();
  float* c = a.alloc(size_t(M) * t.cout);
  float* u = a.alloc(size_t(M) * t.cout);
  WS_PTR(c && u);
  ws_gemm_nt_args g = {};
  g.A = x, g.W = t.w, g.bias = t.bias, g.C = c;
  g.a_div = kBig, g.a_s2 = lda, g.c_div = kBig, g.c_s2 = t.cout, g.st_div1 = 1, g.st_div2 = 1;
  g.M = static_cast<int>(M), g.N = t.cout, g.act = 2, g.vec = 3 | 4;
  if (t.k == 1) {
    g.K = t.cin, g.ldw = t.cin;
  } else {
    g.K = t.k * t.k * t.cin, g.ldw = g.K;
    g.conv.on = 1, g.conv.mode = 0, g.conv.H = 1, g.conv.W = T, g.conv.C = t.cin, g.conv.Ho = 1, g.conv.Wo = T;
    g.conv.k = t.k, g.conv.sh = 1, g.conv.sw = 1, g.conv.p = t.dil * (t.k / 2), g.conv.dil = t.dil;
  }
  WS_RUN(e, ws_gemm_nt(&g, s));
  WS_RUN(e, ws_bn_prelu_fwd(c, t.st, t.gamma, t.beta, nullptr, e->slope1, M, t.cout, u, y, s));
  a.release(mk);
  return WS_OK;
}

// mean over the T frames of each utterance, [R*T][C] -> sums [R][2][C] scaled by 1/T (the first C of each row)
int time_mean(ws_engine* e, const float* x, int R, int T, int C, float* mean2) {
  void* s = e->stream;
  Arena& a = e->work;
  int nsplit = T / 32;
  const int cap = 1024 / R > 1 ? 1024 / R : 1;
  if (nsplit > cap) nsplit = cap;
  if (nsplit < 1) nsplit = 1;
  float* slab = a.alloc(size_t(nsplit) * R * 2 * C);
  float* sums = a.alloc(size_t(R) * 2 * C);
  WS_PTR(slab && sums);
  WS_RUN(e, ws_chan_sums(x, nullptr, nullptr, 1, T, R, nsplit, C, slab, s));
  WS_RUN(e, ws_reduce_slabs(slab, nsplit, (long long)R * 2 * C, (long long)R * 2 * C, sums, 0, 0, s));
  WS_RUN(e, ws_bcast_rows(sums, 1.0f / T, 1, R, 2 * C, mean2, s));
  return WS_OK;
}

// SE_Res2Block (models/ecapa_tdnn.py:78-92): x [M][C] dense -> out [M][C] dense
int se_res2_block(ws_engine* e, const SeRes2Prep& b, const float* x, int R, int T, float* out) {
  const int C = e->spk_channels, scale = 8, w = C / scale, B = 128;
  const long long M = (long long)R * T;
  void* s = e->stream;
  Arena& a = e->work;
  const Arena::Mark mk = a.mark();
  float* h = a.alloc(size_t(M) * C);          // first 1x1 TDNN
  float* r2 = a.alloc(size_t(M) * C);         // the branches' outputs, concatenated
  float* slice = a.alloc(size_t(M) * w);
  float* in = a.alloc(size_t(M) * w);
  float* y = a.alloc(size_t(M) * w);
  float* scratch = a.alloc(size_t(M) * C);
  WS_PTR(h && r2 && slice && in && y && scratch);
  int rc;
  if ((rc = tdnn(e, b.in, x, C, R, T, h)) != WS_OK) return rc;
  for (int i = 0; i < scale - 1; ++i) {       // group i >= 1 adds the previous group's output before its own TDNN
    const float* src = slice;
    if ((rc = copy_cols(e, slice, w, h + size_t(i) * w, C, w, M)) != WS_OK) return rc;
    if (i > 0) {
      if ((rc = add_rows(e, y, slice, M, w, scratch, in)) != WS_OK) return rc;
      src = in;
    }
    if ((rc = tdnn(e, b.branch[i], src, w, R, T, y)) != WS_OK) return rc;
    if ((rc = copy_cols(e, r2 + size_t(i) * w, C, y, w, w, M)) != WS_OK) return rc;
  }
  if ((rc = copy_cols(e, r2 + size_t(scale - 1) * w, C, h + size_t(scale - 1) * w, C, w, M)) != WS_OK) return rc;
  if ((rc = tdnn(e, b.out, r2, C, R, T, h)) != WS_OK) return rc;
  // squeeze-excitation: gate [R][C] = sigmoid(W2 relu(W1 mean_t + b1) + b2), broadcast over the frames
  float* mean2 = a.alloc(size_t(R) * 2 * C);
  float* z = a.alloc(size_t(R) * B);
  float* gate = a.alloc(size_t(R) * C);
  WS_PTR(mean2 && z && gate);
  if ((rc = time_mean(e, h, R, T, C, mean2)) != WS_OK) return rc;
  {
    ws_gemm_nt_args g = {};
    g.A = mean2, g.W = e->dev(b.se + "linear1.weight"), g.bias = e->dev(b.se + "linear1.bias"), g.C = z;
    g.a_div = kBig, g.a_s2 = 2 * C, g.c_div = kBig, g.c_s2 = B, g.st_div1 = 1, g.st_div2 = 1;
    g.M = R, g.N = B, g.K = C, g.ldw = C, g.act = 2, g.vec = 3 | 4;
    WS_RUN(e, ws_gemm_nt(&g, s));
  }
  if ((rc = linear(e, z, R, B, e->dev(b.se + "linear2.weight"), B, C, e->dev(b.se + "linear2.bias"), 0, gate)) != WS_OK)
    return rc;
  WS_RUN(e, ws_rowbias_act_fwd(gate, nullptr, R, C, 1, 3, gate, s));
  WS_RUN(e, ws_bcast_rows(gate, 1.0f, T, M, C, r2, s));
  WS_RUN(e, ws_maskmul_fwd(h, C, r2, M, C, scratch, s));
  float* u = r2;                               // free again: pre-activation scratch of the residual add
  if ((rc = add_rows(e, scratch, x, M, C, u, out)) != WS_OK) return rc;
  a.release(mk);
  return WS_OK;
}

// fbank [R][Te][F] (device) -> embedding [R][E]   (wespeaker ECAPA-TDNN, eval mode; models/ecapa_tdnn.py:135-160)
int ecapa_embed(ws_engine* e, const float* fbank, int R, int Te, float* emb) {
  const int C = e->spk_channels, P = 1536, B = 128, T = Te;
  const long long M = (long long)R * T;
  const std::string p = "spk_model.";
  void* s = e->stream;
  Arena& a = e->work;
  const Arena::Mark mk = a.mark();
  float* cur = a.alloc(size_t(M) * C);
  float* nxt = a.alloc(size_t(M) * C);
  float* cat = a.alloc(size_t(M) * 3 * C);
  WS_PTR(cur && nxt && cat);
  int rc;
  if ((rc = tdnn(e, e->tdnn1, fbank, e->feat_dim, R, T, cur)) != WS_OK) return rc;
  for (int li = 0; li < 3; ++li) {
    if ((rc = se_res2_block(e, e->se_blocks[li], cur, R, T, nxt)) != WS_OK) return rc;
    if ((rc = copy_cols(e, cat + size_t(li) * C, 3 * C, nxt, C, C, M)) != WS_OK) return rc;
    std::swap(cur, nxt);
  }
  float* h = a.alloc(size_t(M) * P);           // relu(conv1x1(cat)): the pooled sequence
  float* att = a.alloc(size_t(M) * B);
  float* logits = a.alloc(size_t(M) * P);
  float* pooled = a.alloc(size_t(R) * 2 * P);
  float* aux = a.alloc(size_t(R) * 4 * P);
  float* normed = a.alloc(size_t(R) * 2 * P);
  float* u = a.alloc(size_t(R) * 2 * P);
  WS_PTR(h && att && logits && pooled && aux && normed && u);
  if ((rc = linear(e, cat, static_cast<int>(M), 3 * C, e->dev(p + "conv.weight"), 3 * C, P, e->dev(p + "conv.bias"), 2, h)) != WS_OK)
    return rc;
  // attentive statistics pooling (models/ecapa_tdnn.py:96-123); global context: cat(x, mean, std) W1^T =
  // x Wx^T + (mean Wm^T + std Ws^T + b1), the context a per-utterance bias of the bottleneck
  const float* W1 = e->dev(p + "pool.linear1.weight");
  const float* rowbias = nullptr;
  if (e->spk_glob) {
    float* ctx = a.alloc(size_t(R) * 2 * P);
    float* rb = a.alloc(size_t(R) * B);
    WS_PTR(ctx && rb);
    WS_RUN(e, ws_tstp_fwd(h, R, 1, T, P, kTstpEps, ctx, s));
    if ((rc = linear(e, ctx, R, 2 * P, W1 + P, 3 * P, B, e->dev(p + "pool.linear1.bias"), 0, rb)) != WS_OK) return rc;
    if ((rc = linear(e, h, static_cast<int>(M), P, W1, 3 * P, B, nullptr, 0, att)) != WS_OK) return rc;
    rowbias = rb;
  } else {
    if ((rc = linear(e, h, static_cast<int>(M), P, W1, P, B, e->dev(p + "pool.linear1.bias"), 0, att)) != WS_OK) return rc;
  }
  WS_RUN(e, ws_rowbias_act_fwd(att, rowbias, M, B, T, 1, att, s));
  if ((rc = linear(e, att, static_cast<int>(M), B, e->dev(p + "pool.linear2.weight"), B, P, e->dev(p + "pool.linear2.bias"), 0,
                   logits)) != WS_OK)
    return rc;
  WS_RUN(e, ws_astp_fwd(h, logits, R, T, P, kAstpFloor, pooled, aux, s));
  WS_RUN(e, ws_bn_prelu_fwd(pooled, e->pool_bn_st, e->dev(p + "bn.weight"), e->dev(p + "bn.bias"), nullptr, e->slope1, R,
                            2 * P, u, normed, s));
  if (e->spk_emb_bn) {
    float* raw = a.alloc(size_t(R) * e->E);
    float* u2 = a.alloc(size_t(R) * e->E);
    WS_PTR(raw && u2);
    if ((rc = linear(e, normed, R, 2 * P, e->dev(p + "linear.weight"), 2 * P, e->E, e->dev(p + "linear.bias"), 0, raw)) != WS_OK)
      return rc;
    WS_RUN(e, ws_bn_prelu_fwd(raw, e->emb_bn_st, e->dev(p + "bn2.weight"), e->dev(p + "bn2.bias"), nullptr, e->slope1, R, e->E,
                              u2, emb, s));
  } else if ((rc = linear(e, normed, R, 2 * P, e->dev(p + "linear.weight"), 2 * P, e->E, e->dev(p + "linear.bias"), 0, emb)) !=
             WS_OK) {
    return rc;
  }
  a.release(mk);
  return WS_OK;
}

// ---- CAM++ forward (models/campplus.py, eval mode) ---------------------------------------------------------------------
// y = act(BatchNorm(x)) on dense rows [M][c]
int cam_bn_act(ws_engine* e, const CamBn& b, const float* x, long long M, bool relu, float* scratch, float* y) {
  WS_RUN(e, ws_bn_prelu_fwd(x, b.st, b.gamma, b.beta, nullptr, relu ? e->slope0 : e->slope1, M, b.c, scratch, y, e->stream));
  return WS_OK;
}

// y [M][nout] = x [M][k] (rows lda apart) W^T + bias
int cam_lin(ws_engine* e, const float* x, long long lda, long long M, int k, const float* W, int nout, const float* bias, float* y) {
  ws_gemm_nt_args a = {};
  a.A = x, a.W = W, a.bias = bias, a.C = y;
  a.a_div = kBig, a.a_s2 = lda, a.c_div = kBig, a.c_s2 = nout, a.st_div1 = 1, a.st_div2 = 1;
  a.M = static_cast<int>(M), a.N = nout, a.K = k, a.ldw = k;
  a.vec = vec_bits({(long long)k, lda});
  WS_RUN(e, ws_gemm_nt(&a, e->stream));
  return WS_OK;
}

// y [R*To][cout] = conv1d(x [R][T][cin], k taps, dilation dil, stride sw, 'same' padding), bias-free: the k x k view of the
// one-row image (functional_campplus.Conv1dFn)
int cam_conv(ws_engine* e, const float* x, int R, int T, int cin, const float* Wv, int cout, int k, int dil, int sw, float* y,
             int* To_out) {
  const int p = dil * (k / 2), To = (T + 2 * p - dil * (k - 1) - 1) / sw + 1;
  ws_gemm_nt_args g = {};
  g.A = x, g.W = Wv, g.C = y;
  g.a_div = kBig, g.a_s2 = cin, g.c_div = kBig, g.c_s2 = cout, g.st_div1 = 1, g.st_div2 = 1;
  g.M = R * To, g.N = cout, g.K = k * k * cin, g.ldw = g.K, g.vec = 3 | 4;
  g.conv.on = 1, g.conv.mode = 0, g.conv.H = 1, g.conv.W = T, g.conv.C = cin, g.conv.Ho = 1, g.conv.Wo = To;
  g.conv.k = k, g.conv.sh = 1, g.conv.sw = sw, g.conv.p = p, g.conv.dil = dil;
  WS_RUN(e, ws_gemm_nt(&g, e->stream));
  *To_out = To;
  return WS_OK;
}

// CAMDenseTDNNLayer (campplus.py:120-170): x = the first l.cin columns of `cat` (rows ld apart) -> 32 new channels written
// behind them.  Context-aware mask: m = sigmoid(W2 relu(W1 (segment mean + utterance mean) + b1) + b2) per 100-frame segment
int cam_layer(ws_engine* e, const CamLayer& l, float* cat, long long ld, int R, int T) {
  const int bnc = e->cam_bn, growth = e->cam_growth, hid = bnc / 2, seg = 100, nseg = (T + seg - 1) / seg;
  const long long M = (long long)R * T, Ms = (long long)R * nseg;
  void* s = e->stream;
  Arena& a = e->work;
  const Arena::Mark mk = a.mark();
  float* xin = a.alloc(size_t(M) * l.cin);
  float* u = a.alloc(size_t(M) * (l.cin > bnc ? l.cin : bnc));
  float* x1 = a.alloc(size_t(M) * l.cin);
  float* h = a.alloc(size_t(M) * bnc);
  float* x2 = a.alloc(size_t(M) * bnc);
  float* yl = a.alloc(size_t(M) * growth);
  float* sums = a.alloc(size_t(Ms) * bnc);
  float* mseg = a.alloc(size_t(Ms) * bnc);
  float* mean2 = a.alloc(size_t(R) * 2 * bnc);
  float* rb = a.alloc(size_t(R) * hid);
  float* rbf = a.alloc(size_t(Ms) * hid);
  float* g1 = a.alloc(size_t(Ms) * hid);
  float* g1u = a.alloc(size_t(Ms) * hid);
  float* hh = a.alloc(size_t(Ms) * hid);
  float* m = a.alloc(size_t(Ms) * growth);
  WS_PTR(xin && u && x1 && h && x2 && yl && sums && mseg && mean2 && rb && rbf && g1 && g1u && hh && m);
  int rc, To;
  if ((rc = copy_cols(e, xin, l.cin, cat, ld, l.cin, M)) != WS_OK) return rc;
  if ((rc = cam_bn_act(e, l.bn1, xin, M, true, u, x1)) != WS_OK) return rc;
  if ((rc = cam_lin(e, x1, l.cin, M, l.cin, l.w1, bnc, nullptr, h)) != WS_OK) return rc;
  if ((rc = cam_bn_act(e, l.bn2, h, M, true, u, x2)) != WS_OK) return rc;
  if ((rc = cam_conv(e, x2, R, T, bnc, l.wloc, growth, 3, l.dil, 1, yl, &To)) != WS_OK) return rc;
  // context: mean over each segment (the last one may be shorter) + mean over the utterance
  WS_RUN(e, ws_seg_sums(x2, nullptr, R, T, bnc, seg, sums, s));
  WS_RUN(e, ws_bcast_rows(sums, 1.0f / seg, 1, Ms, bnc, mseg, s));
  const int last = T - (nseg - 1) * seg;
  if (last != seg)
    for (int r = 0; r < R; ++r) {
      const size_t o = (size_t(r) * nseg + nseg - 1) * bnc;
      WS_RUN(e, ws_bcast_rows(sums + o, 1.0f / last, 1, 1, bnc, mseg + o, s));
    }
  if ((rc = time_mean(e, x2, R, T, bnc, mean2)) != WS_OK) return rc;
  // W1 (mean_seg + mean_all) + b1 = W1 mean_seg + (W1 mean_all + b1): the utterance part is a per-row bias
  if ((rc = cam_lin(e, mean2, 2 * bnc, R, bnc, l.l1w, hid, l.l1b, rb)) != WS_OK) return rc;
  if ((rc = cam_lin(e, mseg, bnc, Ms, bnc, l.l1w, hid, nullptr, g1)) != WS_OK) return rc;
  WS_RUN(e, ws_bcast_rows(rb, 1.0f, nseg, Ms, hid, rbf, s));
  WS_RUN(e, ws_bn_prelu_fwd(g1, e->cam_id_st + (1024 - hid), e->cam_one, e->cam_zero, rbf, e->slope0, Ms, hid, g1u, hh, s));
  if ((rc = cam_lin(e, hh, hid, Ms, hid, l.l2w, growth, l.l2b, m)) != WS_OK) return rc;
  WS_RUN(e, ws_rowbias_act_fwd(m, nullptr, Ms, growth, 1, 3, m, s));
  WS_RUN(e, ws_seg_scale(yl, m, R, T, growth, seg, yl, s));
  if ((rc = copy_cols(e, cat + l.cin, ld, yl, growth, growth, M)) != WS_OK) return rc;
  a.release(mk);
  return WS_OK;
}

// fbank [R][Te][F] (device) -> embedding [R][E]   (wespeaker CAMPPlus, eval mode; models/campplus.py:225-262)
int campplus_embed(ws_engine* e, const float* fbank, int R, int Te, float* emb) {
  const int F = e->feat_dim, mc = 32;
  void* s = e->stream;
  Arena& a = e->work;
  const Arena::Mark mk = a.mark();
  int rc;
  // ---- FCM head on [R][F][Te][1]; the mel axis is strided three times, the frame axis never ----
  float* x = a.alloc(size_t(R) * F * Te);
  WS_PTR(x);
  for (int r = 0; r < R; ++r) WS_RUN(e, ws_transpose(fbank + size_t(r) * Te * F, Te, F, F, x + size_t(r) * F * Te, s));
  const size_t act = size_t(R) * F * Te * mc;
  float* bufs[3] = {a.alloc(act), a.alloc(act), a.alloc(act)};
  WS_PTR(bufs[0] && bufs[1] && bufs[2]);
  int H = F, W = Te, Ho, Wo;
  const float* cur = x;
  int ci = 0;                 // buffer that holds `cur` (-1: x)
  auto other = [&](int a0, int a1) { for (int i = 0; i < 3; ++i) if (i != a0 && i != a1) return i; return 0; };
  size_t k = 0;
  {
    if ((rc = conv_bn_act(e, e->cam_fcm[k++], cur, nullptr, R, H, W, bufs[0], &Ho, &Wo)) != WS_OK) return rc;
    cur = bufs[0], ci = 0;
  }
  while (k < e->cam_fcm.size()) {
    const int kind = e->cam_fcm_kind[k];
    if (kind == 0) {          // the final strided convolution
      const int o = other(ci, ci);
      if ((rc = conv_bn_act(e, e->cam_fcm[k++], cur, nullptr, R, H, W, bufs[o], &Ho, &Wo)) != WS_OK) return rc;
      cur = bufs[o], ci = o, H = Ho, W = Wo;
      continue;
    }
    // BasicResBlock: conv1 [, shortcut], conv2 + residual
    const int o1 = other(ci, ci);
    int H1, W1;
    if ((rc = conv_bn_act(e, e->cam_fcm[k++], cur, nullptr, R, H, W, bufs[o1], &H1, &W1)) != WS_OK) return rc;
    const float* sc = cur;
    int o2 = other(ci, o1);
    if (e->cam_fcm_kind[k] == 2) {
      int Hs, Ws;
      if ((rc = conv_bn_act(e, e->cam_fcm[k++], cur, nullptr, R, H, W, bufs[o2], &Hs, &Ws)) != WS_OK) return rc;
      sc = bufs[o2];
      // conv2 may now overwrite the block input
      if ((rc = conv_bn_act(e, e->cam_fcm[k++], bufs[o1], sc, R, H1, W1, bufs[ci], &Ho, &Wo)) != WS_OK) return rc;
      cur = bufs[ci];
    } else {
      if ((rc = conv_bn_act(e, e->cam_fcm[k++], bufs[o1], sc, R, H1, W1, bufs[o2], &Ho, &Wo)) != WS_OK) return rc;
      cur = bufs[o2], ci = o2;
    }
    H = Ho, W = Wo;
  }
  // [R][H'][T][32] -> [R*T][32 * H'] with channel index c * H' + h (the reference's reshape of [B, C, H', T])
  const int Hp = H, T0 = W, c0 = mc * Hp;
  float* feat = a.alloc(size_t(R) * T0 * c0);
  WS_PTR(feat);
  for (int r = 0; r < R; ++r)
    WS_RUN(e, ws_transpose(cur + size_t(r) * Hp * T0 * mc, Hp, T0 * mc, T0 * mc, feat + size_t(r) * T0 * c0, s));
  // ---- D-TDNN backbone ----
  int T;
  const int init = e->cam_init, growth = e->cam_growth;
  const int Tmax = (T0 - 1) / 2 + 1;
  float* t0 = a.alloc(size_t(R) * Tmax * init);
  float* scratch = a.alloc(size_t(R) * Tmax * 1024);
  WS_PTR(t0 && scratch);
  if ((rc = cam_conv(e, feat, R, T0, c0, e->cam_tdnn_w, init, 5, 1, 2, t0, &T)) != WS_OK) return rc;
  const long long M = (long long)R * T;
  int ch = init;
  float* y = a.alloc(size_t(M) * init);
  WS_PTR(y);
  if ((rc = cam_bn_act(e, e->cam_tdnn_bn, t0, M, true, scratch, y)) != WS_OK) return rc;
  for (size_t bi = 0; bi < e->cam_blocks.size(); ++bi) {
    const std::vector<CamLayer>& layers = e->cam_blocks[bi];
    const long long ld = ch + (long long)layers.size() * growth;
    float* cat = a.alloc(size_t(M) * ld);
    float* tin = a.alloc(size_t(M) * ld);
    float* tout = a.alloc(size_t(M) * (ld / 2));
    WS_PTR(cat && tin && tout);
    if ((rc = copy_cols(e, cat, ld, y, ch, ch, M)) != WS_OK) return rc;
    for (const CamLayer& l : layers)
      if ((rc = cam_layer(e, l, cat, ld, R, T)) != WS_OK) return rc;
    const CamTransit& t = e->cam_transit[bi];
    if ((rc = cam_bn_act(e, t.bn, cat, M, true, scratch, tin)) != WS_OK) return rc;
    if ((rc = cam_lin(e, tin, ld, M, static_cast<int>(ld), t.w, t.cout, nullptr, tout)) != WS_OK) return rc;
    y = tout, ch = t.cout;
  }
  float* yo = a.alloc(size_t(M) * ch);
  float* stats = a.alloc(size_t(R) * 2 * ch);
  float* raw = a.alloc(size_t(R) * e->E);
  float* u2 = a.alloc(size_t(R) * e->E);
  WS_PTR(yo && stats && raw && u2);
  if ((rc = cam_bn_act(e, e->cam_out_bn, y, M, true, scratch, yo)) != WS_OK) return rc;
  WS_RUN(e, ws_tstp_fwd(yo, R, 1, T, ch, kTstpEps, stats, s));
  if ((rc = cam_lin(e, stats, 2 * ch, R, 2 * ch, e->dev("spk_model.xvector.dense.linear.weight"), e->E, nullptr, raw)) != WS_OK)
    return rc;
  if ((rc = cam_bn_act(e, e->cam_dense_bn, raw, R, false, u2, emb)) != WS_OK) return rc;
  a.release(mk);
  return WS_OK;
}

// per-row CMN over the frames: feats [R][Te][nb] -= mean_t          (shared by both front-ends)
int subtract_time_mean(ws_engine* e, float* feats, int R, int Te, int nb) {
  const long long M = (long long)R * Te;
  void* s = e->stream;
  Arena& a = e->work;
  int nsplit = Te / 32;
  const int cap = 1024 / R > 1 ? 1024 / R : 1;
  if (nsplit > cap) nsplit = cap;
  if (nsplit < 1) nsplit = 1;
  float* slab = a.alloc(size_t(nsplit) * R * 2 * nb);
  float* sums = a.alloc(size_t(R) * 2 * nb);
  float* neg_mean = a.alloc(size_t(R) * nb);
  WS_PTR(slab && sums && neg_mean);
  WS_RUN(e, ws_chan_sums(feats, nullptr, nullptr, 1, Te, R, nsplit, nb, slab, s));
  WS_RUN(e, ws_reduce_slabs(slab, nsplit, (long long)R * 2 * nb, (long long)R * 2 * nb, sums, 0, 0, s));
  // rows of `sums` are [2][nb] per utterance: scale the first half of each by -1/Te into a dense [R][nb]
  for (int r = 0; r < R; ++r)
    WS_RUN(e, ws_affine_fwd(sums + size_t(r) * 2 * nb, nullptr, nullptr, -1.0f / Te, 1, 1, nb, neg_mean + size_t(r) * nb, s));
  WS_RUN(e, ws_affine_fwd(feats, nullptr, neg_mean, 1.0f, M, Te, nb, feats, s));
  return WS_OK;
}

// waveform [R][Tw] in [-1, 1] (device) -> mean-normalised kaldi fbank [R][Te][F]  (utils/funcs.py compute_fbank +
// apply_cmvn with dither 0; reference: SeparateEngine::ExtractFeature, separate_engine.cc:53-74)
int kaldi_fbank(ws_engine* e, const float* wav, int R, int Tw, float* feats, int Te) {
  const int win = e->fb_win, shift = e->fb_shift, nf = e->fb_padded / 2, nb = e->feat_dim;
  const long long M = (long long)R * Te;
  void* s = e->stream;
  Arena& a = e->work;
  const Arena::Mark mk = a.mark();
  float* spec = a.alloc(size_t(M) * 2 * nf);
  float* power = a.alloc(size_t(M) * nf);
  float* mel = a.alloc(size_t(M) * nb);
  WS_PTR(spec && power && mel);
  ws_gemm_nt_args g = {};
  g.A = wav, g.W = e->fb_basis, g.C = spec;
  g.a_div = Te, g.a_s1 = Tw, g.a_s2 = shift;            // frame f of row r starts at r*Tw + f*shift: overlapping view
  g.c_div = kBig, g.c_s2 = 2 * nf, g.st_div1 = 1, g.st_div2 = 1;
  g.M = static_cast<int>(M), g.N = 2 * nf, g.K = win, g.ldw = win;
  g.vec = (Tw % 4 == 0 && shift % 4 == 0 && win % 4 == 0) ? 3 : (win % 4 == 0 ? 2 : 0);     // exact-fp32 products
  WS_RUN(e, ws_gemm_nt(&g, s));
  WS_RUN(e, ws_power_spec(spec, M, nf, 2 * nf, nf, power, s));
  ws_gemm_nt_args h = {};
  h.A = power, h.W = e->fb_bank, h.C = mel;
  h.a_div = kBig, h.a_s2 = nf, h.c_div = kBig, h.c_s2 = nb, h.st_div1 = 1, h.st_div2 = 1;
  h.M = static_cast<int>(M), h.N = nb, h.K = nf, h.ldw = nf, h.vec = 3;
  WS_RUN(e, ws_gemm_nt(&h, s));
  // log(max(x, eps)) = log(relu(x - eps) + eps)
  WS_RUN(e, ws_prelu_fwd(mel, e->fb_floor, e->slope0, M, nb, static_cast<int>(M), feats, s));
  WS_RUN(e, ws_log_eps(feats, M * nb, kGnEps, s));
  {
    const int rc = subtract_time_mean(e, feats, R, Te, nb);
    if (rc != WS_OK) return rc;
  }
  a.release(mk);
  return WS_OK;
}

// waveform [R][Tw] (device) -> log-mel features [R][Te][F], mean-normalised over time: the in-model front-end of
// spk_feat = False models (bsrnn.py:343-350; modules/common/frontend.py fbank_frontend); Te = 1 + Tw / 128
int mel_frontend(ws_engine* e, const float* wav, int R, int Tw, float* feats, int Te) {
  const int n = 512, hop = kHop, pad = n / 2, nf = n / 2 + 1, nm = e->feat_dim;
  const int ldo = (Tw + 2 * pad + 3) / 4 * 4;
  const long long M = (long long)R * Te;
  void* s = e->stream;
  Arena& a = e->work;
  const Arena::Mark mk = a.mark();
  float* xp = a.alloc(size_t(R) * ldo);
  float* spec = a.alloc(size_t(M) * e->mel_lds);
  float* power = a.alloc(size_t(M) * e->mel_ldp);
  WS_PTR(xp && spec && power);
  int rc = zero_device(e, xp, size_t(R) * ldo * 4);
  if (rc != WS_OK) return rc;
  WS_RUN(e, ws_preemph_pad(wav, R, Tw, pad, ldo, e->mel_coef, xp, s));
  ws_gemm_nt_args g = {};
  g.A = xp, g.W = e->mel_basis, g.C = spec;
  g.a_div = Te, g.a_s1 = ldo, g.a_s2 = hop;             // centred frames as an overlapping row view
  g.c_div = kBig, g.c_s2 = e->mel_lds, g.st_div1 = 1, g.st_div2 = 1;
  g.M = static_cast<int>(M), g.N = e->mel_lds, g.K = n, g.ldw = n, g.vec = 3;       // exact-fp32 products
  WS_RUN(e, ws_gemm_nt(&g, s));
  WS_RUN(e, ws_power_spec(spec, M, nf, e->mel_lds, e->mel_ldp, power, s));
  ws_gemm_nt_args h = {};
  h.A = power, h.W = e->mel_fbt, h.C = feats;
  h.a_div = kBig, h.a_s2 = e->mel_ldp, h.c_div = kBig, h.c_s2 = nm, h.st_div1 = 1, h.st_div2 = 1;
  h.M = static_cast<int>(M), h.N = nm, h.K = e->mel_ldp, h.ldw = e->mel_ldp, h.vec = 3;
  WS_RUN(e, ws_gemm_nt(&h, s));
  WS_RUN(e, ws_log_eps(feats, M * nm, 1e-8f, s));
  if ((rc = subtract_time_mean(e, feats, R, Te, nm)) != WS_OK) return rc;
  a.release(mk);
  return WS_OK;
}

// BSRNN.forward (bsrnn.py:300-394) with the embedding already computed: wav [R][T], emb [R][E] -> est [R][T] (device)
int separate_device(ws_engine* e, const float* wav, int R, int T, const float* emb_in, float* est) {
  const int K = e->K, Tf = 1 + T / kHop, H1 = 4 * kN;
  const long long M = (long long)R * Tf;
  void* s = e->stream;
  Arena& a = e->work;
  int rc = build_descriptors(e, R, Tf);
  if (rc != WS_OK) return rc;
  ws_bands bands = {e->d_band_of_bin, e->d_f0, e->d_bw, K, kNBin};
  float* xbs = a.alloc(size_t(M) * 2 * kNBin);
  float* zA = a.alloc(size_t(R) * K * Tf * kN);
  float* zB = a.alloc(size_t(R) * K * Tf * kN);
  WS_PTR(xbs && zA && zB);
  // STFT + band split + per-band GroupNorm + Conv1d(k = 1)   (bsrnn.py:309-337)
  WS_RUN(e, ws_stft_bandsplit(wav, R, T, &bands, xbs, s));
  {
    const Arena::Mark mk = a.mark();
    float* stats = a.alloc(size_t(R) * K * 2);
    WS_PTR(stats);
    ws_groups_geom geo = {};
    geo.band_w = e->d_bw2, geo.band_off = e->d_off2;
    geo.gs1 = (long long)Tf * 2 * kNBin, geo.gs2 = 0, geo.rs = 2 * kNBin;
    geo.ngroups = R * K, geo.gdiv = K, geo.L = Tf, geo.W = 128, geo.nbands = K;
    WS_RUN(e, ws_group_stats(xbs, &geo, kGnEps, stats, s));
    ws_gemm_nt_args g = {};
    g.A = xbs, g.C = zA, g.stats = stats, g.groups = e->d_bn;
    g.a_div = kBig, g.a_s2 = 2 * kNBin;
    g.c_div = Tf, g.c_s1 = (long long)K * Tf * kN, g.c_s2 = kN;
    g.st_div1 = Tf, g.st_m1 = K, g.st_div2 = 1, g.st_m2 = 0;
    g.M = static_cast<int>(M), g.ngroups = K, g.max_n = kN, g.vec = 0 | 4;
    WS_RUN(e, ws_gemm_nt(&g, s));
    a.release(mk);
  }
  // speaker embedding -> (optional) SpeakerTransform (speaker.py:26-49)
  const float* emb = emb_in;
  if (e->use_xform) {
    const Tensor* t0 = e->find("spk_transform.transforms.0.weight");
    const int hid = static_cast<int>(t0->dims[0]);
    float* h0 = a.alloc(size_t(R) * hid);
    float* h1 = a.alloc(size_t(R) * hid);
    float* eo = a.alloc(size_t(R) * e->E);
    WS_PTR(h0 && h1 && eo);
    if ((rc = linear(e, emb, R, e->E, e->dev("spk_transform.transforms.0.weight"), e->E, hid,
                     e->dev("spk_transform.transforms.0.bias"), 0, h0)) != WS_OK ||
        (rc = linear(e, h0, R, hid, e->dev("spk_transform.transforms.1.weight"), hid, hid,
                     e->dev("spk_transform.transforms.1.bias"), 1, h1)) != WS_OK ||
        (rc = linear(e, h1, R, hid, e->dev("spk_transform.transforms.3.weight"), hid, e->E,
                     e->dev("spk_transform.transforms.3.bias"), 0, eo)) != WS_OK)
      return rc;
    emb = eo;
  }
  // separator (bsrnn.py:86-148)
  float* z = zA;
  float* other = zB;
  size_t net = 0;
  for (size_t i = 0; i < e->sep_kind.size(); ++i) {
    if (e->sep_kind[i] == 0) {
      if ((rc = fuse_layer(e, "separator.separation." + std::to_string(i) + ".", z, emb, R, Tf)) != WS_OK) return rc;
    } else {
      if ((rc = resrnn(e, e->rnn[2 * net], true, z, R, Tf, other)) != WS_OK) return rc;
      if ((rc = resrnn(e, e->rnn[2 * net + 1], false, other, R, Tf, z)) != WS_OK) return rc;
      ++net;
    }
  }
  // mask MLP + GLU complex mask + iSTFT (bsrnn.py:366-392)
  {
    const Arena::Mark mk = a.mark();
    float* stats = a.alloc(size_t(R) * K * 2);
    float* h1 = a.alloc(size_t(K) * M * H1);
    float* h2 = a.alloc(size_t(K) * M * H1);
    float* m3 = a.alloc(size_t(M) * 4 * kNBin);
    float* frames = a.alloc(size_t(M) * 512);
    WS_PTR(stats && h1 && h2 && m3 && frames);
    ws_groups_geom geo = {};
    geo.gs1 = (long long)Tf * kN, geo.gs2 = 0, geo.rs = kN;
    geo.ngroups = R * K, geo.gdiv = 1, geo.L = Tf, geo.W = kN, geo.nbands = K;
    WS_RUN(e, ws_group_stats(z, &geo, kGnEps, stats, s));
    int maxbw = 0;
    for (int b : e->bw) maxbw = b > maxbw ? b : maxbw;
    ws_gemm_nt_args g = {};
    g.A = z, g.C = h1, g.stats = stats, g.groups = e->d_l1;
    g.a_div = Tf, g.a_s1 = (long long)K * Tf * kN, g.a_s2 = kN;
    g.c_div = kBig, g.c_s2 = H1;
    g.st_div1 = Tf, g.st_m1 = K, g.st_div2 = 1, g.st_m2 = 0;
    g.M = static_cast<int>(M), g.act = 1, g.ngroups = K, g.max_n = H1, g.vec = 3 | 4;
    WS_RUN(e, ws_gemm_nt(&g, s));
    ws_gemm_nt_args g2 = {};
    g2.A = h1, g2.C = h2, g2.groups = e->d_l2;
    g2.a_div = kBig, g2.a_s2 = H1, g2.c_div = kBig, g2.c_s2 = H1, g2.st_div1 = 1, g2.st_div2 = 1;
    g2.M = static_cast<int>(M), g2.act = 1, g2.ngroups = K, g2.max_n = H1, g2.vec = 3 | 4;
    WS_RUN(e, ws_gemm_nt(&g2, s));
    ws_gemm_nt_args g3 = {};
    g3.A = h2, g3.C = m3, g3.groups = e->d_l3;
    g3.a_div = kBig, g3.a_s2 = H1, g3.c_div = kBig, g3.c_s2 = 4 * kNBin, g3.st_div1 = 1, g3.st_div2 = 1;
    g3.M = static_cast<int>(M), g3.ngroups = K, g3.max_n = 4 * maxbw, g3.vec = 3 | 4;
    WS_RUN(e, ws_gemm_nt(&g3, s));
    WS_RUN(e, ws_mask_istft_frames(xbs, m3, R, Tf, &bands, frames, s));
    WS_RUN(e, ws_istft_ola(frames, R, Tf, T, est, s));
    a.release(mk);
  }
  return WS_OK;
}


// =================================================================================================================
// Conv-TasNet / SpEx+ (arch 1): the launch plan of wesep_amd/functional_tasnet.py in eval mode for the shipped
// configuration -- MultiEncoder / MultiDecoder, gLN, non-causal, no skip connection, concatConv multi-fusion
// (convtasnet.py:162-219, separation.py:57-186, convs.py:41-160, encoder.py:66-114, decoder.py:66-114), fixed embeddings
// or the SpEx+ speaker encoder on the enrollment waveform through the shared encoder (tasnet/speaker.py:47-64).
// Only the first of the three decoder branches is computed: it is the estimate the reference's inference writes.
// =================================================================================================================
constexpr float kLnEps = 1e-5f;

struct TasGemm {
  const float* A = nullptr;
  long long lda = 0;
  long long M = 0;
  int K = 0;
  const float* W = nullptr;
  int ldw = 0, N = 0;
  const float* bias = nullptr;
  int act = 0;
  float* C = nullptr;
  long long ldc = 0;
  const float* R = nullptr;                                       // residual, addressed like C
  const float *stats = nullptr, *gamma = nullptr, *beta = nullptr; // norm-on-load
  int st_div1 = 1;                                                 // rows per statistics pair (gLN: T', cLN: 1)
  bool f32 = false;                                                // exact-fp32 products (the SpEx+ speaker encoder)
  int a_div = kBig;                                                // frames view: row m -> (m / a_div) * a_s1 + (m % a_div) * lda
  long long a_s1 = 0;
};

int tas_gemm(ws_engine* e, const TasGemm& t) {
  ws_gemm_nt_args g = {};
  g.A = t.A, g.W = t.W, g.bias = t.bias, g.C = t.C, g.R = t.R;
  g.stats = t.stats, g.gamma = t.gamma, g.beta = t.beta;
  g.a_div = t.a_div, g.a_s1 = t.a_s1, g.a_s2 = t.lda;
  g.c_div = kBig, g.c_s2 = t.ldc;
  g.st_div1 = t.st_div1, g.st_m1 = 1, g.st_div2 = 1, g.st_m2 = 0;
  g.M = static_cast<int>(t.M), g.N = t.N, g.K = t.K, g.ldw = t.ldw, g.act = t.act;
  int vec = 0;
  if (t.a_div == kBig) {
    vec = (t.K % 4 == 0 && t.ldw % 4 == 0 && t.lda % 4 == 0) ? 3 : 0;
  } else {
    vec = (t.K % 4 == 0 && t.ldw % 4 == 0) ? 2 : 0;   // overlapping frames: only the weight rows are 16-byte loadable
  }
  g.vec = vec | (t.f32 ? 0 : 4);
  WS_RUN(e, ws_gemm_nt(&g, e->stream));
  return WS_OK;
}

int tas_row_stats(ws_engine* e, const float* x, long long M, int C, float* st) {   // cLN statistics per frame
  ws_groups_geom geo = {};
  geo.gs1 = C, geo.gs2 = 0, geo.rs = C, geo.ngroups = static_cast<int>(M), geo.gdiv = 1, geo.L = 1, geo.W = C, geo.nbands = 1;
  WS_RUN(e, ws_group_stats(x, &geo, kLnEps, st, e->stream));
  return WS_OK;
}

int tas_flat_stats(ws_engine* e, const float* x, int R, long long n, float* st) {  // gLN statistics per utterance
  int nchunk = static_cast<int>(n / 16384);
  const int cap = 512 / R > 1 ? 512 / R : 1;
  if (nchunk > cap) nchunk = cap;
  if (nchunk < 1) nchunk = 1;
  float* scratch = e->work.alloc(size_t(R) * nchunk * 4);
  WS_PTR(scratch);
  WS_RUN(e, ws_flat_stats(x, R, n, kLnEps, nchunk, scratch, st, e->stream));
  return WS_OK;
}

// MultiEncoder (encoder.py:66-114): wav [R][T] -> cat [M][3N] (ReLU outputs of the three filterbanks) and, if wanted,
// e [M][B] = proj(LayerNorm(cat)); M = R * T', T' = (T - L) / stride + 1
int tas_encode(ws_engine* e, const float* wav, int R, int T, float* cat, float* feat) {
  const int N = e->tN, L = e->tL, B = e->tB, stride = L / 2;
  const int Ls[3] = {L, 80, 160};
  const char* names[3] = {"encoder.encoder_1d_short.", "encoder.encoder_1d_middle.", "encoder.encoder_1d_long."};
  const int Tp = (T - L) / stride + 1;
  const long long M = (long long)R * Tp;
  int Tpad = (Tp - 1) * stride + Ls[2];
  if (Tpad < T) Tpad = T;
  Tpad = (Tpad + 3) / 4 * 4;
  Arena& a = e->work;
  const Arena::Mark mk = a.mark();
  float* xp = a.alloc(size_t(R) * Tpad);
  WS_PTR(xp);
  int rc = zero_device(e, xp, size_t(R) * Tpad * 4);
  if (rc != WS_OK) return rc;
  if ((rc = copy_cols(e, xp, Tpad, wav, T, T, R)) != WS_OK) return rc;
  for (int i = 0; i < 3; ++i) {
    TasGemm g;
    g.A = xp, g.a_div = Tp, g.a_s1 = Tpad, g.lda = stride, g.M = M, g.K = Ls[i];
    g.W = e->dev(std::string(names[i]) + "weight"), g.ldw = Ls[i], g.N = N, g.bias = e->dev(std::string(names[i]) + "bias");
    g.act = 2, g.C = cat + (long long)i * N, g.ldc = 3 * N;
    if ((rc = tas_gemm(e, g)) != WS_OK) return rc;
  }
  if (feat) {
    float* st = a.alloc(size_t(M) * 2);
    WS_PTR(st);
    if ((rc = tas_row_stats(e, cat, M, 3 * N, st)) != WS_OK) return rc;
    TasGemm g;
    g.A = cat, g.lda = 3 * N, g.M = M, g.K = 3 * N, g.W = e->dev("encoder.proj.weight"), g.ldw = 3 * N, g.N = B;
    g.bias = e->dev("encoder.proj.bias"), g.C = feat, g.ldc = B;
    g.stats = st, g.gamma = e->dev("encoder.ln.weight"), g.beta = e->dev("encoder.ln.bias"), g.st_div1 = 1;
    if ((rc = tas_gemm(e, g)) != WS_OK) return rc;
  }
  a.release(mk);
  return WS_OK;
}

// one TCN block (convs.py:41-160): out = x + sconv(gLN2(prelu2(dconv(gLN1(prelu1(conv1x1(x) + rb))))))
int tas_block(ws_engine* e, const std::string& pre, bool fuse, int dil, const float* x, const float* rb, int R, int Tp,
              float* out) {
  const int B = e->tB, H = e->tH, P = e->tP;
  const long long M = (long long)R * Tp;
  void* s = e->stream;
  Arena& a = e->work;
  const Arena::Mark mk = a.mark();
  const char* n_p1 = fuse ? "prelu1.weight" : "PReLU_1.weight";
  const char* n_p2 = fuse ? "prelu2.weight" : "PReLU_2.weight";
  const std::string n1 = pre + (fuse ? "lnorm1." : "norm_1."), n2 = pre + (fuse ? "lnorm2." : "norm_2.");
  const std::string dw = pre + (fuse ? "dconv." : "dwconv."), outc = pre + (fuse ? "sconv." : "Output.");
  float* c = a.alloc(size_t(M) * H);
  float* y1 = a.alloc(size_t(M) * H);
  float* z = a.alloc(size_t(M) * H);
  float* st1 = a.alloc(size_t(R) * 2);
  float* st2 = a.alloc(size_t(R) * 2);
  WS_PTR(c && y1 && z && st1 && st2);
  TasGemm g;
  g.A = x, g.lda = B, g.M = M, g.K = B, g.W = e->dev(pre + "conv1x1.weight"), g.ldw = fuse ? B + e->E : B, g.N = H;
  g.bias = rb ? nullptr : e->dev(pre + "conv1x1.bias"), g.C = c, g.ldc = H;
  int rc = tas_gemm(e, g);
  if (rc != WS_OK) return rc;
  WS_RUN(e, ws_prelu_fwd(c, rb, e->dev(pre + n_p1), M, H, Tp, y1, s));
  if ((rc = tas_flat_stats(e, y1, R, (long long)Tp * H, st1)) != WS_OK) return rc;
  WS_RUN(e, ws_dwconv_ex_fwd(y1, st1, e->dev(n1 + "weight"), e->dev(n1 + "bias"), e->dev(dw + "weight"), e->dev(dw + "bias"), R,
                             Tp, H, P, dil, Tp, 0, z, s));
  WS_RUN(e, ws_prelu_fwd(z, nullptr, e->dev(pre + n_p2), M, H, Tp, c, s));      // y2 -> c (its contents are dead)
  if ((rc = tas_flat_stats(e, c, R, (long long)Tp * H, st2)) != WS_OK) return rc;
  TasGemm o;
  o.A = c, o.lda = H, o.M = M, o.K = H, o.W = e->dev(outc + "weight"), o.ldw = H, o.N = B, o.bias = e->dev(outc + "bias");
  o.C = out, o.ldc = B, o.R = x, o.stats = st2, o.gamma = e->dev(n2 + "weight"), o.beta = e->dev(n2 + "bias"), o.st_div1 = Tp;
  if ((rc = tas_gemm(e, o)) != WS_OK) return rc;
  a.release(mk);
  return WS_OK;
}

// SpEx+ speaker encoder (tasnet/speaker.py:7-64) in eval mode: cat_aux [R*T0][3N] -> emb [R][E]
int tas_spk_embed(ws_engine* e, const float* cat, int R, int T0, float* emb) {
  const int C0 = 3 * e->tN;
  void* s = e->stream;
  Arena& a = e->work;
  const Arena::Mark mk = a.mark();
  const std::string sp = "spk_model.aux_enc3.";
  long long M = (long long)R * T0;
  float* st0 = a.alloc(size_t(M) * 2);
  float* x = a.alloc(size_t(M) * 256);
  WS_PTR(st0 && x);
  int rc = tas_row_stats(e, cat, M, C0, st0);
  if (rc != WS_OK) return rc;
  {
    TasGemm g;
    g.A = cat, g.lda = C0, g.M = M, g.K = C0, g.W = e->dev(sp + "1.weight"), g.ldw = C0, g.N = 256, g.bias = e->dev(sp + "1.bias");
    g.C = x, g.ldc = 256, g.stats = st0, g.gamma = e->dev(sp + "0.weight"), g.beta = e->dev(sp + "0.bias");
    if ((rc = tas_gemm(e, g)) != WS_OK) return rc;
  }
  int T = T0, ci = 256;
  const int cos[3] = {256, 512, 512};
  for (int i = 0; i < 3; ++i) {
    const std::string bp = sp + std::to_string(2 + i) + ".";
    const int co = cos[i];
    M = (long long)R * T;
    float* c1 = a.alloc(size_t(M) * co);
    float* u = a.alloc(size_t(M) * co);
    float* y1 = a.alloc(size_t(M) * co);
    float* c2 = a.alloc(size_t(M) * co);
    float* res = ci != co ? a.alloc(size_t(M) * co) : nullptr;
    float* y2 = a.alloc(size_t(M) * co);
    float* pooled = a.alloc(size_t(R) * (T / 3) * co);
    WS_PTR(c1 && u && y1 && c2 && y2 && pooled && (ci == co || res));
    TasGemm g1;
    g1.A = x, g1.lda = ci, g1.M = M, g1.K = ci, g1.W = e->dev(bp + "conv1.weight"), g1.ldw = ci, g1.N = co, g1.C = c1, g1.ldc = co;
    g1.f32 = true;
    if ((rc = tas_gemm(e, g1)) != WS_OK) return rc;
    WS_RUN(e, ws_bn_prelu_fwd(c1, e->tas_bn_st[i][0], e->dev(bp + "batch_norm1.weight"), e->dev(bp + "batch_norm1.bias"), nullptr,
                              e->dev(bp + "prelu1.weight"), M, co, u, y1, s));
    TasGemm g2;
    g2.A = y1, g2.lda = co, g2.M = M, g2.K = co, g2.W = e->dev(bp + "conv2.weight"), g2.ldw = co, g2.N = co, g2.C = c2, g2.ldc = co;
    g2.f32 = true;
    if ((rc = tas_gemm(e, g2)) != WS_OK) return rc;
    const float* resp = x;
    if (ci != co) {
      TasGemm gd;
      gd.A = x, gd.lda = ci, gd.M = M, gd.K = ci, gd.W = e->dev(bp + "conv_downsample.weight"), gd.ldw = ci, gd.N = co;
      gd.C = res, gd.ldc = co, gd.f32 = true;
      if ((rc = tas_gemm(e, gd)) != WS_OK) return rc;
      resp = res;
    }
    WS_RUN(e, ws_bn_prelu_fwd(c2, e->tas_bn_st[i][1], e->dev(bp + "batch_norm2.weight"), e->dev(bp + "batch_norm2.bias"), resp,
                              e->dev(bp + "prelu2.weight"), M, co, u, y2, s));
    WS_RUN(e, ws_maxpool3_fwd(y2, R, T, co, pooled, s));
    x = pooled, T = T / 3, ci = co;
  }
  float* mean2 = a.alloc(size_t(R) * 2 * ci);
  WS_PTR(mean2);
  if ((rc = time_mean(e, x, R, T, ci, mean2)) != WS_OK) return rc;
  TasGemm g5;
  g5.A = mean2, g5.lda = 2 * ci, g5.M = R, g5.K = ci, g5.W = e->dev(sp + "5.weight"), g5.ldw = ci, g5.N = e->E;
  g5.bias = e->dev(sp + "5.bias"), g5.C = emb, g5.ldc = e->E, g5.f32 = true;
  if ((rc = tas_gemm(e, g5)) != WS_OK) return rc;
  a.release(mk);
  return WS_OK;
}

int spk_transform(ws_engine* e, const float* emb, int R, const float** out) {
  *out = emb;
  if (!e->use_xform) return WS_OK;
  Arena& a = e->work;
  const Tensor* t0 = e->find("spk_transform.transforms.0.weight");
  const int hid = static_cast<int>(t0->dims[0]);
  float* h0 = a.alloc(size_t(R) * hid);
  float* h1 = a.alloc(size_t(R) * hid);
  float* eo = a.alloc(size_t(R) * e->E);
  WS_PTR(h0 && h1 && eo);
  int rc;
  if ((rc = linear(e, emb, R, e->E, e->dev("spk_transform.transforms.0.weight"), e->E, hid,
                   e->dev("spk_transform.transforms.0.bias"), 0, h0)) != WS_OK ||
      (rc = linear(e, h0, R, hid, e->dev("spk_transform.transforms.1.weight"), hid, hid,
                   e->dev("spk_transform.transforms.1.bias"), 1, h1)) != WS_OK ||
      (rc = linear(e, h1, R, hid, e->dev("spk_transform.transforms.3.weight"), hid, e->E,
                   e->dev("spk_transform.transforms.3.bias"), 0, eo)) != WS_OK)
    return rc;
  *out = eo;
  return WS_OK;
}

// wav [R][T], emb_in [R][E] (or NULL with enroll_wave [R][Te] for the SpEx+ encoder) -> est [R][T]: the first
// (T' - 1) * stride + L samples of each row are the model's output, the rest zeros
int tasnet_device(ws_engine* e, const float* wav, int R, int T, const float* emb_in, const float* enroll_wave, int Te,
                  float* est) {
  const int N = e->tN, L = e->tL, B = e->tB, H = e->tH, stride = L / 2;
  const int Tp = (T - L) / stride + 1;
  const long long M = (long long)R * Tp;
  void* s = e->stream;
  Arena& a = e->work;
  float* cat = a.alloc(size_t(M) * 3 * N);
  float* zA = a.alloc(size_t(M) * B);
  float* zB = a.alloc(size_t(M) * B);
  float* emb_own = a.alloc(size_t(R) * e->E);
  WS_PTR(cat && zA && zB && emb_own);
  int rc = tas_encode(e, wav, R, T, cat, zA);
  if (rc != WS_OK) return rc;
  const float* emb = emb_in;
  if (!emb) {                                  // enrollment waveform through the SHARED encoder (convtasnet.py:179-187)
    const Arena::Mark mk = a.mark();
    const int Tpa = (Te - L) / stride + 1;
    float* cat_aux = a.alloc(size_t(R) * Tpa * 3 * N);
    WS_PTR(cat_aux);
    if ((rc = tas_encode(e, enroll_wave, R, Te, cat_aux, nullptr)) != WS_OK) return rc;
    if ((rc = tas_spk_embed(e, cat_aux, R, Tpa, emb_own)) != WS_OK) return rc;
    a.release(mk);
    emb = emb_own;
  }
  if ((rc = spk_transform(e, emb, R, &emb)) != WS_OK) return rc;
  float* x = zA;
  float* other = zB;
  float* rb = a.alloc(size_t(R) * H);
  WS_PTR(rb);
  for (int r = 0; r < e->tR; ++r) {
    const std::string fp = "separation.separation." + std::to_string(2 * r) + ".";
    // conv1x1(cat[x, e]) = W_x x + (W_e e + b): the embedding part is one [R][H] GEMM (convs.py:143-148)
    if ((rc = linear(e, emb, R, e->E, e->dev(fp + "conv1x1.weight") + B, B + e->E, H, e->dev(fp + "conv1x1.bias"), 0, rb)) !=
        WS_OK)
      return rc;
    if ((rc = tas_block(e, fp, true, 1, x, rb, R, Tp, other)) != WS_OK) return rc;
    std::swap(x, other);
    for (int k = 1; k < e->tX; ++k) {
      const std::string bp = "separation.separation." + std::to_string(2 * r + 1) + ".separation." + std::to_string(k - 1) + ".";
      if ((rc = tas_block(e, bp, false, 1 << k, x, nullptr, R, Tp, other)) != WS_OK) return rc;
      std::swap(x, other);
    }
  }
  // MultiDecoder, first branch (decoder.py:66-114): ReLU mask, mask * w1, transposed convolution as GEMM + overlap-add
  {
    const int xlen = (Tp - 1) * stride + L;
    float* m = a.alloc(size_t(M) * N);
    float* sm = a.alloc(size_t(M) * N);
    float* fr = a.alloc(size_t(M) * L);
    float* out = a.alloc(size_t(R) * xlen);
    WS_PTR(m && sm && fr && out);
    TasGemm g;
    g.A = x, g.lda = B, g.M = M, g.K = B, g.W = e->dev("decoder.mask1.weight"), g.ldw = B, g.N = N;
    g.bias = e->dev("decoder.mask1.bias"), g.act = 2, g.C = m, g.ldc = N;
    if ((rc = tas_gemm(e, g)) != WS_OK) return rc;
    WS_RUN(e, ws_maskmul_fwd(cat, 3 * N, m, M, N, sm, s));
    TasGemm d;
    d.A = sm, d.lda = N, d.M = M, d.K = N, d.W = e->tas_dec_wt, d.ldw = N, d.N = L, d.C = fr, d.ldc = L;
    if ((rc = tas_gemm(e, d)) != WS_OK) return rc;
    WS_RUN(e, ws_ola_fwd(fr, e->dev("decoder.decoder_1d_1.bias"), R, Tp, L, stride, xlen, out, s));
    if ((rc = zero_device(e, est, size_t(R) * T * 4)) != WS_OK) return rc;
    if ((rc = copy_cols(e, est, T, out, xlen, xlen, R)) != WS_OK) return rc;
  }
  return WS_OK;
}

int prepare_tasnet(ws_engine* e) {
  e->sr = static_cast<int>(meta_or(e, "sample_rate", 16000));
  e->E = static_cast<int>(meta_or(e, "spk_emb_dim", 256));
  e->use_xform = static_cast<int>(meta_or(e, "use_spk_transform", 0));
  e->joint = static_cast<int>(meta_or(e, "joint_training", 0));
  e->spk_feat = 0;               // a joint Conv-TasNet takes the enrollment WAVEFORM (shared encoder)
  e->tN = static_cast<int>(meta_or(e, "N", 512)), e->tL = static_cast<int>(meta_or(e, "L", 16));
  e->tB = static_cast<int>(meta_or(e, "B", 128)), e->tH = static_cast<int>(meta_or(e, "H", 512));
  e->tP = static_cast<int>(meta_or(e, "P", 3)), e->tX = static_cast<int>(meta_or(e, "X", 8));
  e->tR = static_cast<int>(meta_or(e, "R", 3));
  const int N = e->tN, L = e->tL, B = e->tB, H = e->tH, P = e->tP, E = e->E;
  if (N % 4 || B % 4 || H % 4 || E % 4 || L % 2 || L < 4 || L > 80 || P < 1 || P > 7 || e->tX < 1 || e->tR < 1) {
    set_err("engine: unsupported Conv-TasNet geometry (N %d, L %d, B %d, H %d, P %d, X %d, R %d, E %d)", N, L, B, H, P, e->tX,
            e->tR, E);
    return WS_ERR_INVALID;
  }
  if (e->joint && N != 256) {
    set_err("engine: the SpEx+ speaker encoder is hard-wired to 3 x 256 encoder channels (tasnet/speaker.py:52-53); N = %d", N);
    return WS_ERR_INVALID;
  }
  e->dw = e->persist.alloc(e->hw.size());
  WS_PTR(e->dw);
  int rc = to_device(e, e->dw, e->hw.data(), e->hw.size() * 4);
  if (rc != WS_OK) return rc;
  const int Ls[3] = {L, 80, 160};
  const char* enc[3] = {"encoder.encoder_1d_short.", "encoder.encoder_1d_middle.", "encoder.encoder_1d_long."};
  for (int i = 0; i < 3; ++i)
    if (!require(e, std::string(enc[i]) + "weight", {N, 1, Ls[i]}) || !require(e, std::string(enc[i]) + "bias", {N}))
      return WS_ERR_INVALID;
  if (!require(e, "encoder.ln.weight", {3 * N}) || !require(e, "encoder.ln.bias", {3 * N}) ||
      !require(e, "encoder.proj.weight", {B, 3 * N, 1}) || !require(e, "encoder.proj.bias", {B}) ||
      !require(e, "decoder.mask1.weight", {N, B, 1}) || !require(e, "decoder.mask1.bias", {N}) ||
      !require(e, "decoder.decoder_1d_1.weight", {N, 1, L}) || !require(e, "decoder.decoder_1d_1.bias", {1}))
    return WS_ERR_INVALID;
  for (int r = 0; r < e->tR; ++r) {
    const std::string fp = "separation.separation." + std::to_string(2 * r) + ".";
    if (!require(e, fp + "conv1x1.weight", {H, B + E, 1}) || !require(e, fp + "conv1x1.bias", {H}) ||
        !require(e, fp + "prelu1.weight", {1}) || !require(e, fp + "lnorm1.weight", {H, 1}) ||
        !require(e, fp + "lnorm1.bias", {H, 1}) || !require(e, fp + "dconv.weight", {H, 1, P}) ||
        !require(e, fp + "dconv.bias", {H}) || !require(e, fp + "prelu2.weight", {1}) ||
        !require(e, fp + "lnorm2.weight", {H, 1}) || !require(e, fp + "lnorm2.bias", {H, 1}) ||
        !require(e, fp + "sconv.weight", {B, H, 1}) || !require(e, fp + "sconv.bias", {B}))
      return WS_ERR_INVALID;
    for (int k = 1; k < e->tX; ++k) {
      const std::string bp = "separation.separation." + std::to_string(2 * r + 1) + ".separation." + std::to_string(k - 1) + ".";
      if (!require(e, bp + "conv1x1.weight", {H, B, 1}) || !require(e, bp + "conv1x1.bias", {H}) ||
          !require(e, bp + "PReLU_1.weight", {1}) || !require(e, bp + "norm_1.weight", {H, 1}) ||
          !require(e, bp + "norm_1.bias", {H, 1}) || !require(e, bp + "dwconv.weight", {H, 1, P}) ||
          !require(e, bp + "dwconv.bias", {H}) || !require(e, bp + "PReLU_2.weight", {1}) ||
          !require(e, bp + "norm_2.weight", {H, 1}) || !require(e, bp + "norm_2.bias", {H, 1}) ||
          !require(e, bp + "Output.weight", {B, H, 1}) || !require(e, bp + "Output.bias", {B}))
        return WS_ERR_INVALID;
    }
  }
  if (e->use_xform) {
    const Tensor* t0 = e->find("spk_transform.transforms.0.weight");
    if (!t0 || t0->dims.size() < 2 || t0->dims[1] != E || !e->find("spk_transform.transforms.1.weight") ||
        !e->find("spk_transform.transforms.3.weight")) {
      set_err("engine: spk_transform tensors missing or mis-shaped");
      return WS_ERR_INVALID;
    }
  }
  // synthesis filterbank as a GEMM operand: [N][L] -> [L][N]
  e->tas_dec_wt = e->persist.alloc(size_t(L) * N);
  WS_PTR(e->tas_dec_wt);
  WS_RUN(e, ws_transpose(e->dev("decoder.decoder_1d_1.weight"), N, L, L, e->tas_dec_wt, e->stream));
  if (e->joint) {
    const std::string sp = "spk_model.aux_enc3.";
    if (!require(e, sp + "0.weight", {3 * N}) || !require(e, sp + "0.bias", {3 * N}) ||
        !require(e, sp + "1.weight", {256, 3 * N, 1}) || !require(e, sp + "1.bias", {256}) ||
        !require(e, sp + "5.weight", {E, 512, 1}) || !require(e, sp + "5.bias", {E}))
      return WS_ERR_INVALID;
    int ci = 256;
    const int cos[3] = {256, 512, 512};
    for (int i = 0; i < 3; ++i) {
      const std::string bp = sp + std::to_string(2 + i) + ".";
      const int co = cos[i];
      if (!require(e, bp + "conv1.weight", {co, ci, 1}) || !require(e, bp + "conv2.weight", {co, co, 1}) ||
          !require(e, bp + "prelu1.weight", {1}) || !require(e, bp + "prelu2.weight", {1}) ||
          (ci != co && !require(e, bp + "conv_downsample.weight", {co, ci, 1})))
        return WS_ERR_INVALID;
      for (int j = 0; j < 2; ++j) {
        const std::string bn = bp + "batch_norm" + std::to_string(j + 1) + ".";
        if (!require(e, bn + "weight", {co}) || !require(e, bn + "bias", {co}) || !require(e, bn + "running_mean", {co}) ||
            !require(e, bn + "running_var", {co}))
          return WS_ERR_INVALID;
        std::vector<float> st(size_t(2) * co);
        const float *rm = e->host(bn + "running_mean"), *rv = e->host(bn + "running_var");
        for (int o = 0; o < co; ++o) {
          st[o] = rm[o];
          st[co + o] = 1.0f / sqrtf(rv[o] + kBnEps);
        }
        e->tas_bn_st[i][j] = upload(e, e->persist, st.data(), st.size());
        WS_PTR(e->tas_bn_st[i][j]);
      }
      ci = co;
    }
  }
  if (!e->dry && hipStreamSynchronize(e->stream) != hipSuccess) {
    set_err("engine: weight preparation failed on the device");
    return WS_ERR_LAUNCH;
  }
  return WS_OK;
}

// =================================================================================================================
// DPCCN (arch 2): the launch plan of wesep_amd/models/dpccn.py (wesep/models/dpccn.py:206-290) in eval mode -- STFT as a
// DFT-basis GEMM, Conv2d(2 -> 16), the dense blocks through the halo-tile convolution (ws_conv3x3), (1, 2)-strided
// convolutions and transposed convolutions as implicit GEMMs, ELU + InstanceNorm fused (ws_in_act_*), the TCN stack
// (IN - ELU - depthwise dilated conv - IN - ELU - 1x1 conv + residual), the four pooling branches (AvgPool2d, 1x1
// conv, bilinear upsampling), ConvTranspose2d(32 -> 2) and the inverse STFT.  Channels-last [R * T * F][C] everywhere,
// H = frames, W = bins.  multiply / additive / FiLM fusion; fixed embeddings or the speaker encoders of the pBSRNN
// plan (fbank / waveform enrollment).  InstanceNorm2d / InstanceNorm1d carry no running statistics in the reference
// (affine = False, track_running_stats = False): eval and training forward are the same computation.
// =================================================================================================================
constexpr int kDpWin = 512, kDpBins = 257, kDpLd = 4 * kDpBins;     // 1028: (re, im, 0, 0) per bin
constexpr float kInEps = 1e-5f;
constexpr int kInPre = 1, kInPost = 2;                               // ws_in_act_* flags: IN(ELU(x)) / ELU(IN(x))

uint16_t bf16_rne(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return static_cast<uint16_t>((u >> 16) | 0x40);   // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return static_cast<uint16_t>(u >> 16);
}

float bf16_float(uint16_t h) {
  const uint32_t u = uint32_t(h) << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

// wesep_amd.dev.conv3x3_pack on the host: W2 [Cout][9 * Cin] (tap-major rows) -> the bf16 hi / lo MFMA-fragment order of
// ws_conv3x3 (include/wesep_hip.h): unit (((chunk*9 + tap)*NTP + t)*2 + part)*64 + lane, 8 bf16 each
std::vector<float> dp_pack3x3(const std::vector<float>& W2, int Cin, int Cout) {
  const int ntt = (Cout + 31) / 32, ntp = ntt <= 2 ? ntt : ntt + (ntt & 1), nch = (Cin + 15) / 16;
  std::vector<uint16_t> out(size_t(nch) * 9 * ntp * 2 * 64 * 8, 0);
  for (int chunk = 0; chunk < nch; ++chunk)
    for (int tap = 0; tap < 9; ++tap)
      for (int t = 0; t < ntp; ++t)
        for (int lane = 0; lane < 64; ++lane) {
          const int n = t * 32 + (lane & 31);
          for (int j = 0; j < 8; ++j) {
            const int c = chunk * 16 + 8 * (lane >> 5) + j;
            const float v = (n < Cout && c < Cin) ? W2[size_t(n) * 9 * Cin + size_t(tap) * Cin + c] : 0.f;
            const uint16_t hi = bf16_rne(v), lo = bf16_rne(v - bf16_float(hi));
            const size_t u = ((size_t(chunk) * 9 + tap) * ntp + t) * 2;
            out[((u + 0) * 64 + lane) * 8 + j] = hi;
            out[((u + 1) * 64 + lane) * 8 + j] = lo;
          }
        }
  std::vector<float> f(out.size() / 2);
  memcpy(f.data(), out.data(), out.size() * 2);
  return f;
}

// w [Cout][Cin][3][3] (Conv2d) -> W2 [Cout][(ky*3 + kx)*cin_pad + ci], input channels zero-padded to cin_pad
std::vector<float> dp_conv_rows(const float* w, int Cout, int Cin, int cin_pad) {
  std::vector<float> W2(size_t(Cout) * 9 * cin_pad, 0.f);
  for (int o = 0; o < Cout; ++o)
    for (int c = 0; c < Cin; ++c)
      for (int t = 0; t < 9; ++t) W2[size_t(o) * 9 * cin_pad + size_t(t) * cin_pad + c] = w[(size_t(o) * Cin + c) * 9 + t];
  return W2;
}

// w [Cin][Cout][3][3] (ConvTranspose2d) -> Wt [cout_pad][(ky*3 + kx)*Cin + ci], output channels zero-padded
std::vector<float> dp_convT_rows(const float* w, int Cin, int Cout, int cout_pad) {
  std::vector<float> Wt(size_t(cout_pad) * 9 * Cin, 0.f);
  for (int c = 0; c < Cin; ++c)
    for (int o = 0; o < Cout; ++o)
      for (int t = 0; t < 9; ++t) Wt[size_t(o) * 9 * Cin + size_t(t) * Cin + c] = w[(size_t(c) * Cout + o) * 9 + t];
  return Wt;
}

struct DpDense {
  const char* prefix;
  int C0, g, co5;
};
const DpDense kDpDense[10] = {{"encoder.0.", 16, 16, 16},   {"encoder.1.1.", 32, 32, 32}, {"encoder.2.1.", 32, 32, 32},
                              {"encoder.3.1.", 32, 32, 32}, {"encoder.4.1.", 32, 32, 32}, {"decoder.3.0.", 64, 32, 64},
                              {"decoder.4.0.", 64, 32, 64}, {"decoder.5.0.", 64, 32, 64}, {"decoder.6.0.", 64, 32, 64},
                              {"decoder.7.", 32, 16, 32}};
struct DpConv {
  const char* prefix;
  int cin, cout;
};
const DpConv kDpEnc[7] = {{"encoder.1.0.", 16, 32}, {"encoder.2.0.", 32, 32}, {"encoder.3.0.", 32, 32}, {"encoder.4.0.", 32, 32},
                          {"encoder.5.", 32, 64},   {"encoder.6.", 64, 128},  {"encoder.7.", 128, 384}};
const DpConv kDpDec[7] = {{"decoder.0.", 768, 128}, {"decoder.1.", 256, 64},  {"decoder.2.", 128, 32}, {"decoder.3.1.", 64, 32},
                          {"decoder.4.1.", 64, 32}, {"decoder.5.1.", 64, 32}, {"decoder.6.1.", 64, 16}};

int dp_keep(ws_engine* e, const std::string& key, const std::vector<float>& host) {
  float* d = upload(e, e->persist, host.data(), host.size());
  WS_PTR(d);
  e->dp_w[key] = d;
  return WS_OK;
}

int prepare_dpccn(ws_engine* e) {
  e->sr = static_cast<int>(meta_or(e, "sample_rate", 16000));
  int rc = read_speaker_meta(e);
  if (rc != WS_OK) return rc;
  e->dp_fuse = static_cast<int>(meta_or(e, "spk_fuse_type", 2));
  e->dp_causal = static_cast<int>(meta_or(e, "causal", 0));
  e->dp_tcn_blocks = static_cast<int>(meta_or(e, "tcn_blocks", 10));
  e->dp_tcn_layers = static_cast<int>(meta_or(e, "tcn_layers", 2));
  if (meta_or(e, "win", 512) != kDpWin || meta_or(e, "stride", 128) != kHop || meta_or(e, "feature_dim", kDpBins) != kDpBins ||
      meta_or(e, "multi_fuse", 0) != 0) {
    set_err("engine: the DPCCN plan is built for win 512, stride 128, feature_dim 257, multi_fuse False");
    return WS_ERR_INVALID;
  }
  if (e->dp_fuse < 0 || e->dp_fuse > 3 || e->dp_tcn_blocks < 1 || e->dp_tcn_blocks > 14 || e->dp_tcn_layers < 1 || e->E % 4 ||
      e->feat_dim % 8) {
    set_err("engine: unsupported DPCCN configuration (fuse %d: additive 1 / multiply 2 / FiLM 3; tcn %d x %d; spk_emb_dim %d)",
            e->dp_fuse, e->dp_tcn_layers, e->dp_tcn_blocks, e->E);
    return WS_ERR_INVALID;
  }
  e->dw = e->persist.alloc(e->hw.size());
  WS_PTR(e->dw);
  if ((rc = to_device(e, e->dw, e->hw.data(), e->hw.size() * 4)) != WS_OK) return rc;
  // ---- shapes ----
  if (!require(e, "conv2d.weight", {16, 2, 3, 3}) || !require(e, "conv2d.bias", {16}) ||
      !require(e, "deconv2d.weight", {32, 2, 3, 3}) || !require(e, "deconv2d.bias", {2}) ||
      !require(e, "avg_proj.weight", {32, 64, 1, 1}) || !require(e, "avg_proj.bias", {32}))
    return WS_ERR_INVALID;
  for (int i = 0; i < 4; ++i) {
    const std::string p = "avg_pool." + std::to_string(i) + ".1.";
    if (!require(e, p + "weight", {8, 32, 1, 1}) || !require(e, p + "bias", {8})) return WS_ERR_INVALID;
  }
  if (e->dp_fuse == 3) {
    if (!require(e, "spk_fuse.fc.gamma_fcs.0.weight", {kDpBins, e->E}) || !require(e, "spk_fuse.fc.gamma_fcs.0.bias", {kDpBins}) ||
        !require(e, "spk_fuse.fc.beta_fcs.0.weight", {kDpBins, e->E}) || !require(e, "spk_fuse.fc.beta_fcs.0.bias", {kDpBins}))
      return WS_ERR_INVALID;
    std::vector<float> b1(e->host("spk_fuse.fc.gamma_fcs.0.bias"), e->host("spk_fuse.fc.gamma_fcs.0.bias") + kDpBins);
    for (float& v : b1) v += 1.0f;                     // x (1 + gamma(e)) + beta(e)   (norm.py:116-134)
    if ((rc = dp_keep(e, "film_gamma_bias1", b1)) != WS_OK) return rc;
  } else if (!require(e, "spk_fuse.fc.linear.weight", {kDpBins, e->dp_fuse == 0 ? kDpBins + e->E : e->E}) ||
             !require(e, "spk_fuse.fc.linear.bias", {kDpBins})) {
    return WS_ERR_INVALID;     // (concat: Linear over the frequency axis of cat[x, e], speaker.py:95-101)
  }
  for (int l = 0; l < e->dp_tcn_layers; ++l)
    for (int b = 0; b < e->dp_tcn_blocks; ++b) {
      const std::string p = "tcn_layers." + std::to_string(l) + "." + std::to_string(b) + ".";
      if (!require(e, p + "dconv1.weight", {384, 1, 3}) || !require(e, p + "dconv1.bias", {384}) ||
          !require(e, p + "dconv2.weight", {384, 384, 1}) || !require(e, p + "dconv2.bias", {384}))
        return WS_ERR_INVALID;
    }
  // ---- convolution operands ----
  for (const DpDense& d : kDpDense)
    for (int i = 0; i < 5; ++i) {
      const int ci = d.C0 + i * d.g, co = i < 4 ? d.g : d.co5;
      const std::string p = std::string(d.prefix) + "conv" + std::to_string(i + 1) + ".conv2d.";
      if (!require(e, p + "weight", {co, ci, 3, 3}) || !require(e, p + "bias", {co})) return WS_ERR_INVALID;
      if ((rc = dp_keep(e, p + "pack", dp_pack3x3(dp_conv_rows(e->host(p + "weight"), co, ci, ci), ci, co))) != WS_OK) return rc;
    }
  for (const DpConv& c : kDpEnc) {
    const std::string p = std::string(c.prefix) + "conv2d.";
    if (!require(e, p + "weight", {c.cout, c.cin, 3, 3}) || !require(e, p + "bias", {c.cout})) return WS_ERR_INVALID;
    if ((rc = dp_keep(e, p + "rows", dp_conv_rows(e->host(p + "weight"), c.cout, c.cin, c.cin))) != WS_OK) return rc;
  }
  for (const DpConv& c : kDpDec) {
    const std::string p = std::string(c.prefix) + "convtrans2d.";
    if (!require(e, p + "weight", {c.cin, c.cout, 3, 3}) || !require(e, p + "bias", {c.cout})) return WS_ERR_INVALID;
    if ((rc = dp_keep(e, p + "rows", dp_convT_rows(e->host(p + "weight"), c.cin, c.cout, c.cout))) != WS_OK) return rc;
  }
  {
    std::vector<float> w_in = dp_conv_rows(e->host("conv2d.weight"), 16, 2, 4);            // (re, im, 0, 0) pixels
    std::vector<float> w_out = dp_convT_rows(e->host("deconv2d.weight"), 32, 2, 4), b_out(4, 0.f);
    b_out[0] = e->host("deconv2d.bias")[0], b_out[1] = e->host("deconv2d.bias")[1];
    e->dp_w_in = upload(e, e->persist, w_in.data(), w_in.size());
    e->dp_w_out = upload(e, e->persist, w_out.data(), w_out.size());
    e->dp_b_out = upload(e, e->persist, b_out.data(), b_out.size());
    WS_PTR(e->dp_w_in && e->dp_w_out && e->dp_b_out);
  }
  // ---- DFT bases (functional_dpccn._dft_tables: periodic hann window, float64 then rounded) ----
  {
    const int n = kDpWin, nf = kDpBins;
    std::vector<float> ana(size_t(kDpLd) * n, 0.f), syn(size_t(n) * kDpLd, 0.f);
    const double pi = 3.14159265358979323846;
    for (int f = 0; f < nf; ++f) {
      const double ck = (f == 0 || f == nf - 1) ? 1.0 : 2.0;
      for (int k = 0; k < n; ++k) {
        const double win = 0.5 - 0.5 * cos(2.0 * pi * k / n), ang = 2.0 * pi * double(f) * k / n;
        ana[size_t(4 * f) * n + k] = static_cast<float>(cos(ang) * win);
        ana[size_t(4 * f + 1) * n + k] = static_cast<float>(-sin(ang) * win);
        syn[size_t(k) * kDpLd + 4 * f] = static_cast<float>(ck * cos(ang) / n * win);
        if (f != 0 && f != nf - 1) syn[size_t(k) * kDpLd + 4 * f + 1] = static_cast<float>(-(ck * sin(ang)) / n * win);
      }
    }
    e->dp_ana4 = upload(e, e->persist, ana.data(), ana.size());
    e->dp_syn4 = upload(e, e->persist, syn.data(), syn.size());
    WS_PTR(e->dp_ana4 && e->dp_syn4);
  }
  {
    std::vector<float> ones(384, 1.f), zeros(384, 0.f);
    e->dp_ones = upload(e, e->persist, ones.data(), ones.size());
    e->dp_zeros = upload(e, e->persist, zeros.data(), zeros.size());
    WS_PTR(e->dp_ones && e->dp_zeros);
  }
  if (e->use_xform) {
    const Tensor* t0 = e->find("spk_transform.transforms.0.weight");
    if (!t0 || t0->dims.size() < 2 || t0->dims[1] != e->E || !e->find("spk_transform.transforms.1.weight") ||
        !e->find("spk_transform.transforms.3.weight")) {
      set_err("engine: spk_transform tensors missing or mis-shaped");
      return WS_ERR_INVALID;
    }
  }
  if (e->joint) {
    if ((rc = e->spk_kind == 2 ? prep_campplus(e) : e->spk_kind == 1 ? prep_ecapa(e) : prep_resnet(e)) != WS_OK) return rc;
    if ((rc = e->spk_feat ? prep_fbank(e) : prep_mel_frontend(e)) != WS_OK) return rc;
  }
  if (!e->dry && hipStreamSynchronize(e->stream) != hipSuccess) {
    set_err("engine: weight preparation failed on the device");
    return WS_ERR_LAUNCH;
  }
  return WS_OK;
}

// y (rows of stride ldy) = IN(ELU(x)) or ELU(IN(x)) over the P positions of each of G rows, x dense [G*P][C]
int dp_in_act(ws_engine* e, const float* x, int G, long long P, int C, int flags, float* y, long long ldy) {
  void* s = e->stream;
  Arena& a = e->work;
  const Arena::Mark mk = a.mark();
  int nsplit = static_cast<int>(P / 32);
  const int cap = 1024 / G > 1 ? 1024 / G : 1;
  if (nsplit > cap) nsplit = cap;
  if (nsplit < 1) nsplit = 1;
  const long long cnt = (long long)G * 2 * C;
  float* slab = a.alloc(size_t(nsplit) * cnt);
  float* sums = a.alloc(size_t(cnt));
  float* stats = a.alloc(size_t(cnt));
  WS_PTR(slab && sums && stats);
  WS_RUN(e, ws_in_act_sums(x, nullptr, 0, nullptr, static_cast<int>(P), G, nsplit, C, flags, slab, s));
  WS_RUN(e, ws_reduce_slabs(slab, nsplit, cnt, cnt, sums, 0, 0, s));
  WS_RUN(e, ws_inorm_finalize(sums, G, C, P, kInEps, stats, s));
  WS_RUN(e, ws_in_act_apply(x, stats, (long long)G * P, static_cast<int>(P), C, flags, y, ldy, s));
  a.release(mk);
  return WS_OK;
}

// y[M][Cout] (row stride ldy) = 3 x 3 / padding 1 convolution of the image x [R][H][W][Cin] with stride (1, sw)
// (mode 0: Conv2d, W [Cout][9 Cin]) or its transposed counterpart (mode 1: ConvTranspose2d, output grid [H][Wo])
int dp_conv_view(ws_engine* e, const float* x, int R, int H, int W, int Cin, int mode, int Wo, int sw, const float* Wm, int Cout,
                 const float* bias, float* y, long long ldy) {
  ws_gemm_nt_args g = {};
  g.A = x, g.W = Wm, g.bias = bias, g.C = y;
  g.a_div = kBig, g.a_s2 = 9 * Cin, g.c_div = kBig, g.c_s2 = ldy, g.st_div1 = 1, g.st_div2 = 1;
  g.M = R * H * Wo, g.N = Cout, g.K = 9 * Cin, g.ldw = 9 * Cin, g.vec = 3 | 4;
  g.conv.on = 1, g.conv.mode = mode, g.conv.H = H, g.conv.W = W, g.conv.C = Cin, g.conv.Ho = H, g.conv.Wo = Wo;
  g.conv.k = 3, g.conv.sh = 1, g.conv.sw = sw, g.conv.p = 1, g.conv.dil = 1;
  WS_RUN(e, ws_gemm_nt(&g, e->stream));
  return WS_OK;
}

// C[M][N] (row stride ldc) = A[M][K] W[N][K]^T + bias (+ Rm, addressed like C): the 1 x 1 convolutions
int dp_gemm(ws_engine* e, const float* A, long long M, int K, const float* Wm, int N, const float* bias, const float* Rm, float* C,
            long long ldc) {
  ws_gemm_nt_args g = {};
  g.A = A, g.W = Wm, g.bias = bias, g.C = C, g.R = Rm;
  g.a_div = kBig, g.a_s2 = K, g.c_div = kBig, g.c_s2 = ldc, g.st_div1 = 1, g.st_div2 = 1;
  g.M = static_cast<int>(M), g.N = N, g.K = K, g.ldw = K, g.vec = vec_bits({K});
  WS_RUN(e, ws_gemm_nt(&g, e->stream));
  return WS_OK;
}

// DenseBlock (convs.py:80-112): big [M][C0 + 4g] holds the input in its first C0 columns; out [M][co5]
int dp_dense(ws_engine* e, const DpDense& d, int R, int H, int W, float* big, float* out) {
  const long long M = (long long)R * H * W;
  const int Ctot = d.C0 + 4 * d.g;
  Arena& a = e->work;
  for (int i = 0; i < 5; ++i) {
    const int ci = d.C0 + i * d.g, co = i < 4 ? d.g : d.co5;
    const std::string p = std::string(d.prefix) + "conv" + std::to_string(i + 1) + ".conv2d.";
    const Arena::Mark mk = a.mark();
    float* pre = a.alloc(size_t(M) * co);
    WS_PTR(pre);
    ws_conv3x3_args c = {};
    c.X = big, c.W = e->dp_w[p + "pack"], c.bias = e->dev(p + "bias"), c.Y = pre;
    c.ldx = Ctot, c.ldw = 9 * ci, c.ldy = co, c.B = R, c.H = H, c.Wd = W, c.Cin = ci, c.Cout = co;
    WS_RUN(e, ws_conv3x3(&c, e->stream));
    int rc;
    if (i < 4)
      rc = dp_in_act(e, pre, R, (long long)H * W, co, kInPre, big + ci, Ctot);
    else
      rc = dp_in_act(e, pre, R, (long long)H * W, co, kInPre, out, co);
    if (rc != WS_OK) return rc;
    a.release(mk);
  }
  return WS_OK;
}

// Conv2dBlock with stride (1, 2) (convs.py:28-50): y [R*H*Wo][cout] = IN(ELU(conv(x)))
int dp_conv_block(ws_engine* e, const DpConv& c, const float* x, int R, int H, int W, float* y, long long ldy) {
  const int Wo = (W - 1) / 2 + 1;
  const long long M = (long long)R * H * Wo;
  const std::string p = std::string(c.prefix) + "conv2d.";
  Arena& a = e->work;
  const Arena::Mark mk = a.mark();
  float* pre = a.alloc(size_t(M) * c.cout);
  WS_PTR(pre);
  int rc = dp_conv_view(e, x, R, H, W, c.cin, 0, Wo, 2, e->dp_w[p + "rows"], c.cout, e->dev(p + "bias"), pre, c.cout);
  if (rc != WS_OK) return rc;
  if ((rc = dp_in_act(e, pre, R, (long long)H * Wo, c.cout, kInPre, y, ldy)) != WS_OK) return rc;
  a.release(mk);
  return WS_OK;
}

// ConvTrans2dBlock with stride (1, 2) (convs.py:53-77): y [R*H*(2W - 1)][cout] = IN(ELU(conv_transpose(x)))
int dp_convT_block(ws_engine* e, const DpConv& c, const float* x, int R, int H, int W, float* y, long long ldy) {
  const int Wt = 2 * W - 1;
  const long long M = (long long)R * H * Wt;
  const std::string p = std::string(c.prefix) + "convtrans2d.";
  Arena& a = e->work;
  const Arena::Mark mk = a.mark();
  float* pre = a.alloc(size_t(M) * c.cout);
  WS_PTR(pre);
  int rc = dp_conv_view(e, x, R, H, W, c.cin, 1, Wt, 2, e->dp_w[p + "rows"], c.cout, e->dev(p + "bias"), pre, c.cout);
  if (rc != WS_OK) return rc;
  if ((rc = dp_in_act(e, pre, R, (long long)H * Wt, c.cout, kInPre, y, ldy)) != WS_OK) return rc;
  a.release(mk);
  return WS_OK;
}

// TCNBlock (convs.py:115-152) on [R][L][384]: out = x + conv1x1(ELU(IN(dwconv(ELU(IN(x))))))
int dp_tcn_block(ws_engine* e, const std::string& p, int dil, const float* x, int R, long long L, const float* ident, float* out) {
  const int C = 384;
  const long long M = (long long)R * L;
  Arena& a = e->work;
  const Arena::Mark mk = a.mark();
  float* y1 = a.alloc(size_t(M) * C);
  float* y2 = a.alloc(size_t(M) * C);
  WS_PTR(y1 && y2);
  int rc = dp_in_act(e, x, R, L, C, kInPost, y1, C);
  if (rc != WS_OK) return rc;
  WS_RUN(e, ws_dwconv_ex_fwd(y1, ident, e->dp_ones, e->dp_zeros, e->dev(p + "dconv1.weight"), e->dev(p + "dconv1.bias"), R,
                             static_cast<int>(L), C, 3, dil, static_cast<int>(L), e->dp_causal, y2, e->stream));
  if ((rc = dp_in_act(e, y2, R, L, C, kInPost, y1, C)) != WS_OK) return rc;
  if ((rc = dp_gemm(e, y1, M, C, e->dev(p + "dconv2.weight"), C, e->dev(p + "dconv2.bias"), x, out, C)) != WS_OK) return rc;
  a.release(mk);
  return WS_OK;
}

// wav [R][T], emb [R][E] -> est [R][T]
int dpccn_device(ws_engine* e, const float* wav, int R, int T, const float* emb_in, float* est) {
  const int n = kDpWin, hop = kHop, pad = n / 2, Tf = 1 + T / hop, F0 = kDpBins;
  void* s = e->stream;
  Arena& a = e->work;
  int rc;
  const long long M0 = (long long)R * Tf * F0;
  // ---- STFT (torch.stft, hann, centre, reflect): frames of the padded rows x the analysis basis ----
  const int ldo = (T + 2 * pad + 3) / 4 * 4;
  float* xp = a.alloc(size_t(R) * ldo);
  float* spec4 = a.alloc(size_t(R) * Tf * kDpLd);      // == [M0][4]: (re, im, 0, 0) per (row, frame, bin)
  WS_PTR(xp && spec4);
  if ((rc = zero_device(e, xp, size_t(R) * ldo * 4)) != WS_OK) return rc;
  WS_RUN(e, ws_preemph_pad(wav, R, T, pad, ldo, 0.0f, xp, s));
  {
    ws_gemm_nt_args g = {};
    g.A = xp, g.W = e->dp_ana4, g.C = spec4;
    g.a_div = Tf, g.a_s1 = ldo, g.a_s2 = hop, g.c_div = kBig, g.c_s2 = kDpLd, g.st_div1 = 1, g.st_div2 = 1;
    g.M = R * Tf, g.N = kDpLd, g.K = n, g.ldw = n, g.vec = 3;           // exact fp32 products, like the Python path
    WS_RUN(e, ws_gemm_nt(&g, s));
  }
  // ---- Conv2d(2 -> 16) straight into the first dense block's map, then the block, then the speaker fusion ----
  float* skip[8];
  int skipW[8], skipC[8];
  {
    const DpDense& d = kDpDense[0];
    float* big = a.alloc(size_t(M0) * (d.C0 + 4 * d.g));
    float* o = a.alloc(size_t(M0) * d.co5);
    skip[0] = a.alloc(size_t(M0) * d.co5);
    WS_PTR(big && o && skip[0]);
    if ((rc = dp_conv_view(e, spec4, R, Tf, F0, 4, 0, F0, 1, e->dp_w_in, 16, e->dev("conv2d.bias"), big, d.C0 + 4 * d.g)) != WS_OK)
      return rc;
    if ((rc = dp_dense(e, d, R, Tf, F0, big, o)) != WS_OK) return rc;
    const float* emb = emb_in;
    if ((rc = spk_transform(e, emb, R, &emb)) != WS_OK) return rc;
    float* sf = a.alloc(size_t(R) * F0);
    WS_PTR(sf);
    if (e->dp_fuse == 3) {
      float* bt = a.alloc(size_t(R) * F0);
      float* tmp = a.alloc(size_t(M0) * d.co5);
      WS_PTR(bt && tmp);
      if ((rc = linear(e, emb, R, e->E, e->dev("spk_fuse.fc.gamma_fcs.0.weight"), e->E, F0, e->dp_w["film_gamma_bias1"], 0, sf)) != WS_OK ||
          (rc = linear(e, emb, R, e->E, e->dev("spk_fuse.fc.beta_fcs.0.weight"), e->E, F0, e->dev("spk_fuse.fc.beta_fcs.0.bias"), 0, bt)) != WS_OK)
        return rc;
      WS_RUN(e, ws_scale_bf_fwd(o, sf, R, Tf, F0, d.co5, 0, tmp, s));
      WS_RUN(e, ws_scale_bf_fwd(tmp, bt, R, Tf, F0, d.co5, 1, skip[0], s));
    } else if (e->dp_fuse == 0) {   // concat: out[b, c, :, t] = Wx x[b, c, :, t] + (We e + bias)
      if ((rc = linear(e, emb, R, e->E, e->dev("spk_fuse.fc.linear.weight") + F0, F0 + e->E, F0, e->dev("spk_fuse.fc.linear.bias"), 0, sf)) != WS_OK)
        return rc;
      WS_RUN(e, ws_freq_linear_fwd(o, e->dev("spk_fuse.fc.linear.weight"), F0 + e->E, sf, R, Tf, F0, d.co5, skip[0], s));
    } else {
      if ((rc = linear(e, emb, R, e->E, e->dev("spk_fuse.fc.linear.weight"), e->E, F0, e->dev("spk_fuse.fc.linear.bias"), 0, sf)) != WS_OK)
        return rc;
      WS_RUN(e, ws_scale_bf_fwd(o, sf, R, Tf, F0, d.co5, e->dp_fuse == 2 ? 0 : 1, skip[0], s));
    }
    skipW[0] = F0, skipC[0] = d.co5;
  }
  // ---- encoder: four (strided conv, dense block) stages, three strided convs ----
  for (int i = 0; i < 7; ++i) {
    const DpConv& c = kDpEnc[i];
    const int Wi = skipW[i], Wo = (Wi - 1) / 2 + 1;
    const long long M = (long long)R * Tf * Wo;
    if (i < 4) {
      const DpDense& d = kDpDense[1 + i];
      float* big = a.alloc(size_t(M) * (d.C0 + 4 * d.g));
      skip[i + 1] = a.alloc(size_t(M) * d.co5);
      WS_PTR(big && skip[i + 1]);
      if ((rc = dp_conv_block(e, c, skip[i], R, Tf, Wi, big, d.C0 + 4 * d.g)) != WS_OK) return rc;
      if ((rc = dp_dense(e, d, R, Tf, Wo, big, skip[i + 1])) != WS_OK) return rc;
      skipC[i + 1] = d.co5;
    } else {
      skip[i + 1] = a.alloc(size_t(M) * c.cout);
      WS_PTR(skip[i + 1]);
      if ((rc = dp_conv_block(e, c, skip[i], R, Tf, Wi, skip[i + 1], c.cout)) != WS_OK) return rc;
      skipC[i + 1] = c.cout;
    }
    skipW[i + 1] = Wo;
  }
  // ---- TCN stack on rows of L = Tf * W positions ----
  const int W7 = skipW[7];
  const long long L = (long long)Tf * W7;
  float* tA = a.alloc(size_t(R) * L * 384);
  float* tB = a.alloc(size_t(R) * L * 384);
  float* ident = nullptr;
  {
    std::vector<float> id(size_t(R) * 2);
    for (int r = 0; r < R; ++r) id[2 * r] = 0.f, id[2 * r + 1] = 1.f;
    ident = upload(e, a, id.data(), id.size());
  }
  WS_PTR(tA && tB && ident);
  const float* cur = skip[7];
  float* nxt = tA;
  for (int l = 0; l < e->dp_tcn_layers; ++l)
    for (int b = 0; b < e->dp_tcn_blocks; ++b) {
      const std::string p = "tcn_layers." + std::to_string(l) + "." + std::to_string(b) + ".";
      if ((rc = dp_tcn_block(e, p, 1 << b, cur, R, L, ident, nxt)) != WS_OK) return rc;
      cur = nxt;
      nxt = nxt == tA ? tB : tA;
    }
  // ---- decoder: cat[skip, out] -> (dense block ->) transposed conv ----
  const float* out = cur;
  int Wc = W7, Cc = 384;
  for (int i = 0; i < 7; ++i) {
    const DpConv& c = kDpDec[i];
    const float* sk = skip[7 - i];
    const int Cs = skipC[7 - i];
    const long long M = (long long)R * Tf * Wc;
    const int Wt = 2 * Wc - 1;
    float* y = a.alloc(size_t(R) * Tf * Wt * c.cout);
    WS_PTR(y);
    if (i < 3) {
      float* cat = a.alloc(size_t(M) * (Cs + Cc));
      WS_PTR(cat);
      if ((rc = copy_cols(e, cat, Cs + Cc, sk, Cs, Cs, M)) != WS_OK || (rc = copy_cols(e, cat + Cs, Cs + Cc, out, Cc, Cc, M)) != WS_OK)
        return rc;
      if ((rc = dp_convT_block(e, c, cat, R, Tf, Wc, y, c.cout)) != WS_OK) return rc;
    } else {
      const DpDense& d = kDpDense[5 + (i - 3)];
      const int Ctot = d.C0 + 4 * d.g;
      float* big = a.alloc(size_t(M) * Ctot);
      float* o = a.alloc(size_t(M) * d.co5);
      WS_PTR(big && o);
      if ((rc = copy_cols(e, big, Ctot, sk, Cs, Cs, M)) != WS_OK || (rc = copy_cols(e, big + Cs, Ctot, out, Cc, Cc, M)) != WS_OK)
        return rc;
      if ((rc = dp_dense(e, d, R, Tf, Wc, big, o)) != WS_OK) return rc;
      if ((rc = dp_convT_block(e, c, o, R, Tf, Wc, y, c.cout)) != WS_OK) return rc;
    }
    out = y, Wc = Wt, Cc = c.cout;
  }
  if (Wc != F0) {
    set_err("engine: DPCCN decoder grid %d does not match the spectrogram's %d bins", Wc, F0);
    return WS_ERR_LAUNCH;
  }
  // ---- last dense block on cat[skip0, out], pooling branches, projection, ConvTranspose2d(32 -> 2) ----
  float* cat64 = a.alloc(size_t(M0) * 64);
  float* feat = a.alloc(size_t(M0) * 32);
  WS_PTR(cat64 && feat);
  {
    const DpDense& d = kDpDense[9];
    const int Ctot = d.C0 + 4 * d.g;
    float* big = a.alloc(size_t(M0) * Ctot);
    WS_PTR(big);
    if ((rc = copy_cols(e, big, Ctot, skip[0], skipC[0], skipC[0], M0)) != WS_OK ||
        (rc = copy_cols(e, big + skipC[0], Ctot, out, Cc, Cc, M0)) != WS_OK)
      return rc;
    if ((rc = dp_dense(e, d, R, Tf, F0, big, feat)) != WS_OK) return rc;
  }
  if ((rc = copy_cols(e, cat64, 64, feat, 32, 32, M0)) != WS_OK) return rc;
  const int pool[4] = {4, 8, 16, 32};
  for (int i = 0; i < 4; ++i) {
    const int sz = pool[i], h = Tf / sz, w = F0 / sz;
    const std::string p = "avg_pool." + std::to_string(i) + ".1.";
    const Arena::Mark mk = a.mark();
    float* av = a.alloc(size_t(R) * h * w * 32);
    float* pc = a.alloc(size_t(R) * h * w * 8);
    float* up = a.alloc(size_t(M0) * 8);
    WS_PTR(av && pc && up);
    WS_RUN(e, ws_avgpool_fwd(feat, R, Tf, F0, 32, sz, av, s));
    if ((rc = dp_gemm(e, av, (long long)R * h * w, 32, e->dev(p + "weight"), 8, e->dev(p + "bias"), nullptr, pc, 8)) != WS_OK) return rc;
    WS_RUN(e, ws_bilinear_fwd(pc, R, h, w, Tf, F0, 8, up, s));
    if ((rc = copy_cols(e, cat64 + 32 + 8 * i, 64, up, 8, 8, M0)) != WS_OK) return rc;
    a.release(mk);
  }
  float* proj = a.alloc(size_t(M0) * 32);
  float* est4 = a.alloc(size_t(M0) * 4);               // == [R * Tf][1028]
  WS_PTR(proj && est4);
  if ((rc = dp_gemm(e, cat64, M0, 64, e->dev("avg_proj.weight"), 32, e->dev("avg_proj.bias"), nullptr, proj, 32)) != WS_OK) return rc;
  if ((rc = dp_conv_view(e, proj, R, Tf, F0, 32, 1, F0, 1, e->dp_w_out, 4, e->dp_b_out, est4, 4)) != WS_OK) return rc;
  // ---- inverse STFT (torch.istft, hann, centre, length = T): synthesis GEMM, overlap-add, 1 / window envelope ----
  {
    const int full = pad + T, ld = (T + 3) / 4 * 4;
    float* fr = a.alloc(size_t(R) * Tf * n);
    float* y = a.alloc(size_t(R) * full);
    float* o = a.alloc(size_t(R) * ld);
    WS_PTR(fr && y && o);
    ws_gemm_nt_args g = {};
    g.A = est4, g.W = e->dp_syn4, g.C = fr;
    g.a_div = kBig, g.a_s2 = kDpLd, g.c_div = kBig, g.c_s2 = n, g.st_div1 = 1, g.st_div2 = 1;
    g.M = R * Tf, g.N = n, g.K = kDpLd, g.ldw = kDpLd, g.vec = 3;
    WS_RUN(e, ws_gemm_nt(&g, s));
    WS_RUN(e, ws_ola_fwd(fr, nullptr, R, Tf, n, hop, full, y, s));
    std::vector<double> env(size_t(Tf - 1) * hop + n, 0.0);
    const double pi = 3.14159265358979323846;
    for (int k = 0; k < n; ++k) {
      const float wf = static_cast<float>(0.5 - 0.5 * cos(2.0 * pi * k / n));
      const double w2 = double(wf) * double(wf);
      for (int t = 0; t < Tf; ++t) env[size_t(t) * hop + k] += w2;
    }
    std::vector<float> inv(ld, 0.f);
    for (int i = 0; i < T; ++i) inv[i] = static_cast<float>(1.0 / env[size_t(pad) + i]);
    float* dinv = upload(e, a, inv.data(), inv.size());
    WS_PTR(dinv);
    if ((rc = zero_device(e, o, size_t(R) * ld * 4)) != WS_OK) return rc;
    if ((rc = copy_cols(e, o, ld, y + pad, full, T, R)) != WS_OK) return rc;
    WS_RUN(e, ws_affine_fwd(o, dinv, nullptr, 0.0f, R, R, ld, o, s));
    if ((rc = copy_cols(e, est, T, o, ld, T, R)) != WS_OK) return rc;
  }
  return WS_OK;
}

// =================================================================================================================
// TF-GridNet (arch 3): the launch plan of wesep_amd/models/tfgridnet.py (wesep/models/tfgridnet.py:197-302,
// wesep/modules/tfgridnet/gridnet_block.py:118-227) for the shipped recipe's geometry -- one microphone, one source,
// emb_dim 128, emb_ks = emb_hs = 1, lstm_hidden_units <= 256 (zero-padded to the 256 units the recurrence kernels are
// built for), 4 heads.  STFT / iSTFT as DFT-basis GEMMs like the DPCCN plan; Conv2d(2 -> C) + GroupNorm(1, C); per
// block: speaker fusion, the intra-frame path (row LayerNorm, BLSTM over the bins of a frame on the blocked-layout
// kernels of the pBSRNN plan, Linear + residual in the output GEMM), the inter-frame path (the same on a STRIDED
// sequence map: sequence (b, q) walks the frames -- the Python path transposes the map twice instead), attention (one
// projection GEMM for Q / K / V, ws_heads_fwd, grouped logits GEMM with the padded keys masked, row softmax, grouped
// value GEMM, head merge, projection + PReLU + LayerNorm over (bins, channels), residual); ConvTranspose2d(C -> 2).
// The mixture is scaled by its standard deviation on the host (tfgridnet.py:222-226), the estimate scaled back.
// =================================================================================================================
struct GridBlock {
  RnnPrep intra, inter;
  float *wqkv, *bqkv;                    // [nh*(2E + cp)][C] rows (Q heads | K heads | V heads), bias
  float *gam[3], *bet[3];                // per projection: [nh][Q*ch], index q*ch + e
  float *proj_g, *proj_b;                // [Q*C], index q*C + c
};
struct GridNet {
  int n_fft = 128, hop = 64, Q = 65, C = 128, hid = 192, nh = 4, E = 8, layers = 6, fuse = 2;
  float *ana4 = nullptr, *syn4 = nullptr, *w_in = nullptr, *w_out = nullptr, *b_out = nullptr;
  float *ones_c = nullptr, *zeros_c = nullptr, *ones_qc = nullptr, *zeros_qc = nullptr;
  float *id_st = nullptr, *slope1 = nullptr, *film_bias1 = nullptr;
  std::vector<GridBlock> blocks;
};
void grid_free(GridNet* g) { delete g; }

// nn.LSTM tensors of hidden size h -> the 256-unit layout (functional_tfgridnet.pad_lstm): gate-major rows g*256 + u
int grid_prep_rnn(ws_engine* e, const std::string& path, int C, int h, RnnPrep* r) {
  static const char* sfx[2] = {"", "_reverse"};
  const std::string rnn = path + "rnn.";
  float* dev_w[2][3];                    // per direction: w_ih [1024][C], w_hh [1024][256], b [1024]
  for (int d = 0; d < 2; ++d) {
    const std::string s = sfx[d];
    if (!require(e, rnn + "weight_ih_l0" + s, {4 * h, C}) || !require(e, rnn + "weight_hh_l0" + s, {4 * h, h}) ||
        !require(e, rnn + "bias_ih_l0" + s, {4 * h}) || !require(e, rnn + "bias_hh_l0" + s, {4 * h}))
      return WS_ERR_INVALID;
    const float *wi = e->host(rnn + "weight_ih_l0" + s), *wh = e->host(rnn + "weight_hh_l0" + s);
    const float *bi = e->host(rnn + "bias_ih_l0" + s), *bh = e->host(rnn + "bias_hh_l0" + s);
    std::vector<float> wip(size_t(kG4) * C, 0.f), whp(size_t(kG4) * kH, 0.f), bp(kG4, 0.f);
    for (int g = 0; g < 4; ++g)
      for (int u = 0; u < h; ++u) {
        const size_t src = size_t(g) * h + u, dst = size_t(g) * kH + u;
        memcpy(&wip[dst * C], wi + src * C, size_t(C) * 4);
        memcpy(&whp[dst * kH], wh + src * h, size_t(h) * 4);
        bp[dst] = bi[src] + bh[src];
      }
    dev_w[d][0] = upload(e, e->persist, wip.data(), wip.size());
    dev_w[d][1] = upload(e, e->persist, whp.data(), whp.size());
    dev_w[d][2] = upload(e, e->persist, bp.data(), bp.size());
    WS_PTR(dev_w[d][0] && dev_w[d][1] && dev_w[d][2]);
  }
  if (!require(e, path + "norm.weight", {C}) || !require(e, path + "norm.bias", {C}) ||
      !require(e, path + "linear.weight", {C, 2 * h}) || !require(e, path + "linear.bias", {C}))
    return WS_ERR_INVALID;
  // Linear(2h -> C): each half of the hidden columns zero-padded to 256 (pad_hidden_cols)
  std::vector<float> lin(size_t(C) * 2 * kH, 0.f), zero(kG4, 0.f);
  const float* lw = e->host(path + "linear.weight");
  for (int n = 0; n < C; ++n) {
    memcpy(&lin[size_t(n) * 2 * kH], lw + size_t(n) * 2 * h, size_t(h) * 4);
    memcpy(&lin[size_t(n) * 2 * kH + kH], lw + size_t(n) * 2 * h + h, size_t(h) * 4);
  }
  Arena& a = e->persist;
  float* dlin = upload(e, a, lin.data(), lin.size());
  float* dzero = upload(e, a, zero.data(), zero.size());
  float* wcat = a.alloc(size_t(2) * kG4 * C);
  r->norm_w = e->dev(path + "norm.weight"), r->norm_b = e->dev(path + "norm.bias"), r->proj_b = e->dev(path + "linear.bias");
  r->whf = dev_w[0][1], r->whr = dev_w[1][1];
  r->bcat = a.alloc(2 * kG4);
  r->wih_pack = a.alloc(size_t(2) * kG4 * C);
  r->proj_pack = a.alloc(size_t(C) * 2 * kH);
  r->fpack = a.alloc(WS_LSTM_FUSED_PACK_FLOATS);
  r->pack16 = a.alloc(WS_LSTM_PACK_FLOATS);
  r->pack32 = a.alloc(WS_LSTM_PACK_FLOATS);
  float* bwd_scratch = a.alloc(WS_LSTM_PACK_FLOATS);
  WS_PTR(dlin && dzero && wcat && r->bcat && r->wih_pack && r->proj_pack && r->fpack && r->pack16 && r->pack32 && bwd_scratch);
  void* s = e->stream;
  WS_RUN(e, ws_lstm_cat_ih(dev_w[0][0], dev_w[1][0], dev_w[0][2], dzero, dev_w[1][2], dzero, C, wcat, r->bcat, s));
  WS_RUN(e, ws_pack_w(wcat, 2 * kG4, C, C, 0, 0, r->wih_pack, s));
  WS_RUN(e, ws_pack_w(dlin, C, 2 * kH, 2 * kH, 0, 1, r->proj_pack, s));
  WS_RUN(e, ws_lstm_pack_fused(dev_w[0][0], dev_w[1][0], r->whf, r->whr, r->fpack, s));
  WS_RUN(e, ws_lstm_pack(r->whf, r->whr, r->pack16, bwd_scratch, WS_LSTM_BF16X3_BLK16, s));
  WS_RUN(e, ws_lstm_pack(r->whf, r->whr, r->pack32, bwd_scratch, WS_LSTM_BF16X3_BLK, s));
  return WS_OK;
}

int prepare_gridnet(ws_engine* e) {
  if (!e->grid) e->grid = new GridNet();
  GridNet& n = *e->grid;
  e->sr = static_cast<int>(meta_or(e, "sample_rate", 16000));
  int rc = read_speaker_meta(e);
  if (rc != WS_OK) return rc;
  n.n_fft = static_cast<int>(meta_or(e, "n_fft", 128)), n.hop = static_cast<int>(meta_or(e, "stride", 64));
  n.Q = n.n_fft / 2 + 1, n.C = static_cast<int>(meta_or(e, "emb_dim", 128)), n.hid = static_cast<int>(meta_or(e, "lstm_hidden_units", 192));
  n.nh = static_cast<int>(meta_or(e, "attn_n_head", 4)), n.E = static_cast<int>(meta_or(e, "attn_E", 8));
  n.layers = static_cast<int>(meta_or(e, "n_layers", 6)), n.fuse = static_cast<int>(meta_or(e, "spk_fuse_type", 2));
  const int C = n.C, Q = n.Q, nh = n.nh, E = n.E, cp = C / (nh > 0 ? nh : 1);
  if (meta_or(e, "emb_ks", 1) != 1 || meta_or(e, "emb_hs", 1) != 1 || meta_or(e, "n_srcs", 1) != 1 || meta_or(e, "n_imics", 1) != 1 ||
      C != kN || n.hid < 4 || n.hid > kH || n.hid % 4 || nh < 1 || nh > 8 || C % nh || E % 4 || cp % 4 || n.n_fft % 8 ||
      n.n_fft < 16 || n.n_fft > 1024 || n.hop * 2 != n.n_fft || (long long)Q * C > 9216 || n.fuse < 0 || n.fuse > 3 ||
      n.layers < 1 || e->E % 4 || e->feat_dim % 8) {
    set_err("engine: the TF-GridNet plan is built for the recipe's geometry (emb_dim 128, emb_ks = emb_hs = 1, one microphone "
            "and source, hidden <= 256 and %% 4, heads <= 8 with widths %% 4, stride = n_fft / 2, (n_fft / 2 + 1) * 128 <= 9216, "
            "concat / multiply / additive / FiLM fusion)");
    return WS_ERR_INVALID;
  }
  e->dw = e->persist.alloc(e->hw.size());
  WS_PTR(e->dw);
  if ((rc = to_device(e, e->dw, e->hw.data(), e->hw.size() * 4)) != WS_OK) return rc;
  if (!require(e, "conv.0.weight", {C, 2, 3, 3}) || !require(e, "conv.0.bias", {C}) || !require(e, "conv.1.weight", {C}) ||
      !require(e, "conv.1.bias", {C}) || !require(e, "deconv.weight", {C, 2, 3, 3}) || !require(e, "deconv.bias", {2}))
    return WS_ERR_INVALID;
  if (n.fuse == 3) {
    if (!require(e, "spk_fuse.fc.gamma_fcs.0.weight", {Q, e->E}) || !require(e, "spk_fuse.fc.gamma_fcs.0.bias", {Q}) ||
        !require(e, "spk_fuse.fc.beta_fcs.0.weight", {Q, e->E}) || !require(e, "spk_fuse.fc.beta_fcs.0.bias", {Q}))
      return WS_ERR_INVALID;
    std::vector<float> b1(e->host("spk_fuse.fc.gamma_fcs.0.bias"), e->host("spk_fuse.fc.gamma_fcs.0.bias") + Q);
    for (float& v : b1) v += 1.0f;
    n.film_bias1 = upload(e, e->persist, b1.data(), b1.size());
    WS_PTR(n.film_bias1);
  } else if (!require(e, "spk_fuse.fc.linear.weight", {Q, n.fuse == 0 ? Q + e->E : e->E}) || !require(e, "spk_fuse.fc.linear.bias", {Q})) {
    return WS_ERR_INVALID;     // (concat: Linear over the frequency axis of cat[x, e], speaker.py:95-101)
  }
  {
    std::vector<float> w_in = dp_conv_rows(e->host("conv.0.weight"), C, 2, 4);
    std::vector<float> w_out = dp_convT_rows(e->host("deconv.weight"), C, 2, 4), b_out(4, 0.f);
    b_out[0] = e->host("deconv.bias")[0], b_out[1] = e->host("deconv.bias")[1];
    n.w_in = upload(e, e->persist, w_in.data(), w_in.size());
    n.w_out = upload(e, e->persist, w_out.data(), w_out.size());
    n.b_out = upload(e, e->persist, b_out.data(), b_out.size());
    WS_PTR(n.w_in && n.w_out && n.b_out);
  }
  {   // DFT bases with (re, im, 0, 0) per bin (functional_dpccn._dft_tables)
    const int nn = n.n_fft, ld = 4 * Q;
    std::vector<float> ana(size_t(ld) * nn, 0.f), syn(size_t(nn) * ld, 0.f);
    const double pi = 3.14159265358979323846;
    for (int f = 0; f < Q; ++f) {
      const double ck = (f == 0 || f == Q - 1) ? 1.0 : 2.0;
      for (int k = 0; k < nn; ++k) {
        const double win = 0.5 - 0.5 * cos(2.0 * pi * k / nn), ang = 2.0 * pi * double(f) * k / nn;
        ana[size_t(4 * f) * nn + k] = static_cast<float>(cos(ang) * win);
        ana[size_t(4 * f + 1) * nn + k] = static_cast<float>(-sin(ang) * win);
        syn[size_t(k) * ld + 4 * f] = static_cast<float>(ck * cos(ang) / nn * win);
        if (f != 0 && f != Q - 1) syn[size_t(k) * ld + 4 * f + 1] = static_cast<float>(-(ck * sin(ang)) / nn * win);
      }
    }
    n.ana4 = upload(e, e->persist, ana.data(), ana.size());
    n.syn4 = upload(e, e->persist, syn.data(), syn.size());
    WS_PTR(n.ana4 && n.syn4);
  }
  {
    std::vector<float> ones(size_t(Q) * C, 1.f), zeros(size_t(Q) * C, 0.f), id(size_t(2) * C, 0.f);
    for (int c = 0; c < C; ++c) id[C + c] = 1.f;                        // (mean 0 | rstd 1)
    const float one = 1.f;
    n.ones_qc = upload(e, e->persist, ones.data(), ones.size());
    n.zeros_qc = upload(e, e->persist, zeros.data(), zeros.size());
    n.id_st = upload(e, e->persist, id.data(), id.size());
    n.slope1 = upload(e, e->persist, &one, 1);
    WS_PTR(n.ones_qc && n.zeros_qc && n.id_st && n.slope1);
    n.ones_c = n.ones_qc, n.zeros_c = n.zeros_qc;                       // any prefix of C elements
  }
  n.blocks.resize(n.layers);
  for (int l = 0; l < n.layers; ++l) {
    GridBlock& b = n.blocks[l];
    const std::string p = "blocks." + std::to_string(l) + ".";
    // nn.Module names: intra_norm / intra_rnn / intra_linear -> one prefix per path
    for (int path = 0; path < 2; ++path) {
      const std::string q = p + (path ? "inter_" : "intra_");
      // grid_prep_rnn reads <q>norm., <q>rnn., <q>linear.
      if ((rc = grid_prep_rnn(e, q, C, n.hid, path ? &b.inter : &b.intra)) != WS_OK) return rc;
    }
    const char* proj[3] = {"attn_conv_Q.", "attn_conv_K.", "attn_conv_V."};
    const char* norm[3] = {"attn_norm_Q.", "attn_norm_K.", "attn_norm_V."};
    const int width[3] = {nh * E, nh * E, C}, chs[3] = {E, E, cp};
    const int ld = 2 * nh * E + C;
    std::vector<float> wq(size_t(ld) * C), bq(ld);
    int row = 0;
    for (int j = 0; j < 3; ++j) {
      if (!require(e, p + proj[j] + "weight", {width[j], C, 1, 1}) || !require(e, p + proj[j] + "bias", {width[j]}) ||
          !require(e, p + norm[j] + "gamma", {1, nh, chs[j], 1, Q}) || !require(e, p + norm[j] + "beta", {1, nh, chs[j], 1, Q}) ||
          !require(e, p + norm[j] + "act.weight", {nh}))
        return WS_ERR_INVALID;
      memcpy(&wq[size_t(row) * C], e->host(p + proj[j] + "weight"), size_t(width[j]) * C * 4);
      memcpy(&bq[row], e->host(p + proj[j] + "bias"), size_t(width[j]) * 4);
      row += width[j];
      const int ch = chs[j];
      std::vector<float> g(size_t(nh) * Q * ch), bt(size_t(nh) * Q * ch);
      const float *gs = e->host(p + norm[j] + "gamma"), *bs = e->host(p + norm[j] + "beta");
      for (int h = 0; h < nh; ++h)
        for (int ee = 0; ee < ch; ++ee)
          for (int q = 0; q < Q; ++q) {
            g[(size_t(h) * Q + q) * ch + ee] = gs[(size_t(h) * ch + ee) * Q + q];
            bt[(size_t(h) * Q + q) * ch + ee] = bs[(size_t(h) * ch + ee) * Q + q];
          }
      b.gam[j] = upload(e, e->persist, g.data(), g.size());
      b.bet[j] = upload(e, e->persist, bt.data(), bt.size());
      WS_PTR(b.gam[j] && b.bet[j]);
    }
    b.wqkv = upload(e, e->persist, wq.data(), wq.size());
    b.bqkv = upload(e, e->persist, bq.data(), bq.size());
    WS_PTR(b.wqkv && b.bqkv);
    if (!require(e, p + "attn_concat_proj.0.weight", {C, C, 1, 1}) || !require(e, p + "attn_concat_proj.0.bias", {C}) ||
        !require(e, p + "attn_concat_proj.1.weight", {1}) || !require(e, p + "attn_concat_proj.2.gamma", {1, C, 1, Q}) ||
        !require(e, p + "attn_concat_proj.2.beta", {1, C, 1, Q}))
      return WS_ERR_INVALID;
    std::vector<float> pg(size_t(Q) * C), pb(size_t(Q) * C);
    const float *gs = e->host(p + "attn_concat_proj.2.gamma"), *bs = e->host(p + "attn_concat_proj.2.beta");
    for (int c = 0; c < C; ++c)
      for (int q = 0; q < Q; ++q) {
        pg[size_t(q) * C + c] = gs[size_t(c) * Q + q];
        pb[size_t(q) * C + c] = bs[size_t(c) * Q + q];
      }
    b.proj_g = upload(e, e->persist, pg.data(), pg.size());
    b.proj_b = upload(e, e->persist, pb.data(), pb.size());
    WS_PTR(b.proj_g && b.proj_b);
  }
  if (e->use_xform) {
    const Tensor* t0 = e->find("spk_transform.transforms.0.weight");
    if (!t0 || t0->dims.size() < 2 || t0->dims[1] != e->E || !e->find("spk_transform.transforms.1.weight") ||
        !e->find("spk_transform.transforms.3.weight")) {
      set_err("engine: spk_transform tensors missing or mis-shaped");
      return WS_ERR_INVALID;
    }
  }
  if (e->joint) {
    if ((rc = e->spk_kind == 2 ? prep_campplus(e) : e->spk_kind == 1 ? prep_ecapa(e) : prep_resnet(e)) != WS_OK) return rc;
    if ((rc = e->spk_feat ? prep_fbank(e) : prep_mel_frontend(e)) != WS_OK) return rc;
  }
  if (!e->dry && hipStreamSynchronize(e->stream) != hipSuccess) {
    set_err("engine: weight preparation failed on the device");
    return WS_ERR_LAUNCH;
  }
  return WS_OK;
}

// out = res + Linear(BLSTM(xn)) on the sequences of `sm` (rows of 128 features; xn = the layer-normed rows): the body of
// resrnn() without its GroupNorm (functional_tfgridnet.BlstmLinearBlkFn)
int grid_rnn(ws_engine* e, const RnnPrep& w, const ws_seqmap& sm, const float* xn_rows, const float* res, float* out) {
  const int ntile = (sm.nseq + 31) / 32;
  const size_t nb = size_t(ntile) * sm.L;
  const int lmode = 2 * ntile <= 128 ? WS_LSTM_BF16X3_BLK16 : WS_LSTM_BF16X3_BLK;
  static const bool no_cluster = getenv("WS_ENGINE_NO_CLUSTER") != nullptr;
  const bool cluster = !no_cluster && sm.nseq % 64 == 0 && (sm.nseq / 32) * 8 <= e->cu_count && sm.L >= 64;
  const bool fused = !cluster && lmode == WS_LSTM_BF16X3_BLK;
  void* s = e->stream;
  Arena& a = e->work;
  const Arena::Mark mk = a.mark();
  float* gates = a.alloc(nb * 32 * 2 * kG4);
  float* cbuf = a.alloc(nb * 32 * 2 * kH);
  float* hcat = a.alloc(nb * 32 * 2 * kH);
  float* xn = a.alloc(nb * 32 * kN);
  WS_PTR(gates && cbuf && hcat && xn);
  ws_gemm_p2b_args p = {};
  p.A = xn_rows, p.sm = sm, p.lda = kN, p.K = kN, p.A_bl = xn;
  p.st_div1 = 1, p.st_m1 = 0, p.st_div2 = 1, p.st_m2 = 0, p.st_base = 0;
  if (fused) {
    p.N = 0;
    WS_RUN(e, ws_gemm_p2b(&p, s));
    ws_lstm_fused_args f = {};
    f.gates = gates, f.cbuf = cbuf, f.hcat = hcat, f.xn = xn, f.wpack = w.fpack, f.bias = w.bcat;
    f.nseq = sm.nseq, f.L = sm.L;
    WS_RUN(e, ws_lstm_fwd_fused(&f, s));
  } else {
    p.Wpack = w.wih_pack, p.bias = w.bcat, p.C = gates, p.N = 2 * kG4;
    WS_RUN(e, ws_gemm_p2b(&p, s));
    ws_lstm_args l = {};
    l.gates = gates, l.cbuf = cbuf, l.hcat = hcat;
    l.wpack = lmode == WS_LSTM_BF16X3_BLK16 ? w.pack16 : w.pack32;
    l.sq_s1 = sm.sq_s1, l.sq_s2 = sm.sq_s2, l.step_rows = sm.step_rows;
    l.nseq = sm.nseq, l.sq_div = sm.sq_div, l.L = sm.L, l.mode = lmode;
    if (cluster) {
      const int ncl = sm.nseq / 32;
      float* xchg = a.alloc(size_t(ncl) * 2 * 8 * 8192 / 4);
      unsigned* flags = reinterpret_cast<unsigned*>(a.alloc(size_t(ncl) * 8 + 8));
      WS_PTR(xchg && flags);
      if (!e->cl_status) {
        e->cl_status = reinterpret_cast<unsigned*>(e->persist.alloc(2));
        WS_PTR(e->cl_status);
        if (zero_device(e, e->cl_status, 8) != WS_OK) return WS_ERR_LAUNCH;
      }
      ws_lstm_cluster_args c = {};
      c.gates = gates, c.cbuf = cbuf, c.hcat = hcat, c.whh_f = w.whf, c.whh_r = w.whr;
      c.xchg = xchg, c.flags = flags, c.nseq = sm.nseq, c.L = sm.L, c.status = e->cl_status;
      WS_RUN(e, ws_lstm_fwd_cluster(&c, s));
      p.run_if = flags + size_t(ncl) * 8;      // the streaming pair repeats the layer only after a cluster time-out
      WS_RUN(e, ws_gemm_p2b(&p, s));
      l.run_if = p.run_if;
    }
    WS_RUN(e, ws_lstm_fwd(&l, s));
  }
  ws_gemm_b2p_args b = {};
  b.A = hcat, b.Wpack = w.proj_pack, b.bias = w.proj_b, b.R = res, b.C = out, b.sm = sm, b.ldc = kN, b.N = kN, b.K = 2 * kH;
  WS_RUN(e, ws_gemm_b2p(&b, s));
  a.release(mk);
  return WS_OK;
}

// C[g][M][N] = A[g][M][K] W[g][N][K]^T (+ bias[N]) for G groups in one launch (functional_tfgridnet.BatchedMatmulNTFn)
int grid_bmm(ws_engine* e, const float* A, const float* W, const float* bias, int G, int M, int K, int N, float* C) {
  std::vector<ws_group_nt> tab(G);
  for (int g = 0; g < G; ++g) {
    ws_group_nt d = {};
    d.W = W + size_t(g) * N * K, d.bias = bias, d.a_off = (long long)g * M * K, d.c_off = (long long)g * M * N;
    d.K = K, d.N = N, d.ldw = K;
    tab[g] = d;
  }
  const size_t nf = (sizeof(ws_group_nt) * G + 3) / 4;
  ws_group_nt* dt = reinterpret_cast<ws_group_nt*>(e->work.alloc(nf));
  WS_PTR(dt);
  int rc = to_device(e, dt, tab.data(), sizeof(ws_group_nt) * G);
  if (rc != WS_OK) return rc;
  ws_gemm_nt_args g = {};
  g.A = A, g.C = C, g.groups = dt;
  g.a_div = kBig, g.a_s2 = K, g.c_div = kBig, g.c_s2 = N, g.st_div1 = 1, g.st_div2 = 1;
  g.M = M, g.ngroups = G, g.max_n = N;
  g.vec = ((K % 4 == 0 && ((long long)M * K) % 4 == 0) ? 3 : 0) | 4;
  WS_RUN(e, ws_gemm_nt(&g, e->stream));
  return WS_OK;
}

// wav [R][T] (already divided by its standard deviation), emb [R][E] -> est [R][T] (still in normalised units)
int gridnet_device(ws_engine* e, const float* wav, int R, int T, const float* emb_in, float* est) {
  const GridNet& n = *e->grid;
  const int nf = n.n_fft, hop = n.hop, pad = nf / 2, Tf = 1 + T / hop, Q = n.Q, C = n.C, nh = n.nh, E = n.E, cp = C / nh;
  const int ld4 = 4 * Q, Tp = (Tf + 3) / 4 * 4, G = nh * R, D = Q * E, Dv = Q * cp, ldq = 2 * nh * E + C;
  const long long M = (long long)R * Tf * Q;
  void* s = e->stream;
  Arena& a = e->work;
  int rc;
  // ---- STFT ----
  const int ldo = (T + 2 * pad + 3) / 4 * 4;
  float* xp = a.alloc(size_t(R) * ldo);
  float* spec4 = a.alloc(size_t(R) * Tf * ld4);
  float* hA = a.alloc(size_t(M) * C);
  float* hB = a.alloc(size_t(M) * C);
  float* hC = a.alloc(size_t(M) * C);
  WS_PTR(xp && spec4 && hA && hB && hC);
  if ((rc = zero_device(e, xp, size_t(R) * ldo * 4)) != WS_OK) return rc;
  WS_RUN(e, ws_preemph_pad(wav, R, T, pad, ldo, 0.0f, xp, s));
  {
    ws_gemm_nt_args g = {};
    g.A = xp, g.W = n.ana4, g.C = spec4;
    g.a_div = Tf, g.a_s1 = ldo, g.a_s2 = hop, g.c_div = kBig, g.c_s2 = ld4, g.st_div1 = 1, g.st_div2 = 1;
    g.M = R * Tf, g.N = ld4, g.K = nf, g.ldw = nf, g.vec = 3;
    WS_RUN(e, ws_gemm_nt(&g, s));
  }
  // ---- Conv2d(2 -> C) + GroupNorm(1, C) ----
  if ((rc = dp_conv_view(e, spec4, R, Tf, Q, 4, 0, Q, 1, n.w_in, C, e->dev("conv.0.bias"), hB, C)) != WS_OK) return rc;
  {
    float* st = a.alloc(size_t(R) * 2);
    WS_PTR(st);
    if ((rc = tas_flat_stats(e, hB, R, (long long)Tf * Q * C, st)) != WS_OK) return rc;
    WS_RUN(e, ws_dwconv_fwd(hB, st, e->dev("conv.1.weight"), e->dev("conv.1.bias"), n.ones_c, n.zeros_c, R, Tf * Q, C, 1, 1, Tf * Q,
                            hA, s));
  }
  // ---- speaker fusion operands (the same before every block) ----
  const float* emb = emb_in;
  if ((rc = spk_transform(e, emb, R, &emb)) != WS_OK) return rc;
  float* sf = a.alloc(size_t(R) * Q);
  float* bt = n.fuse == 3 ? a.alloc(size_t(R) * Q) : nullptr;
  WS_PTR(sf && (n.fuse != 3 || bt));
  if (n.fuse == 3) {
    if ((rc = linear(e, emb, R, e->E, e->dev("spk_fuse.fc.gamma_fcs.0.weight"), e->E, Q, n.film_bias1, 0, sf)) != WS_OK ||
        (rc = linear(e, emb, R, e->E, e->dev("spk_fuse.fc.beta_fcs.0.weight"), e->E, Q, e->dev("spk_fuse.fc.beta_fcs.0.bias"), 0, bt)) != WS_OK)
      return rc;
  } else if (n.fuse == 0) {      // concat: the embedding's share of the Linear, We e + bias; the x share runs per block below
    if ((rc = linear(e, emb, R, e->E, e->dev("spk_fuse.fc.linear.weight") + Q, Q + e->E, Q, e->dev("spk_fuse.fc.linear.bias"), 0, sf)) != WS_OK)
      return rc;
  } else if ((rc = linear(e, emb, R, e->E, e->dev("spk_fuse.fc.linear.weight"), e->E, Q, e->dev("spk_fuse.fc.linear.bias"), 0, sf)) != WS_OK) {
    return rc;
  }
  std::vector<float> mask_h(Tp, 0.f);
  for (int t = Tf; t < Tp; ++t) mask_h[t] = -1e30f;
  float* mask = upload(e, a, mask_h.data(), mask_h.size());
  WS_PTR(mask);
  ws_seqmap intra = {}, inter = {};
  intra.nseq = R * Tf, intra.sq_div = kBig, intra.sq_s1 = 0, intra.sq_s2 = Q, intra.step_rows = 1, intra.L = Q;
  inter.nseq = R * Q, inter.sq_div = Q, inter.sq_s1 = (long long)Tf * Q, inter.sq_s2 = 1, inter.step_rows = Q, inter.L = Tf;
  float* h = hA;                          // block input / output; hB, hC rotate as scratch
  for (int l = 0; l < n.layers; ++l) {
    const GridBlock& b = n.blocks[l];
    const std::string p = "blocks." + std::to_string(l) + ".";
    const Arena::Mark mk = a.mark();
    float* x = hB;                         // fused input
    if (n.fuse == 3) {
      WS_RUN(e, ws_scale_bf_fwd(h, sf, R, Tf, Q, C, 0, hC, s));
      WS_RUN(e, ws_scale_bf_fwd(hC, bt, R, Tf, Q, C, 1, x, s));
    } else if (n.fuse == 0) {
      WS_RUN(e, ws_freq_linear_fwd(h, e->dev("spk_fuse.fc.linear.weight"), Q + e->E, sf, R, Tf, Q, C, x, s));
    } else {
      WS_RUN(e, ws_scale_bf_fwd(h, sf, R, Tf, Q, C, n.fuse == 2 ? 0 : 1, x, s));
    }
    float* y = a.alloc(size_t(M) * C);
    float* lnst = a.alloc(size_t(M) * 2);
    WS_PTR(y && lnst);
    // intra-frame path: x -> hC
    WS_RUN(e, ws_rowln_fwd(x, b.intra.norm_w, b.intra.norm_b, M, C, kLnEps, y, lnst, s));
    if ((rc = grid_rnn(e, b.intra, intra, y, x, hC)) != WS_OK) return rc;
    // inter-frame path: hC -> x  (strided sequences: no transposes)
    WS_RUN(e, ws_rowln_fwd(hC, b.inter.norm_w, b.inter.norm_b, M, C, kLnEps, y, lnst, s));
    if ((rc = grid_rnn(e, b.inter, inter, y, hC, x)) != WS_OK) return rc;
    // attention on `x` (the block's `inter` tensor)
    float* qkv = a.alloc(size_t(M) * ldq);
    float* Qa = a.alloc(size_t(G) * Tf * D);
    float* Ka = a.alloc(size_t(G) * Tp * D);
    float* Va = a.alloc(size_t(G) * Tp * Dv);
    float* VaT = a.alloc(size_t(G) * Tp * Dv);
    float* hst = a.alloc(size_t(nh) * R * Tf * 2);
    float* logits = a.alloc(size_t(G) * Tf * Tp);
    float* att = a.alloc(size_t(G) * Tf * Tp);
    float* ov = a.alloc(size_t(G) * Tf * Dv);
    WS_PTR(qkv && Qa && Ka && Va && VaT && hst && logits && att && ov);
    if ((rc = dp_gemm(e, x, M, C, b.wqkv, ldq, b.bqkv, nullptr, qkv, ldq)) != WS_OK) return rc;
    const char* norm[3] = {"attn_norm_Q.", "attn_norm_K.", "attn_norm_V."};
    float* outs[3] = {Qa, Ka, Va};
    const int chs[3] = {E, E, cp}, tps[3] = {Tf, Tp, Tp}, offs[3] = {0, nh * E, 2 * nh * E};
    for (int j = 0; j < 3; ++j) {
      ws_heads_args ha = {};
      ha.x = qkv + offs[j], ha.slope = e->dev(p + norm[j] + "act.weight"), ha.gamma = b.gam[j], ha.beta = b.bet[j];
      ha.y = outs[j], ha.stats = hst, ha.ldx = ldq, ha.B = R, ha.T = Tf, ha.Tp = tps[j], ha.Q = Q, ha.nh = nh, ha.ch = chs[j];
      ha.eps = kLnEps;
      WS_RUN(e, ws_heads_fwd(&ha, s));
    }
    if ((rc = grid_bmm(e, Qa, Ka, mask, G, Tf, D, Tp, logits)) != WS_OK) return rc;
    WS_RUN(e, ws_softmax_rows_fwd(logits, (long long)G * Tf, Tp, 1.0f / sqrtf(static_cast<float>(D)), att, s));
    for (int g = 0; g < G; ++g)
      WS_RUN(e, ws_transpose(Va + size_t(g) * Tp * Dv, Tp, Dv, Dv, VaT + size_t(g) * Tp * Dv, s));
    if ((rc = grid_bmm(e, att, VaT, nullptr, G, Tf, Tp, Dv, ov)) != WS_OK) return rc;
    // head merge: ov [nh][R][Tf][Q][cp] -> [R][Tf][Q][nh*cp]
    float* o = y;                          // y is free again
    for (int hd = 0; hd < nh; ++hd)
      for (int r = 0; r < R; ++r)
        if ((rc = copy_cols(e, o + (size_t(r) * Tf * Q) * C + hd * cp, C, ov + (size_t(hd) * R + r) * Tf * Dv, cp, cp,
                            (long long)Tf * Q)) != WS_OK)
          return rc;
    // projection + PReLU + LayerNorm over (bins, channels) + residual -> the next block's input
    float* p1 = a.alloc(size_t(M) * C);
    float* p2 = a.alloc(size_t(M) * C);
    float* rst = a.alloc(size_t(R) * Tf * 2);
    float* scr = a.alloc(size_t(M) * C);
    WS_PTR(p1 && p2 && rst && scr);
    if ((rc = dp_gemm(e, o, M, C, e->dev(p + "attn_concat_proj.0.weight"), C, e->dev(p + "attn_concat_proj.0.bias"), nullptr, p1, C)) != WS_OK)
      return rc;
    WS_RUN(e, ws_prelu_fwd(p1, nullptr, e->dev(p + "attn_concat_proj.1.weight"), M * C / 4, 4, static_cast<int>(M * C / 4), p2, s));
    if ((rc = tas_row_stats(e, p2, (long long)R * Tf, Q * C, rst)) != WS_OK) return rc;
    WS_RUN(e, ws_dwconv_fwd(p2, rst, b.proj_g, b.proj_b, n.ones_qc, n.zeros_qc, R * Tf, 1, Q * C, 1, 1, 1, p1, s));
    WS_RUN(e, ws_bn_prelu_fwd(p1, n.id_st, n.ones_c, n.zeros_c, x, n.slope1, M, C, scr, h, s));    // h = LN(..) + inter
    a.release(mk);
  }
  // ---- ConvTranspose2d(C -> 2) and the inverse STFT ----
  float* est4 = a.alloc(size_t(M) * 4);
  WS_PTR(est4);
  if ((rc = dp_conv_view(e, h, R, Tf, Q, C, 1, Q, 1, n.w_out, 4, n.b_out, est4, 4)) != WS_OK) return rc;
  {
    const int full = pad + T, ld = (T + 3) / 4 * 4;
    float* fr = a.alloc(size_t(R) * Tf * nf);
    float* y = a.alloc(size_t(R) * full);
    float* o = a.alloc(size_t(R) * ld);
    WS_PTR(fr && y && o);
    ws_gemm_nt_args g = {};
    g.A = est4, g.W = n.syn4, g.C = fr;
    g.a_div = kBig, g.a_s2 = ld4, g.c_div = kBig, g.c_s2 = nf, g.st_div1 = 1, g.st_div2 = 1;
    g.M = R * Tf, g.N = nf, g.K = ld4, g.ldw = ld4, g.vec = 3;
    WS_RUN(e, ws_gemm_nt(&g, s));
    WS_RUN(e, ws_ola_fwd(fr, nullptr, R, Tf, nf, hop, full, y, s));
    std::vector<double> env(size_t(Tf - 1) * hop + nf, 0.0);
    const double pi = 3.14159265358979323846;
    for (int k = 0; k < nf; ++k) {
      const float wf = static_cast<float>(0.5 - 0.5 * cos(2.0 * pi * k / nf));
      const double w2 = double(wf) * double(wf);
      for (int t = 0; t < Tf; ++t) env[size_t(t) * hop + k] += w2;
    }
    std::vector<float> inv(ld, 0.f);
    for (int i = 0; i < T; ++i) inv[i] = static_cast<float>(1.0 / env[size_t(pad) + i]);
    float* dinv = upload(e, a, inv.data(), inv.size());
    WS_PTR(dinv);
    if ((rc = zero_device(e, o, size_t(R) * ld * 4)) != WS_OK) return rc;
    if ((rc = copy_cols(e, o, ld, y + pad, full, T, R)) != WS_OK) return rc;
    WS_RUN(e, ws_affine_fwd(o, dinv, nullptr, 0.0f, R, R, ld, o, s));
    if ((rc = copy_cols(e, est, T, o, ld, T, R)) != WS_OK) return rc;
  }
  return WS_OK;
}

int check_engine(const ws_engine* e, const char* who) {
  if (!e) {
    set_err("%s: null engine", who);
    return WS_ERR_INVALID;
  }
  return WS_OK;
}

}  // namespace

// host-facing forward of a Conv-TasNet engine (same contract as ws_engine_separate)
static int tasnet_separate(ws_engine* e, const float* mix, int R, int T, const void* enroll, int enroll_kind, int enroll_len,
                           float* est) {
  const int L = e->tL, stride = L / 2;
  if (!mix || !enroll || !est || R < 1 || T < 160 || (long long)R * ((T - L) / stride + 1) * 3 * e->tN > 0x7fffffffLL) {
    set_err("ws_engine_separate: bad arguments (R=%d, T=%d; Conv-TasNet needs T >= 160)", R, T);
    return WS_ERR_INVALID;
  }
  const bool want_wave = e->joint != 0;
  if ((enroll_kind == WS_ENROLL_WAVE) != want_wave || (enroll_kind != WS_ENROLL_WAVE && enroll_kind != WS_ENROLL_EMBEDDING)) {
    set_err("ws_engine_separate: a Conv-TasNet engine takes %s (got enrollment kind %d)",
            want_wave ? "the enrollment waveform (SpEx+ speaker encoder on the shared encoder)" : "fixed embeddings", enroll_kind);
    return WS_ERR_INVALID;
  }
  if (want_wave && ((enroll_len - L) / stride + 1) / 27 < 1) {
    set_err("ws_engine_separate: enrollment of %d samples is too short for three MaxPool1d(3) stages", enroll_len);
    return WS_ERR_INVALID;
  }
  if (!e->dry && hipSetDevice(e->device) != hipSuccess) {
    set_err("ws_engine_separate: hipSetDevice(%d) failed", e->device);
    return WS_ERR_LAUNCH;
  }
  e->n_launches = 0;
  Arena& a = e->work;
  a.reset();
  int rc;
  float* d_mix = a.alloc(size_t(R) * T);
  float* d_est = a.alloc(size_t(R) * T);
  float* d_enr = a.alloc(want_wave ? size_t(R) * enroll_len : size_t(R) * e->E);
  WS_PTR(d_mix && d_est && d_enr);
  if ((rc = to_device(e, d_mix, mix, size_t(R) * T * 4)) != WS_OK) return rc;
  if ((rc = to_device(e, d_enr, enroll, (want_wave ? size_t(R) * enroll_len : size_t(R) * e->E) * 4)) != WS_OK) return rc;
  if ((rc = tasnet_device(e, d_mix, R, T, want_wave ? nullptr : d_enr, want_wave ? d_enr : nullptr, enroll_len, d_est)) != WS_OK)
    return rc;
  if ((rc = to_host(e, est, d_est, size_t(R) * T * 4)) != WS_OK) return rc;
  a.reset();
  a.consolidate();
  return WS_OK;
}

// ---- C ABI ----------------------------------------------------------------------------------------------------
extern "C" int ws_engine_abi_version(void) { return WS_ENGINE_ABI_VERSION; }
extern "C" const char* ws_engine_last_error(void) { return g_err; }

extern "C" void ws_engine_destroy(ws_engine* e) {
  if (!e) return;
  grid_free(e->grid);
  e->work.free_all();
  e->persist.free_all();
  if (e->stream) (void)hipStreamDestroy(e->stream);
  delete e;
}

extern "C" int ws_engine_create(const char* weights_path, int device, int flags, ws_engine** out) {
  if (!weights_path || !out) {
    set_err("ws_engine_create: null argument");
    return WS_ERR_INVALID;
  }
  *out = nullptr;
  if (ws_abi_version() != WS_ABI_VERSION) {
    set_err("ws_engine_create: libwesep_hip.so ABI %d, engine built for %d", ws_abi_version(), WS_ABI_VERSION);
    return WS_ERR_INVALID;
  }
  ws_engine* e = new ws_engine();
  e->dry = (flags & WS_ENGINE_DRY_RUN) != 0;
  e->device = device;
  e->persist.dry = e->work.dry = e->dry;
  int rc = load_container(e, weights_path);
  if (rc == WS_OK && e->dry) {
    // with a device present the launches of a dry run would really execute -- on host pointers
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0) {
      set_err("ws_engine_create: WS_ENGINE_DRY_RUN is for machines without a GPU (%d HIP device(s) visible)", ndev);
      rc = WS_ERR_INVALID;
    }
  }
  if (rc == WS_OK && !e->dry) {
    hipDeviceProp_t prop;
    if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&prop, device) != hipSuccess ||
        hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess) {
      set_err("ws_engine_create: no usable HIP device %d (use WS_ENGINE_DRY_RUN to validate without a GPU)", device);
      rc = WS_ERR_LAUNCH;
    } else {
      e->cu_count = prop.multiProcessorCount;
    }
  }
  if (rc == WS_OK) rc = prepare(e);
  if (rc != WS_OK) {
    ws_engine_destroy(e);
    return rc;
  }
  *out = e;
  return WS_OK;
}

extern "C" long long ws_engine_info(const ws_engine* e, const char* key) {
  if (!e || !key) return -1;
  const std::string k(key);
  if (k == "n_tensors") return static_cast<long long>(e->tensors.size());
  if (k == "n_launches") return e->n_launches;
  if (k == "arena_bytes") return static_cast<long long>(e->work.peak_bytes);
  if (k == "cluster_fallbacks") return e->cluster_fallbacks;
  if (k == "nband") return e->K;
  if (k == "arch") return e->arch;
  auto it = e->meta.find(k);
  return it == e->meta.end() ? -1 : it->second;
}

extern "C" int ws_engine_separate(ws_engine* e, const float* mix, int R, int T, const void* enroll, int enroll_kind,
                                  int enroll_len, float* est) {
  int rc = check_engine(e, "ws_engine_separate");
  if (rc != WS_OK) return rc;
  if (e->arch == 1) return tasnet_separate(e, mix, R, T, enroll, enroll_kind, enroll_len, est);
  if (!mix || !enroll || !est || R < 1 || T < 512 || (long long)R * (1 + T / kHop) * 4 * kNBin > 0x7fffffffLL) {
    set_err("ws_engine_separate: bad arguments (R=%d, T=%d; T >= 512)", R, T);
    return WS_ERR_INVALID;
  }
  if (e->arch == 3) {
    const GridNet& gn = *e->grid;
    // (any sample count: the standard-deviation scaling that ties the Python path to multiples of 4 samples runs on the host here)
    if (T < 2 * gn.n_fft || (long long)R * (1 + T / gn.hop) * gn.Q * (2 * gn.nh * gn.E + gn.C) > 0x7fffffffLL) {
      set_err("ws_engine_separate: a TF-GridNet engine needs T >= %d and R * frames * bins * %d below 2^31 (R=%d, T=%d)",
              2 * gn.n_fft, 2 * gn.nh * gn.E + gn.C, R, T);
      return WS_ERR_INVALID;
    }
  }
  if (e->arch == 2 && (T < 31 * kHop || (long long)R * (1 + T / kHop) * kDpBins * 160 > 0x7fffffffLL)) {
    set_err("ws_engine_separate: a DPCCN engine needs T >= %d samples (32 frames for the AvgPool2d(32) branch) and "
            "R * frames * 257 * 160 below 2^31 (R=%d, T=%d)", 31 * kHop, R, T);
    return WS_ERR_INVALID;
  }
  if ((enroll_kind == WS_ENROLL_EMBEDDING) == (e->joint != 0) || enroll_kind < 0 || enroll_kind > WS_ENROLL_WAVE) {
    set_err("ws_engine_separate: enrollment kind %d does not fit this model (joint_training = %d)", enroll_kind, e->joint);
    return WS_ERR_INVALID;
  }
  int Te = enroll_len;
  if (enroll_kind == WS_ENROLL_WAVE && e->spk_feat) {          // kaldi fbank, snip-edges framing
    if (enroll_len < e->fb_win) {
      set_err("ws_engine_separate: enrollment shorter than one %d-sample frame", e->fb_win);
      return WS_ERR_INVALID;
    }
    Te = 1 + (enroll_len - e->fb_win) / e->fb_shift;
  } else if (enroll_kind == WS_ENROLL_WAVE) {                  // in-model MelSpectrogram, centred framing
    if (enroll_len <= 256) {
      set_err("ws_engine_separate: enrollment must be longer than the 256-sample reflect padding");
      return WS_ERR_INVALID;
    }
    Te = 1 + enroll_len / kHop;
  } else if (enroll_kind == WS_ENROLL_FBANK && !e->spk_feat) {
    set_err("ws_engine_separate: this model computes its own features (spk_feat = False): pass the waveform");
    return WS_ERR_INVALID;
  }
  if (enroll_kind != WS_ENROLL_EMBEDDING && Te < 8) {
    set_err("ws_engine_separate: enrollment of %d frames is too short for the speaker encoder", Te);
    return WS_ERR_INVALID;
  }
  if (!e->dry && hipSetDevice(e->device) != hipSuccess) {
    set_err("ws_engine_separate: hipSetDevice(%d) failed", e->device);
    return WS_ERR_LAUNCH;
  }
  // Engines that share a GPU overlap on the device.  Round 2 serialised them here (one forward at a time per GPU)
  // because ws_gemm_b2p / the grouped ws_gemm_nt / ws_gemm_tn disturbed this plan's STFT / iSTFT kernels on another
  // stream.  Round 3 named the victim class -- packed FP32 instructions with an operand selection -- and the library is
  // built without them (profiles/r03_kernel_race.md; tests/test_cross_stream_gpu.py), so the lock is opt-in:
  // WS_ENGINE_SERIALIZE=1 restores one forward at a time (e.g. beside third-party kernels on the same GPU).
  static const bool serialize = getenv("WS_ENGINE_SERIALIZE") != nullptr && atoi(getenv("WS_ENGINE_SERIALIZE")) != 0;
  std::unique_lock<std::mutex> device_turn(g_device_mutex[e->device & 15], std::defer_lock);
  if (serialize) device_turn.lock();
  e->n_launches = 0;
  Arena& a = e->work;
  a.reset();
  float* d_mix = a.alloc(size_t(R) * T);
  float* d_est = a.alloc(size_t(R) * T);
  float* d_emb = a.alloc(size_t(R) * e->E);
  WS_PTR(d_mix && d_est && d_emb);
  // TF-GridNet scales the mixture by its (unbiased) standard deviation and the estimate back (tfgridnet.py:222-226,292)
  std::vector<float> mixn, stds;
  if (e->arch == 3) {
    mixn.resize(size_t(R) * T);
    stds.resize(R);
    for (int r = 0; r < R; ++r) {
      const float* x = mix + size_t(r) * T;
      double m = 0.0, v = 0.0;
      for (int i = 0; i < T; ++i) m += x[i];
      m /= T;
      for (int i = 0; i < T; ++i) v += (x[i] - m) * (x[i] - m);
      stds[r] = static_cast<float>(sqrt(v / (T - 1.0)));
      const float inv = 1.0f / stds[r];
      for (int i = 0; i < T; ++i) mixn[size_t(r) * T + i] = x[i] * inv;
    }
    mix = mixn.data();
  }
  if ((rc = to_device(e, d_mix, mix, size_t(R) * T * 4)) != WS_OK) return rc;
  if (enroll_kind == WS_ENROLL_EMBEDDING) {
    if ((rc = to_device(e, d_emb, enroll, size_t(R) * e->E * 4)) != WS_OK) return rc;
  } else {
    const Arena::Mark mk = a.mark();
    float* fb = a.alloc(size_t(R) * Te * e->feat_dim);
    WS_PTR(fb);
    if (enroll_kind == WS_ENROLL_FBANK) {
      if ((rc = to_device(e, fb, enroll, size_t(R) * Te * e->feat_dim * 4)) != WS_OK) return rc;
    } else {
      float* d_wave = a.alloc(size_t(R) * enroll_len);
      WS_PTR(d_wave);
      if ((rc = to_device(e, d_wave, enroll, size_t(R) * enroll_len * 4)) != WS_OK) return rc;
      if ((rc = e->spk_feat ? kaldi_fbank(e, d_wave, R, enroll_len, fb, Te) : mel_frontend(e, d_wave, R, enroll_len, fb, Te)) != WS_OK)
        return rc;
    }
    if ((rc = e->spk_kind == 2 ? campplus_embed(e, fb, R, Te, d_emb)
                               : e->spk_kind == 1 ? ecapa_embed(e, fb, R, Te, d_emb) : resnet_embed(e, fb, R, Te, d_emb)) != WS_OK)
      return rc;
    a.release(mk);
  }
  rc = e->arch == 2 ? dpccn_device(e, d_mix, R, T, d_emb, d_est)
                    : e->arch == 3 ? gridnet_device(e, d_mix, R, T, d_emb, d_est) : separate_device(e, d_mix, R, T, d_emb, d_est);
  if (rc != WS_OK) return rc;
  if ((rc = to_host(e, est, d_est, size_t(R) * T * 4)) != WS_OK) return rc;
  if (e->arch == 3 && !e->dry)
    for (int r = 0; r < R; ++r)
      for (int i = 0; i < T; ++i) est[size_t(r) * T + i] *= stds[r];
  if (e->cl_status && !e->dry) {   // did a cluster recurrence time out (and the predicated streaming pair repair it)?
    unsigned st = 0;
    if ((rc = to_host(e, &st, e->cl_status, 4)) != WS_OK) return rc;
    if (st) {
      ++e->cluster_fallbacks;
      if ((rc = zero_device(e, e->cl_status, 4)) != WS_OK) return rc;
    }
  }
  a.reset();
  a.consolidate();
  return WS_OK;
}

extern "C" int ws_engine_forward_pcm16(ws_engine* e, const int16_t* mix, int n, const int16_t* spk1, const int16_t* spk2,
                                       int n_enroll, float* out) {
  int rc = check_engine(e, "ws_engine_forward_pcm16");
  if (rc != WS_OK) return rc;
  if (!mix || !spk1 || !spk2 || !out || n < 512 || n_enroll < 1) {
    set_err("ws_engine_forward_pcm16: bad arguments");
    return WS_ERR_INVALID;
  }
  // separate_engine.cc:78-98: the mixture twice (one row per enrollment), scaled to [-1, 1]; the enrollment fbank is
  // computed on int16-valued samples, which the folded basis' 2^15 factor reproduces from the [-1, 1] rows
  std::vector<float> m(size_t(2) * n), enr(size_t(2) * n_enroll);
  for (int i = 0; i < n; ++i) m[i] = m[size_t(n) + i] = static_cast<float>(mix[i]) / 32768.0f;
  for (int i = 0; i < n_enroll; ++i) {
    enr[i] = static_cast<float>(spk1[i]) / 32768.0f;
    enr[size_t(n_enroll) + i] = static_cast<float>(spk2[i]) / 32768.0f;
  }
  return ws_engine_separate(e, m.data(), 2, n, enr.data(), WS_ENROLL_WAVE, n_enroll, out);
}
