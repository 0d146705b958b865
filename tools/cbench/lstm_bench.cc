// lstm_bench -- Python-free microbenchmark of the BLSTM recurrences through the C ABI (include/wesep_hip.h).
// A GPU call that imports torch spends 1-2 minutes before the first kernel; this binary starts in milliseconds, so a
// gpurun call can sweep many configurations inside a few GPU-seconds.  Not part of the product.
//
//   lstm_bench [--view time|band] [--rows 32] [--seconds 4] [--iters 5] [--what fwd,bwd,fused,cluster,cluster_bwd,pair]
//
// Prints one line per (kernel, configuration): ms per launch, microseconds per recurrence step, and the algorithmic
// HBM rate (bench.py's convention: 10 fp32 per position, direction and hidden unit) against the 8 TB/s peak.
// Also the tanh-bounded checksum of h / d(gates) so that NaNs or an all-zero output are visible, and -- with
// `--compare 1` -- the relative difference of h and d(gates) between the 16-sequence, 32-sequence and cluster kernels
// on the same inputs (they share the blocked layout), a device-side parity check for kernel experiments.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <random>
#include <string>
#include <vector>

#include "../../include/wesep_hip.h"

#define HIP_OK(x)                                                              \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_));           \
      exit(2);                                                                 \
    }                                                                          \
  } while (0)
#define WS_OK_(x)                                                              \
  do {                                                                         \
    int r_ = (x);                                                              \
    if (r_ != WS_OK) {                                                         \
      fprintf(stderr, "%s failed (rc=%d): %s\n", #x, r_, ws_last_error());     \
      exit(3);                                                                 \
    }                                                                          \
  } while (0)

static float* dalloc(size_t n) {
  float* p = nullptr;
  HIP_OK(hipMalloc(reinterpret_cast<void**>(&p), n * 4));
  return p;
}

static float* drandom(size_t n, float scale, unsigned seed) {
  // a 1 M-sample Gaussian block tiled over the buffer: big buffers (GBs of gate pre-activations) in milliseconds
  const size_t blk = n < (1u << 20) ? n : (1u << 20);
  std::vector<float> h(blk);
  std::mt19937 rng(seed);
  std::normal_distribution<float> nd(0.f, scale);
  for (auto& v : h) v = nd(rng);
  float* p = dalloc(n);
  for (size_t o = 0; o < n; o += blk)
    HIP_OK(hipMemcpy(p + o, h.data(), (n - o < blk ? n - o : blk) * 4, hipMemcpyHostToDevice));
  return p;
}

static std::vector<float> fetch(const float* d, size_t n) {
  std::vector<float> h(n);
  HIP_OK(hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost));
  return h;
}

static double rel_diff(const std::vector<float>& a, const std::vector<float>& b) {
  double num = 0.0, den = 0.0;
  for (size_t i = 0; i < a.size(); ++i) {
    const double d = double(a[i]) - b[i];
    num += d * d;
    den += double(b[i]) * b[i];
  }
  return sqrt(num / (den + 1e-300));
}

static double checksum(const float* d, size_t n) {
  std::vector<float> h(n > (1u << 22) ? (1u << 22) : n);
  HIP_OK(hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost));
  double s = 0.0;
  for (float v : h) s += fabs(static_cast<double>(v));
  return s / h.size();
}

template <class F>
static double time_ms(F&& launch, int iters, hipStream_t s) {
  hipEvent_t a, b;
  HIP_OK(hipEventCreate(&a));
  HIP_OK(hipEventCreate(&b));
  launch();
  HIP_OK(hipStreamSynchronize(s));
  HIP_OK(hipEventRecord(a, s));
  for (int i = 0; i < iters; ++i) launch();
  HIP_OK(hipEventRecord(b, s));
  HIP_OK(hipEventSynchronize(b));
  float ms = 0.f;
  HIP_OK(hipEventElapsedTime(&ms, a, b));
  return ms / iters;
}

int main(int argc, char** argv) {
  std::map<std::string, std::string> kv;
  for (int i = 1; i + 1 < argc; i += 2) kv[argv[i]] = argv[i + 1];
  auto get = [&](const char* k, const char* d) { return kv.count(k) ? kv[k] : std::string(d); };
  const std::string view = get("--view", "time"), what = get("--what", "fwd,bwd");
  const int R = atoi(get("--rows", "32").c_str()), iters = atoi(get("--iters", "5").c_str());
  const int T = static_cast<int>(16000 * atof(get("--seconds", "4").c_str()));
  const int K = 32, Tf = 1 + T / 128, H = WS_LSTM_H, G4 = 4 * H, N = 128;
  const int nseq = view == "time" ? R * K : R * Tf, L = view == "time" ? Tf : K;
  const int ntile = (nseq + 31) / 32;
  const size_t nb = size_t(ntile) * L, P = size_t(nseq) * L;
  const int mode_default = 2 * ntile <= 128 ? WS_LSTM_BF16X3_BLK16 : WS_LSTM_BF16X3_BLK;
  const int mode = kv.count("--mode") ? atoi(kv["--mode"].c_str()) : mode_default;
  hipStream_t s;
  HIP_OK(hipStreamCreate(&s));
  hipDeviceProp_t prop;
  HIP_OK(hipGetDeviceProperties(&prop, 0));
  printf("# %s, %d CUs; view %s: %d sequences x %d steps (%d tiles), mode %d\n", prop.name, prop.multiProcessorCount,
         view.c_str(), nseq, L, ntile, mode);

  float* whf = drandom(size_t(G4) * H, 0.06f, 1);
  float* whr = drandom(size_t(G4) * H, 0.06f, 2);
  float* wif = drandom(size_t(G4) * N, 0.08f, 3);
  float* wir = drandom(size_t(G4) * N, 0.08f, 4);
  float* bias = drandom(2 * G4, 0.05f, 5);
  float* pack_f = dalloc(WS_LSTM_PACK_FLOATS);
  float* pack_b = dalloc(WS_LSTM_PACK_FLOATS);
  float* fpack = dalloc(WS_LSTM_FUSED_PACK_FLOATS);
  float* gates0 = drandom(nb * 32 * 2 * G4, 1.0f, 6);     // pre-activations (x W_ih^T + b)
  float* gates = dalloc(nb * 32 * 2 * G4);
  float* cbuf = dalloc(nb * 32 * 2 * H);
  float* hcat = dalloc(nb * 32 * 2 * H);
  float* dh = drandom(nb * 32 * 2 * H, 1e-3f, 7);
  float* xn = drandom(nb * 32 * N, 1.0f, 8);
  const size_t gbytes = nb * 32 * 2 * G4 * 4;
  WS_OK_(ws_lstm_pack(whf, whr, pack_f, pack_b, mode, s));
  WS_OK_(ws_lstm_pack_fused(wif, wir, whf, whr, fpack, s));

  ws_lstm_args a = {};
  a.gates = gates, a.cbuf = cbuf, a.hcat = hcat, a.nseq = nseq, a.L = L, a.mode = mode, a.sq_div = 1 << 30;
  const double bytes = 10.0 * 4 * double(P) * 2 * H;
  auto report = [&](const char* name, double ms, const float* out, size_t n) {
    printf("%-12s %8.3f ms/launch  %6.2f us/step  %7.1f GB/s algorithmic (%.1f%% of 8 TB/s)  mean|out| %.4e\n", name, ms,
           1e3 * ms / L, bytes / ms * 1e-6, 100.0 * bytes / ms * 1e-6 / 8000.0, checksum(out, n));
  };
  auto reset_gates = [&]() { HIP_OK(hipMemcpyAsync(gates, gates0, gbytes, hipMemcpyDeviceToDevice, s)); };

  // forward first (the backward consumes its activated gates / cells / h)
  reset_gates();
  a.wpack = pack_f;
  if (what.find("fwd") != std::string::npos) {
    // timing re-runs the forward on already activated gates: same memory traffic and arithmetic, values saturate
    const double ms = time_ms([&] { WS_OK_(ws_lstm_fwd(&a, s)); }, iters, s);
    report("fwd", ms, hcat, nb * 32 * 2 * H);
  }
  if (what.find("fused") != std::string::npos && mode == WS_LSTM_BF16X3_BLK) {
    ws_lstm_fused_args f = {};
    f.gates = gates, f.cbuf = cbuf, f.hcat = hcat, f.xn = xn, f.wpack = fpack, f.bias = bias, f.nseq = nseq, f.L = L;
    // both tilings of the fused forward on the same inputs: timing and a bit-for-bit comparison of everything they write
    std::vector<float> ref_g, ref_c, ref_h;
    for (const char* seqs : {"32", "64"}) {
      setenv("WS_FUSED_SEQS", seqs, 1);
      HIP_OK(hipMemsetAsync(gates, 0xff, gbytes, s));
      HIP_OK(hipMemsetAsync(cbuf, 0xff, nb * 32 * 2 * H * 4, s));
      HIP_OK(hipMemsetAsync(hcat, 0xff, nb * 32 * 2 * H * 4, s));
      const double ms = time_ms([&] { WS_OK_(ws_lstm_fwd_fused(&f, s)); }, iters, s);
      report((std::string("fwd_fused/") + seqs).c_str(), ms, hcat, nb * 32 * 2 * H);
      if (ref_g.empty()) {
        ref_g = fetch(gates, nb * 32 * 2 * G4), ref_c = fetch(cbuf, nb * 32 * 2 * H), ref_h = fetch(hcat, nb * 32 * 2 * H);
      } else {
        const std::vector<float> g2 = fetch(gates, nb * 32 * 2 * G4), c2 = fetch(cbuf, nb * 32 * 2 * H),
                                 h2 = fetch(hcat, nb * 32 * 2 * H);
        printf("fused 64 vs 32 sequences per workgroup: gates %s, cells %s, h %s\n",
               memcmp(g2.data(), ref_g.data(), g2.size() * 4) ? "DIFFER" : "bit-identical",
               memcmp(c2.data(), ref_c.data(), c2.size() * 4) ? "DIFFER" : "bit-identical",
               memcmp(h2.data(), ref_h.data(), h2.size() * 4) ? "DIFFER" : "bit-identical");
      }
    }
    unsetenv("WS_FUSED_SEQS");
  }
  const bool cluster_ok = nseq % 64 == 0 && (nseq / 32) * 8 <= prop.multiProcessorCount && L >= 64;
  float* xchg = nullptr;
  unsigned* flags = nullptr;
  if (cluster_ok && what.find("cluster") != std::string::npos) {
    xchg = dalloc(size_t(nseq / 32) * 2 * 64 * 8192 / 4);
    flags = reinterpret_cast<unsigned*>(dalloc(size_t(nseq / 32) * 8 + 8));
    ws_lstm_cluster_args c = {};
    c.gates = gates, c.cbuf = cbuf, c.hcat = hcat, c.whh_f = whf, c.whh_r = whr, c.xchg = xchg, c.flags = flags;
    c.nseq = nseq, c.L = L;
    reset_gates();
    const double ms = time_ms([&] { WS_OK_(ws_lstm_fwd_cluster(&c, s)); }, iters, s);
    report("fwd_cluster", ms, hcat, nb * 32 * 2 * H);
  }
  // a clean forward, then the backward passes
  reset_gates();
  WS_OK_(ws_lstm_fwd(&a, s));
  float* gates_act = dalloc(nb * 32 * 2 * G4);   // activated gates: the backward overwrites them with d(gates)
  HIP_OK(hipMemcpyAsync(gates_act, gates, gbytes, hipMemcpyDeviceToDevice, s));
  if (what.find("bwd") != std::string::npos) {
    a.wpack = pack_b;
    a.dhcat = dh;
    HIP_OK(hipMemcpyAsync(gates, gates_act, gbytes, hipMemcpyDeviceToDevice, s));
    const double ms = time_ms([&] { WS_OK_(ws_lstm_bwd(&a, s)); }, iters, s);
    report("bwd", ms, gates, nb * 32 * 2 * G4);
  }
  if (cluster_ok && what.find("cluster_bwd") != std::string::npos) {
    ws_lstm_cluster_args c = {};
    c.gates = gates, c.cbuf = cbuf, c.dhcat = dh, c.whh_f = whf, c.whh_r = whr, c.xchg = xchg, c.flags = flags;
    c.nseq = nseq, c.L = L;
    HIP_OK(hipMemcpyAsync(gates, gates_act, gbytes, hipMemcpyDeviceToDevice, s));
    const double ms = time_ms([&] { WS_OK_(ws_lstm_bwd_cluster(&c, s)); }, iters, s);
    report("bwd_cluster", ms, gates, nb * 32 * 2 * G4);
  }
  const int npair = 2 * ntile;
  const bool pair_ok = 2 * npair <= prop.multiProcessorCount;
  float* pack_p = nullptr;
  float* xchg_p = nullptr;
  unsigned* flags_p = nullptr;
  unsigned* status_p = nullptr;
  if (pair_ok) {
    pack_p = dalloc(WS_LSTM_PACK_FLOATS);
    xchg_p = dalloc(size_t(npair) * 65536 / 4);
    flags_p = reinterpret_cast<unsigned*>(dalloc(size_t(npair) * 8 + 8));
    status_p = reinterpret_cast<unsigned*>(dalloc(1));
    HIP_OK(hipMemsetAsync(status_p, 0, 4, s));
    WS_OK_(ws_lstm_pack_pair(whf, whr, pack_p, s));
  }
  auto pair_args = [&](int dbg) {
    ws_lstm_pair_args c = {};
    c.gates = gates, c.cbuf = cbuf, c.dhcat = dh, c.wpack = pack_p, c.xchg = xchg_p, c.flags = flags_p;
    c.status = status_p, c.nseq = nseq, c.L = L, c.dbg = dbg;
    return c;
  };
  if (pair_ok && what.find("pair") != std::string::npos) {
    // dbg variants are timing probes (their results are wrong by construction): 1 no flag wait, 2 no exchange,
    // 4 no lo-plane reloads, 32 no wave priorities
    for (int dbg : {0, 2048, 1, 2}) {
      ws_lstm_pair_args c = pair_args(dbg);
      HIP_OK(hipMemcpyAsync(gates, gates_act, gbytes, hipMemcpyDeviceToDevice, s));
      const double ms = time_ms([&] { WS_OK_(ws_lstm_bwd_pair(&c, s)); }, iters, s);
      char name[32];
      snprintf(name, sizeof name, "bwd_pair/%d", dbg);
      report(name, ms, gates, nb * 32 * 2 * G4);
    }
    unsigned st = 0;
    HIP_OK(hipMemcpy(&st, status_p, 4, hipMemcpyDeviceToHost));
    printf("bwd_pair status word after the runs: %u\n", st);
  }
  if (atoi(get("--compare", "0").c_str())) {
    // forward and backward of every applicable kernel family on identical inputs
    // the first 64 M elements are plenty for a parity signal (the band view's d(gates) is 1 G floats)
    const size_t cap = size_t(1) << 26;
    const size_t nh = nb * 32 * 2 * H < cap ? nb * 32 * 2 * H : cap, ng = nb * 32 * 2 * G4 < cap ? nb * 32 * 2 * G4 : cap;
    std::vector<std::vector<float>> hs, gs;
    std::vector<std::string> names;
    for (int m : {WS_LSTM_BF16X3_BLK16, WS_LSTM_BF16X3_BLK}) {
      WS_OK_(ws_lstm_pack(whf, whr, pack_f, pack_b, m, s));
      ws_lstm_args c = a;
      c.mode = m, c.dhcat = dh;
      reset_gates();
      c.wpack = pack_f;
      WS_OK_(ws_lstm_fwd(&c, s));
      HIP_OK(hipStreamSynchronize(s));
      hs.push_back(fetch(hcat, nh));
      c.wpack = pack_b;
      WS_OK_(ws_lstm_bwd(&c, s));
      HIP_OK(hipStreamSynchronize(s));
      gs.push_back(fetch(gates, ng));
      names.push_back(m == WS_LSTM_BF16X3_BLK16 ? "blk16" : "blk32");
    }
    if (cluster_ok) {
      if (!xchg) {
        xchg = dalloc(size_t(nseq / 32) * 2 * 64 * 8192 / 4);
        flags = reinterpret_cast<unsigned*>(dalloc(size_t(nseq / 32) * 8 + 8));
      }
      ws_lstm_cluster_args c = {};
      c.gates = gates, c.cbuf = cbuf, c.hcat = hcat, c.dhcat = dh, c.whh_f = whf, c.whh_r = whr, c.xchg = xchg;
      c.flags = flags, c.nseq = nseq, c.L = L;
      reset_gates();
      WS_OK_(ws_lstm_fwd_cluster(&c, s));
      HIP_OK(hipStreamSynchronize(s));
      hs.push_back(fetch(hcat, nh));
      WS_OK_(ws_lstm_bwd_cluster(&c, s));
      HIP_OK(hipStreamSynchronize(s));
      gs.push_back(fetch(gates, ng));
      names.push_back("cluster");
    }
    if (pair_ok) {
      // same forward state as the blk16 run: redo that forward, then the pair BPTT
      WS_OK_(ws_lstm_pack(whf, whr, pack_f, pack_b, WS_LSTM_BF16X3_BLK16, s));
      ws_lstm_args c = a;
      c.mode = WS_LSTM_BF16X3_BLK16, c.wpack = pack_f;
      reset_gates();
      WS_OK_(ws_lstm_fwd(&c, s));
      HIP_OK(hipStreamSynchronize(s));
      hs.push_back(fetch(hcat, nh));
      for (int rep = 0; rep < 2; ++rep) {
        if (rep) {
          reset_gates();
          WS_OK_(ws_lstm_fwd(&c, s));
        }
        ws_lstm_pair_args pa = pair_args(0);
        WS_OK_(ws_lstm_bwd_pair(&pa, s));
        HIP_OK(hipStreamSynchronize(s));
        gs.push_back(fetch(gates, ng));
      }
      const bool same = memcmp(gs[gs.size() - 1].data(), gs[gs.size() - 2].data(), ng * 4) == 0;
      gs.pop_back();
      names.push_back("pair");
      printf("pair run-to-run identical: %s\n", same ? "yes" : "NO");
    }
    for (size_t i = 1; i < names.size(); ++i)
      printf("compare %-8s vs %s:  h rel %.3e   d(gates) rel %.3e\n", names[i].c_str(), names[0].c_str(),
             rel_diff(hs[i], hs[0]), rel_diff(gs[i], gs[0]));
  }
  HIP_OK(hipStreamSynchronize(s));
  return 0;
}
