// Host-side runtime of libwesep_hip.so: error reporting + HIP-event kernel profiler.
#include <stdarg.h>
#include <stdio.h>

#include <mutex>
#include <vector>

#include "common.h"

static thread_local char g_err[512] = "";

void ws_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int ws_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    ws_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return WS_ERR_LAUNCH;
  }
  return WS_OK;
}

extern "C" const char* ws_last_error(void) { return g_err; }
extern "C" int ws_abi_version(void) { return WS_ABI_VERSION; }

// ---- profiler ---------------------------------------------------------------------------
// When enabled, every launch of a profiled kind is bracketed by two hipEvents recorded on the
// launch stream; ws_prof_collect() synchronises them and returns the summed elapsed time.
namespace {
struct Pair {
  hipEvent_t a, b;
};
struct Prof {
  std::mutex mu;
  bool on = false;
  std::vector<Pair> used[WS_PROF_NKINDS];
  std::vector<Pair> pool;
  hipEvent_t open[WS_PROF_NKINDS] = {};
  hipEvent_t open_b[WS_PROF_NKINDS] = {};
} g_prof;
}  // namespace

extern "C" int ws_prof_enable(int on) {
  std::lock_guard<std::mutex> l(g_prof.mu);
  g_prof.on = on != 0;
  return WS_OK;
}

void ws_prof_begin(int kind, hipStream_t s) {
  if (!g_prof.on) return;
  std::lock_guard<std::mutex> l(g_prof.mu);
  Pair p;
  if (!g_prof.pool.empty()) {
    p = g_prof.pool.back();
    g_prof.pool.pop_back();
  } else {
    (void)hipEventCreate(&p.a);
    (void)hipEventCreate(&p.b);
  }
  (void)hipEventRecord(p.a, s);
  g_prof.open[kind] = p.a;
  g_prof.open_b[kind] = p.b;
}

void ws_prof_end(int kind, hipStream_t s) {
  if (!g_prof.on) return;
  std::lock_guard<std::mutex> l(g_prof.mu);
  if (!g_prof.open[kind]) return;
  (void)hipEventRecord(g_prof.open_b[kind], s);
  g_prof.used[kind].push_back(Pair{g_prof.open[kind], g_prof.open_b[kind]});
  g_prof.open[kind] = nullptr;
}

extern "C" int ws_prof_collect(int kind, double* total_ms, long long* launches) {
  WS_REQUIRE(kind >= 0 && kind < WS_PROF_NKINDS && total_ms && launches, "ws_prof_collect: bad args");
  std::lock_guard<std::mutex> l(g_prof.mu);
  double tot = 0.0;
  for (auto& p : g_prof.used[kind]) {
    (void)hipEventSynchronize(p.b);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, p.a, p.b);
    tot += ms;
    g_prof.pool.push_back(p);
  }
  *total_ms = tot;
  *launches = (long long)g_prof.used[kind].size();
  g_prof.used[kind].clear();
  return WS_OK;
}

// ---- test support -------------------------------------------------------------------------
// Leaves `value` in 60 KB of LDS of every CU it lands on (and keeps its waves alive for a few microseconds).  Run on a
// second stream beside the product kernels it turns "a kernel read LDS it never wrote" from a silent dependence on
// whatever ran before into NaN -- the LDS counterpart of NaN-filling uninitialised global memory.
__global__ __launch_bounds__(256) void debug_dirty_lds_kernel(float value, int spins, float* sink) {
  __shared__ float buf[15360];
  for (int i = threadIdx.x; i < 15360; i += 256) buf[i] = value;
  __syncthreads();
  float acc = 0.f;
  for (int s = 0; s < spins; ++s) acc += buf[(threadIdx.x * 7 + s * 13) % 15360];
  if (sink && acc == 12345.678f) sink[0] = acc;   // never true for NaN / the test values: keeps the loop alive
}

extern "C" int ws_debug_dirty_lds(float value, int nblocks, int spins, float* sink, void* stream) {
  WS_REQUIRE(nblocks > 0 && spins >= 0, "ws_debug_dirty_lds: bad args");
  hipLaunchKernelGGL(debug_dirty_lds_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, value, spins, sink);
  return ws_check_launch("ws_debug_dirty_lds");
}

// Holds `nblocks` compute units for `usec` microseconds (or until *stop != 0): each workgroup takes 112 KB of LDS, so no
// product workgroup with a large LDS footprint (the pair BPTT: 145 KB, the cluster forward: 138 KB, the weight-gradient
// GEMM: 160 KB) can share its CU -- the shape of a resident collective kernel (RCCL keeps its channels' workgroups on
// their CUs for the whole all-reduce) as seen by the recurrences whose workgroups have to be co-resident.  The wait is on
// the constant-rate wall clock (100 MHz) and bounded by `usec`: it cannot hang the device.
__global__ __launch_bounds__(256) void debug_occupy_kernel(long long ticks, const unsigned* stop, float* sink) {
  __shared__ float buf[28672];  // 112 KB
  buf[threadIdx.x] = (float)threadIdx.x;
  __syncthreads();
  const long long t0 = wall_clock64();
  float acc = 0.f;
  while (wall_clock64() - t0 < ticks) {
    if (stop && __hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
    acc += buf[(threadIdx.x * 7) & 255];
    __builtin_amdgcn_s_sleep(32);
  }
  if (sink && acc == -1.f) sink[0] = acc;  // never true: keeps the LDS alive
}

extern "C" int ws_debug_occupy(int nblocks, int usec, const unsigned* stop, float* sink, void* stream) {
  WS_REQUIRE(nblocks > 0 && usec >= 0 && usec <= 2000000, "ws_debug_occupy: nblocks > 0, 0 <= usec <= 2 000 000");
  hipLaunchKernelGGL(debug_occupy_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, (long long)usec * 100LL, stop,
                     sink);
  return ws_check_launch("ws_debug_occupy");
}
