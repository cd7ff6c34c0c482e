// Error reporting for the C-ABI (thread-local message, no exceptions across the boundary).
#include "../../include/neurosis_hip.h"
#include "nk_common.h"
#include <stdio.h>

static thread_local char g_err[512] = "";

void nk_set_error(const char* file, int line, const char* what) {
  snprintf(g_err, sizeof(g_err), "%s:%d: %s", file, line, what);
}

int nk_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(g_err, sizeof(g_err), "%s: launch failed: %s", what, hipGetErrorString(e));
    return NK_ERR_LAUNCH;
  }
  return NK_OK;
}

extern "C" const char* nk_last_error(void) { return g_err; }
// 4: round 5 added nk_layernorm_bwd_rows / nk_colpart_reduce_batch / nk_layernorm_part_rows without a bump; round 6 adds none
// 5: round 6 adds the saved-derivative GEGLU entry points (nk_linear_fwd_geglu_s, nk_linear_dgrad_geglu_s, nk_geglu_fwd_s, nk_geglu_bwd_s)
extern "C" int nk_abi_version(void) { return 5; }

// ---- backward-health word ---------------------------------------------------------------------------------------------
#include <mutex>
static std::mutex g_health_mutex;
static unsigned* g_health_dev = nullptr;     // device word
static unsigned* g_health_host = nullptr;    // pinned mirror written by the snapshots
static hipEvent_t g_health_event;
static bool g_health_pending = false;        // a snapshot is in flight
static bool g_health_tripped = false;        // sticky until nk_health_clear()

unsigned* nk_health_word(void) {
  std::lock_guard<std::mutex> lock(g_health_mutex);
  if (!g_health_dev) {
    unsigned* d = nullptr;
    if (hipMalloc((void**)&d, 64) != hipSuccess) return nullptr;
    if (hipMemset(d, 0, 64) != hipSuccess || hipHostMalloc((void**)&g_health_host, 64, hipHostMallocDefault) != hipSuccess ||
        hipEventCreateWithFlags(&g_health_event, hipEventDisableTiming) != hipSuccess) {
      (void)hipFree(d);
      return nullptr;
    }
    g_health_host[0] = 0;
    g_health_dev = d;
  }
  return g_health_dev;
}

void nk_health_snapshot(hipStream_t stream) {
  if (!nk_health_word()) return;
  std::lock_guard<std::mutex> lock(g_health_mutex);
  if (g_health_pending && hipEventQuery(g_health_event) != hipSuccess) return;   // one snapshot in flight is enough
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) return;
  if (g_health_pending && g_health_host[0]) g_health_tripped = true;
  if (hipMemcpyAsync(g_health_host, g_health_dev, sizeof(unsigned), hipMemcpyDeviceToHost, stream) != hipSuccess) return;
  g_health_pending = hipEventRecord(g_health_event, stream) == hipSuccess;
}

int nk_health_poll(void) {
  std::lock_guard<std::mutex> lock(g_health_mutex);
  if (g_health_pending && hipEventQuery(g_health_event) == hipSuccess) {
    g_health_pending = false;
    if (g_health_host[0]) g_health_tripped = true;
  }
  if (g_health_tripped) {
    snprintf(g_err, sizeof(g_err), "backward health word is set: a stream-K GEMM gave up waiting for a partial tile in an earlier step (its output was "
                                   "poisoned with NaN and the optimizer skipped that update); inspect with nk_health_status(), reset with nk_health_clear()");
    return NK_ERR_HEALTH;
  }
  return NK_OK;
}

// 0 = healthy, 1 = raised.  Synchronises the device.
extern "C" int nk_health_status(void) {
  unsigned* d = nk_health_word();
  if (!d || hipDeviceSynchronize() != hipSuccess) return -1;
  unsigned v = 0;
  if (hipMemcpy(&v, d, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  std::lock_guard<std::mutex> lock(g_health_mutex);
  if (v) g_health_tripped = true;
  return (v || g_health_tripped) ? 1 : 0;
}

int nk_gemm_sk_reset(void);   // gemm.hip
extern "C" int nk_health_clear(void) {
  unsigned* d = nk_health_word();
  if (!d || hipDeviceSynchronize() != hipSuccess || hipMemset(d, 0, 64) != hipSuccess) return NK_ERR_LAUNCH;
  if (int e = nk_gemm_sk_reset()) return e;
  std::lock_guard<std::mutex> lock(g_health_mutex);
  g_health_pending = false;
  g_health_tripped = false;
  g_health_host[0] = 0;
  return NK_OK;
}

__global__ void nk_raise_health_kernel(unsigned* w) { __hip_atomic_store(w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// test hook: raises the word the way a kernel would (stream-ordered)
extern "C" int nk_debug_raise_health(void* stream) {
  unsigned* d = nk_health_word();
  if (!d) { nk_set_error(__FILE__, __LINE__, "health word allocation failed"); return NK_ERR_LAUNCH; }
  hipLaunchKernelGGL(nk_raise_health_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, d);
  return nk_check_launch("nk_raise_health_kernel");
}

// diagnostic: a stream-ordered timestamp (the 100 MHz constant clock, s_memrealtime) into a caller-owned device word -- capturable into a
// hipGraph, which HIP events are not: tools/step_timeline.py stamps the replayed segments of the backward with it, UNTRACED (the kernel trace
// perturbs how two streams overlap)
__global__ void nk_stamp_kernel(unsigned long long* dst) { *dst = __builtin_amdgcn_s_memrealtime(); }
extern "C" int nk_debug_stamp(unsigned long long* dst, void* stream) {
  NK_CHECK_ARG(dst != nullptr);
  hipLaunchKernelGGL(nk_stamp_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, dst);
  return nk_check_launch("nk_stamp_kernel");
}

// ---- the health word across data-parallel ranks: export this rank's word into a caller-owned int32 (the caller reduces it over the ranks with
// MAX, on the exchange stream), import the result back (OR).  A rank whose stream-K fix-up gave up poisons its gradient tile with NaN and the
// all-reduce spreads that NaN to every rank: with the word merged, every rank's optimizer kernels skip the update, not only the flagged one's.
__global__ void nk_health_export_kernel(const unsigned* w, int* dst) { *dst = (int)__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__global__ void nk_health_import_kernel(unsigned* w, const int* src) {
  if (*src) __hip_atomic_store(w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
extern "C" int nk_health_export(int* dst, void* stream) {
  unsigned* d = nk_health_word();
  NK_CHECK_ARG(dst != nullptr);
  if (!d) { nk_set_error(__FILE__, __LINE__, "health word allocation failed"); return NK_ERR_LAUNCH; }
  hipLaunchKernelGGL(nk_health_export_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, d, dst);
  return nk_check_launch("nk_health_export_kernel");
}
extern "C" int nk_health_import(const int* src, void* stream) {
  unsigned* d = nk_health_word();
  NK_CHECK_ARG(src != nullptr);
  if (!d) { nk_set_error(__FILE__, __LINE__, "health word allocation failed"); return NK_ERR_LAUNCH; }
  hipLaunchKernelGGL(nk_health_import_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, d, src);
  return nk_check_launch("nk_health_import_kernel");
}

// ---- dynamic-LDS opt-in, once per (kernel, device) ----------------------------------------------------------------------------------------
#include <mutex>
#include <set>
#include <utility>
void nk_optin_lds(const void* kernel, int bytes) {
  static std::mutex mu;
  static std::set<std::pair<const void*, int>> done;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(mu);
  if (done.insert({kernel, dev}).second) (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}
