// gn_common.cpp -- error state and version string of libgraspnet_hip.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>

#include "gn_common.h"

namespace gn {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
// evaluation policy of the index-deciding float expressions (gn_common.h): -1 = not set yet -> GN_FP from the environment
static std::atomic<int> g_fp_policy{-1};
static int fp_policy_from_name(const char *e) {
  if (!e || !*e) return 0;
  if (!strcmp(e, "1") || !strcmp(e, "contracted") || !strcmp(e, "contracted_a") || !strcmp(e, "fma")) return 1;
  if (!strcmp(e, "2") || !strcmp(e, "contracted_b") || !strcmp(e, "fma_b")) return 2;
  return 0;       // "0", "as_written", anything else
}
int fp_policy() {
  int p = g_fp_policy.load(std::memory_order_relaxed);
  if (p < 0) {
    p = fp_policy_from_name(getenv("GN_FP"));
    g_fp_policy.store(p, std::memory_order_relaxed);
  }
  return p;
}
}  // namespace gn

extern "C" int gn_set_fp_policy(int policy) {
  if (policy < 0 || policy > 2) {
    gn::set_error("gn_set_fp_policy: policy must be 0 (as written), 1 (contracted) or 2 (contracted, other order), got %d", policy);
    return GN_ERR_INVALID_ARG;
  }
  gn::g_fp_policy.store(policy, std::memory_order_relaxed);
  return GN_OK;
}
extern "C" int gn_get_fp_policy(void) { return gn::fp_policy(); }

extern "C" const char *gn_last_error(void) { return gn::g_err; }
extern "C" const char *gn_version(void) { return "graspnet_hip 0.5 gfx950"; }
extern "C" int gn_abi_version(void) { return GN_ABI_VERSION; }

// 0 when `stream` is not being captured into a hipGraph, otherwise the capture sequence's unique id (> 0): what a host-side
// cache needs to tell "this entry was enqueued inside the capture that is running now" from everything else.
extern "C" int gn_stream_capture_id(gn_stream_t stream, unsigned long long *id) {
  GN_REQUIRE(id != nullptr, "gn_stream_capture_id: id is NULL");
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  unsigned long long cid = 0;
  const hipError_t e = hipStreamGetCaptureInfo(gn::as_stream(stream), &st, &cid);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    gn::set_error("gn_stream_capture_id: hipStreamGetCaptureInfo: %s", hipGetErrorString(e));
    return GN_ERR_LAUNCH;
  }
  *id = st == hipStreamCaptureStatusNone ? 0ull : (cid ? cid : ~0ull);
  return GN_OK;
}
