// gn_common.cpp -- error state and version string of libgraspnet_hip.
#include <stdarg.h>
#include <string.h>

#include "gn_common.h"

namespace gn {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace gn

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
