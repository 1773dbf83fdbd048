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
extern "C" const char *gn_version(void) { return "graspnet_hip 0.1 gfx950"; }
