// Error plumbing + version for the C ABI (include/svr_hip.h).
#include "common.h"
#include <string.h>

namespace svr {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace svr

extern "C" int svr_version(void) { return 200; }  // 2xx: round 2 (pull plans, item orders, projection, fused gather -> fc_0, bf16 mode, mesh, sample I/O)
extern "C" const char *svr_last_error(void) { return svr::g_err; }
extern "C" int64_t svr_sizeof_level(void) { return (int64_t)sizeof(svr_level); }
extern "C" int64_t svr_sizeof_gather_desc(void) { return (int64_t)sizeof(svr_gather_desc); }
