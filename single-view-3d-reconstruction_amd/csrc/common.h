// Shared host-side helpers for libsvr_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "svr_hip.h"

namespace svr {
void set_error(const char *fmt, ...);
inline int launch_status(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return (int)e;
  }
  return SVR_OK;
}
inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
}  // namespace svr

#define SVR_CHECK(cond, code, ...)      \
  do {                                  \
    if (!(cond)) {                      \
      svr::set_error(__VA_ARGS__);      \
      return (code);                    \
    }                                   \
  } while (0)
