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
__host__ __device__ inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// XCD-aware workgroup order: blocks are dealt round-robin over the 8 XCDs (each with its own 4 MB L2), so tiles
// that share an operand panel should have equal `block id % 8`.  1-D grids are padded to a multiple of 8 and the
// linear block id is mapped to a logical index that is contiguous per XCD; indices >= the real count exit.
constexpr int kXcds = 8;
inline unsigned xcd_grid(int64_t total) { return (unsigned)(kXcds * cdiv(total, kXcds)); }
__device__ __forceinline__ int64_t xcd_logical(int64_t bid, int64_t grid) { return (bid % kXcds) * (grid / kXcds) + bid / kXcds; }
// sort.hip: stable 32-bit key / value radix sort (rocPRIM)
size_t sort_pairs_u32_temp_bytes(int64_t n, int bits);
hipError_t sort_pairs_u32(void *tmp, size_t tmp_bytes, const uint32_t *kin, uint32_t *kout, const int32_t *vin, int32_t *vout,
                          int64_t n, int bits, hipStream_t s);
size_t scan_max_i32_temp_bytes(int64_t n);
hipError_t scan_max_i32(void *tmp, size_t tmp_bytes, const int32_t *in, int32_t *out, int64_t n, hipStream_t s);
// gemm_bf16x3.hip: dX = dY W on the scaled f16 split through the k-step-32 kernel (planes [K][N]); see svr_linear_bwd_data_f16x3
int linear_bwd_data_f16_nn(const float *dY, int64_t lddy, const float *W, int64_t ldw, float *dX, int64_t lddx, int64_t M, int64_t N,
                           int64_t K, int epilogue, const float *mask, int64_t ldmask, const uint32_t *amax_dy, uint32_t *amax_dx,
                           void *workspace, hipStream_t s);
size_t scan_sum_excl_i32_temp_bytes(int64_t n);
hipError_t scan_sum_excl_i32(void *tmp, size_t tmp_bytes, const int32_t *in, int32_t *out, int64_t n, hipStream_t s);
}  // namespace svr

#ifdef __HIPCC__
// Publish a wave's |max| into an amax word (bit pattern of a non-negative float; unsigned order = float order).  Tens of
// thousands of waves hit ONE address: an atomic per wave cost 0.25 ms on a 0.26 ms BatchNorm pass -- so the wave first reads the
// word (a stale, smaller value only costs an atomic that changes nothing) and issues the atomic only if it would raise it.
__device__ __forceinline__ void svr_amax_publish(uint32_t *amax, float vmax) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
  if ((threadIdx.x & 63) == 0) {
    const uint32_t bits = __float_as_uint(vmax);
    if (bits > *reinterpret_cast<volatile uint32_t *>(amax)) atomicMax(amax, bits);
  }
}
#endif

#define SVR_CHECK(cond, code, ...)      \
  do {                                  \
    if (!(cond)) {                      \
      svr::set_error(__VA_ARGS__);      \
      return (code);                    \
    }                                   \
  } while (0)
