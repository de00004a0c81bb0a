// Spatial processing order for the query points (gfx950).
//
// Not a reference op: the reference samples points in the order they arrive.  Sorting the
// (sample, point) pairs by a Morton code of their position makes consecutive work items touch
// neighbouring voxels, which (a) turns the multi-level gather's reads into L2 hits and (b) lets
// the backward scatter combine runs of samples that share the same 8 corners in registers
// before it issues global atomics (gather.hip).  Only the PROCESSING order changes: feature rows,
// logits and gradients keep the caller's point order, so results are order-independent up to
// float summation order in the scatter.
#include "common.h"
#include <cstring>
#include <cstdlib>
#include <rocprim/rocprim.hpp>

using namespace svr;

namespace {

constexpr int MORTON_BITS = 6;  // 64^3 cells

__device__ __forceinline__ uint32_t spread3(uint32_t v) {  // 6 bits -> every third bit
  v &= 0x3f;
  v = (v | (v << 8)) & 0x300f;
  v = (v | (v << 4)) & 0x30c3;
  v = (v | (v << 2)) & 0x9249;
  return v;
}

__global__ void morton_key_kernel(const float *__restrict__ points, uint32_t *__restrict__ keys,
                                  int32_t *__restrict__ vals, int64_t total, int N) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  uint32_t q[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float p = points[i * 3 + a];
    float f = (p + 0.5f) * (float)(1 << MORTON_BITS);
    f = fminf(fmaxf(f, 0.f), (float)((1 << MORTON_BITS) - 1));
    q[a] = (p == p) ? (uint32_t)f : 0u;
  }
  // points[...,0] walks the slowest volume axis (z), [...,2] the fastest (x): x in the low bit
  uint32_t m = spread3(q[2]) | (spread3(q[1]) << 1) | (spread3(q[0]) << 2);
  keys[i] = ((uint32_t)(i / N) << (3 * MORTON_BITS)) | m;
  vals[i] = (int32_t)i;
}

__global__ void permute_points_kernel(const float *__restrict__ points, const int32_t *__restrict__ order,
                                      float *__restrict__ out, int64_t total) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  int64_t src = order[i];
  out[i * 3] = points[src * 3];
  out[i * 3 + 1] = points[src * 3 + 1];
  out[i * 3 + 2] = points[src * 3 + 2];
}

int key_bits(int B) {
  int bb = 0;
  while ((1 << bb) < B) ++bb;
  return 3 * MORTON_BITS + bb;
}

size_t rocprim_temp_bytes(int64_t total, int bits) {
  size_t tmp = 0;
  rocprim::radix_sort_pairs(nullptr, tmp, (uint32_t *)nullptr, (uint32_t *)nullptr, (int32_t *)nullptr,
                            (int32_t *)nullptr, (size_t)total, 0, bits, (hipStream_t)0);
  return tmp;
}

int64_t align256(int64_t x) { return (x + 255) / 256 * 256; }

}  // namespace

extern "C" int64_t svr_points_morton_order_workspace(int32_t B, int32_t N) {
  int64_t total = (int64_t)B * N;
  if (total <= 0) return 256;
  return 3 * align256(total * 4) + align256((int64_t)rocprim_temp_bytes(total, key_bits(B))) + 256;
}

extern "C" int svr_points_morton_order(const float *points, int32_t *order, float *sorted_points, int32_t B, int32_t N,
                                       void *workspace, void *stream) {
  int64_t total = (int64_t)B * N;
  if (total <= 0) return SVR_OK;
  SVR_CHECK(points && order && workspace, SVR_E_BADARG, "morton_order: null pointer");
  SVR_CHECK(B < (1 << (32 - 3 * MORTON_BITS)), SVR_E_UNSUPPORTED, "morton_order: batch %d too large for 32-bit keys", B);
  hipStream_t s = (hipStream_t)stream;
  char *w = (char *)workspace;
  uint32_t *keys_in = (uint32_t *)w;
  w += align256(total * 4);
  uint32_t *keys_out = (uint32_t *)w;
  w += align256(total * 4);
  int32_t *vals_in = (int32_t *)w;
  w += align256(total * 4);
  int bits = key_bits(B);
  size_t tmp = rocprim_temp_bytes(total, bits);
  hipLaunchKernelGGL(morton_key_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, s, points, keys_in, vals_in, total, N);
  hipError_t e = rocprim::radix_sort_pairs((void *)w, tmp, keys_in, keys_out, vals_in, order, (size_t)total, 0, bits, s);
  SVR_CHECK(e == hipSuccess, (int)e, "morton_order: radix sort failed: %s", hipGetErrorString(e));
  if (sorted_points)
    hipLaunchKernelGGL(permute_points_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, s, points, order, sorted_points, total);
  return launch_status("morton_order");
}
