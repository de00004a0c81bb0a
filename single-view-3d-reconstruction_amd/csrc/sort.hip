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

constexpr int MORTON_BITS = 6;  // 64^3 cells for the global order

__device__ __forceinline__ uint64_t spread3_10(uint32_t v) {  // 10 bits -> every third bit
  uint64_t x = v & 0x3ff;
  x = (x | (x << 16)) & 0x30000ffULL;
  x = (x | (x << 8)) & 0x300f00fULL;
  x = (x | (x << 4)) & 0x30c30c3ULL;
  x = (x | (x << 2)) & 0x9249249ULL;
  return x;
}

// mode 0: Morton code of the position on a 64^3 lattice.  mode 1: row-major index (x fastest) of the base voxel
// floor(source index)+1 of the undisplaced sample in a D x H x W volume (same arithmetic as gather.hip;
// a last-bit difference would only cost a run split, never correctness).
__global__ void sort_key_kernel(const float *__restrict__ points, uint64_t *__restrict__ keys, int32_t *__restrict__ vals,
                                int64_t total, int N, int mode, int D, int H, int W, int ac, int code_bits) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  uint32_t q[3];  // z, y, x
  const int S[3] = {D, H, W};
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float p = points[i * 3 + a];
    float f;
    int hi;
    if (mode == 0) {
      f = (p + 0.5f) * (float)(1 << MORTON_BITS);
      hi = (1 << MORTON_BITS) - 1;
    } else {
      float g = 2.0f * p;
      float src = ac ? ((g + 1.0f) / 2.0f) * (float)(S[a] - 1) : ((g + 1.0f) * (float)S[a] - 1.0f) / 2.0f;
      f = floorf(src) + 1.0f;
      hi = S[a];
    }
    f = fminf(fmaxf(f, 0.f), (float)hi);
    q[a] = (p == p) ? (uint32_t)f : 0u;
  }
  // points[...,0] walks the slowest volume axis (z), [...,2] the fastest (x): x in the low bit.
  // mode 1 orders the base voxels ROW-MAJOR (x fastest): consecutive occupied cells are then x neighbours most of
  // the time, which is what the scatter's face hand-over between neighbouring runs needs (Morton order: about half).
  uint64_t m = mode == 1 ? ((uint64_t)q[0] * (uint64_t)(H + 1) + q[1]) * (uint64_t)(W + 1) + q[2]
                         : (spread3_10(q[2]) | (spread3_10(q[1]) << 1) | (spread3_10(q[0]) << 2));
  keys[i] = ((uint64_t)(i / N) << code_bits) | m;  // only the occupied bits are sorted: fewer radix passes
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

// bits of the interleaved cell code: 3 x (bits of the largest per-axis cell index)
int code_bits(int mode, int D, int H, int W) {
  if (mode == 1) {  // row-major cell index < (D+1)(H+1)(W+1)
    const int64_t cells = (int64_t)(D + 1) * (H + 1) * (W + 1);
    int nb = 1;
    while ((1LL << nb) < cells) ++nb;
    return nb;
  }
  int nb = 1;
  while ((1 << nb) <= (1 << MORTON_BITS) - 1) ++nb;
  return 3 * nb;
}

int key_bits(int B, int cbits) {
  int bb = 0;
  while ((1LL << bb) < B) ++bb;
  return cbits + bb;
}

size_t rocprim_temp_bytes(int64_t total, int bits) {
  size_t tmp = 0;
  rocprim::radix_sort_pairs(nullptr, tmp, (uint64_t *)nullptr, (uint64_t *)nullptr, (int32_t *)nullptr,
                            (int32_t *)nullptr, (size_t)total, 0, bits, (hipStream_t)0);
  return tmp;
}

int64_t align256(int64_t x) { return (x + 255) / 256 * 256; }

int sort_points(const float *points, int32_t *order, float *sorted_points, int B, int N, int mode, int D, int H, int W,
                int ac, void *workspace, hipStream_t s, const char *what) {
  int64_t total = (int64_t)B * N;
  if (total <= 0) return SVR_OK;
  SVR_CHECK(points && order && workspace, SVR_E_BADARG, "%s: null pointer", what);
  SVR_CHECK(mode == 0 || (D > 0 && H > 0 && W > 0 && D < 1023 && H < 1023 && W < 1023), SVR_E_UNSUPPORTED,
            "%s: volume dims must be in [1, 1022]", what);
  char *w = (char *)workspace;
  uint64_t *keys_in = (uint64_t *)w;
  w += align256(total * 8);
  uint64_t *keys_out = (uint64_t *)w;
  w += align256(total * 8);
  int32_t *vals_in = (int32_t *)w;
  w += align256(total * 4);
  const int cbits = code_bits(mode, D, H, W);
  int bits = key_bits(B, cbits);
  size_t tmp = rocprim_temp_bytes(total, bits);
  hipLaunchKernelGGL(sort_key_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, s, points, keys_in, vals_in, total, N,
                     mode, D, H, W, ac, cbits);
  hipError_t e = rocprim::radix_sort_pairs((void *)w, tmp, keys_in, keys_out, vals_in, order, (size_t)total, 0, bits, s);
  SVR_CHECK(e == hipSuccess, (int)e, "%s: radix sort failed: %s", what, hipGetErrorString(e));
  if (sorted_points)
    hipLaunchKernelGGL(permute_points_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, s, points, order,
                       sorted_points, total);
  return launch_status(what);
}

}  // namespace

namespace svr {
// 32-bit key / 32-bit value radix sort (stable) for the pull-form scatter plan (gather.hip)
size_t sort_pairs_u32_temp_bytes(int64_t n, int bits) {
  size_t tmp = 0;
  rocprim::radix_sort_pairs(nullptr, tmp, (uint32_t *)nullptr, (uint32_t *)nullptr, (int32_t *)nullptr, (int32_t *)nullptr,
                            (size_t)n, 0, bits, (hipStream_t)0);
  return tmp;
}
hipError_t sort_pairs_u32(void *tmp, size_t tmp_bytes, const uint32_t *kin, uint32_t *kout, const int32_t *vin, int32_t *vout,
                          int64_t n, int bits, hipStream_t s) {
  return rocprim::radix_sort_pairs(tmp, tmp_bytes, kin, kout, vin, vout, (size_t)n, 0, bits, s);
}
// inclusive running maximum (cell -> number of items with a smaller-or-equal key) for the pull plan's CSR offsets
size_t scan_max_i32_temp_bytes(int64_t n) {
  size_t tmp = 0;
  rocprim::inclusive_scan(nullptr, tmp, (const int32_t *)nullptr, (int32_t *)nullptr, (size_t)n, rocprim::maximum<int32_t>(),
                          (hipStream_t)0);
  return tmp;
}
hipError_t scan_max_i32(void *tmp, size_t tmp_bytes, const int32_t *in, int32_t *out, int64_t n, hipStream_t s) {
  return rocprim::inclusive_scan(tmp, tmp_bytes, in, out, (size_t)n, rocprim::maximum<int32_t>(), s);
}
// exclusive prefix sum (run-start flags -> slot numbers of the two-pass projected scatter)
size_t scan_sum_excl_i32_temp_bytes(int64_t n) {
  size_t tmp = 0;
  rocprim::exclusive_scan(nullptr, tmp, (const int32_t *)nullptr, (int32_t *)nullptr, (int32_t)0, (size_t)n,
                          rocprim::plus<int32_t>(), (hipStream_t)0);
  return tmp;
}
hipError_t scan_sum_excl_i32(void *tmp, size_t tmp_bytes, const int32_t *in, int32_t *out, int64_t n, hipStream_t s) {
  return rocprim::exclusive_scan(tmp, tmp_bytes, in, out, (int32_t)0, (size_t)n, rocprim::plus<int32_t>(), s);
}
}  // namespace svr

extern "C" int64_t svr_points_morton_order_workspace(int32_t B, int32_t N) {
  int64_t total = (int64_t)B * N;
  if (total <= 0) return 256;
  return 2 * align256(total * 8) + align256(total * 4) + align256((int64_t)rocprim_temp_bytes(total, key_bits(B, 30))) + 256;
}

extern "C" int svr_points_morton_order(const float *points, int32_t *order, float *sorted_points, int32_t B, int32_t N,
                                       void *workspace, void *stream) {
  return sort_points(points, order, sorted_points, B, N, 0, 0, 0, 0, 0, workspace, (hipStream_t)stream, "morton_order");
}

extern "C" int svr_points_voxel_order(const float *points, int32_t *order, int32_t B, int32_t N, int32_t D, int32_t H,
                                      int32_t W, int32_t align_corners, void *workspace, void *stream) {
  return sort_points(points, order, nullptr, B, N, 1, D, H, W, align_corners, workspace, (hipStream_t)stream, "voxel_order");
}
