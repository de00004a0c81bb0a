// Device helpers of the 3-product f16 split ("f16x3", see gemm_f16x3.hip for the arithmetic), shared by the forward
// GEMM and the forward convolution.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pack_f16(float a, float b) {
  f32x2 v = {a, b};
  f16x2 h = __builtin_convertvector(v, f16x2);  // round to nearest even
  return __builtin_bit_cast(uint32_t, h);
}
__device__ __forceinline__ f32x2 unpack_f16(uint32_t p) {
  return __builtin_convertvector(__builtin_bit_cast(f16x2, p), f32x2);
}

// (x0, x1) -> packed hi pair, packed (lo * 2^11) pair
__device__ __forceinline__ void split_x(float x0, float x1, uint32_t &hi, uint32_t &lo) {
  hi = pack_f16(x0, x1);
  const f32x2 h = unpack_f16(hi);
  lo = pack_f16((x0 - h.x) * 2048.f, (x1 - h.y) * 2048.f);
}

__device__ __forceinline__ f16x8 scale_2m11(f16x8 v) {  // exact: every normal hi(Ws) stays normal (see header)
  return v * (_Float16)(1.f / 2048.f);
}

// amax[0] = bit pattern of max |W| (non-negative floats order like unsigned integers); zeroed by the host first
__global__ __launch_bounds__(256) void w_amax_kernel(const float *__restrict__ W, int64_t ldw, int64_t N, int64_t K,
                                                     uint32_t *__restrict__ amax) {
  __shared__ float red[256];
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N * K; i += (int64_t)gridDim.x * 256)
    m = fmaxf(m, fabsf(W[(i / K) * ldw + i % K]));
  red[threadIdx.x] = m;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x == 0) atomicMax(amax, __float_as_uint(red[0]));
}

// 2^s (or 2^-s) with amax * 2^s in [2^13, 2^14); s = 0 for an all-zero or non-finite W
__device__ __forceinline__ float w_scale(uint32_t amax_bits, bool inverse) {
  const float amax = __uint_as_float(amax_bits);
  int e = 0;
  if (amax > 0.f && amax < 3.0e38f) {
    frexpf(amax, &e);  // amax = f * 2^e, f in [0.5, 1)
    e = 14 - e;
    e = e > 100 ? 100 : (e < -100 ? -100 : e);
  }
  return ldexpf(1.f, inverse ? -e : e);
}

}  // namespace
