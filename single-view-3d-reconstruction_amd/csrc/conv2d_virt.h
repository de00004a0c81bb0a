// Operand loaders of the UNet's implicit-GEMM convolutions (conv2d_igemm.hip: forward / backward-data; gemm_bf16x3.hip: weight
// gradient), gfx950.  The "input" of a convolution block is one channels-last tensor or the channel concatenation of two
// (reference model/unet.py:91-110: torch.cat((bn_out, skip), 1) in front of every decoder convolution) with LeakyReLU(0.2) / ReLU
// applied on the way in -- neither the concatenation nor the activated tensor nor a patch matrix exists in memory.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

struct CvSrc {
  const float *p0, *p1;  // (B, H, W, C0) and (B, H, W, C1) or null
  int C0, C1, H, W;
  int act;               // 0 none, 1 LeakyReLU(0.2), 2 ReLU
};

__device__ __forceinline__ float cv_act(float v, int act) { return act == 1 ? (v > 0.f ? v : 0.2f * v) : (act == 2 ? fmaxf(v, 0.f) : v); }
__device__ __forceinline__ float4 cv_act4(float4 v, int act) {
  return make_float4(cv_act(v.x, act), cv_act(v.y, act), cv_act(v.z, act), cv_act(v.w, act));
}

// channels c .. c+3 of pixel (iy, ix) of sample `pix0 / (H W)`, RAW (no activation); zero outside the image or past the last
// channel.  VEC4: C0, C1 multiples of 4 and 16-byte aligned sources (one 16-byte load); otherwise four predicated loads.
template <bool VEC4>
__device__ __forceinline__ float4 cv_load4(const CvSrc &S, int64_t pix0, int iy, int ix, int c) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if ((unsigned)iy >= (unsigned)S.H || (unsigned)ix >= (unsigned)S.W) return v;
  const int64_t pix = pix0 + (int64_t)iy * S.W + ix;
  if constexpr (VEC4) {
    if (c < S.C0) v = *reinterpret_cast<const float4 *>(S.p0 + pix * S.C0 + c);
    else if (c < S.C0 + S.C1) v = *reinterpret_cast<const float4 *>(S.p1 + pix * S.C1 + (c - S.C0));
  } else {
    float e[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ci = c + i;
      e[i] = ci < S.C0 ? S.p0[pix * S.C0 + ci] : (ci < S.C0 + S.C1 ? S.p1[pix * S.C1 + (ci - S.C0)] : 0.f);
    }
    v = make_float4(e[0], e[1], e[2], e[3]);
  }
  return v;
}

// n / d and n % d for 0 <= n < 2^24, 0 < d < 2^24 through one float multiply (rcp = 1.0f / d) and one correction step
__device__ __forceinline__ void cv_divmod(int n, int d, float rcp, int &q, int &r) {
  q = (int)((float)n * rcp);
  r = n - q * d;
  if (r < 0) { --q; r += d; }
  else if (r >= d) { ++q; r -= d; }
}

}  // namespace
