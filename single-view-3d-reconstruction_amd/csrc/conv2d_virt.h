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

// channels c .. c+3 of pixel (iy, ix) of the sample whose first pixel is `pix0` (= b H W; every caller checks B H W < 2^24, so
// pixel indices are 32-bit and an element offset is ONE v_mad_u64_u32), RAW (no activation); zero outside the image or past the
// last channel.  VEC4: C0, C1 multiples of 4 and 16-byte aligned sources (one 16-byte load); otherwise four predicated loads.
template <bool VEC4>
__device__ __forceinline__ float4 cv_load4(const CvSrc &S, int pix0, int iy, int ix, int c) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if constexpr (VEC4) {
    // branch free: ONE unconditional load from clamped coordinates, zeroed by a select (a load inside a bounds branch is waited
    // for on its own; the GEMM loaders issue 2-4 of these per k-step and want them in flight together)
    const bool ok = (unsigned)iy < (unsigned)S.H && (unsigned)ix < (unsigned)S.W && c < S.C0 + S.C1;
    const int cy = min(max(iy, 0), S.H - 1), cx = min(max(ix, 0), S.W - 1);
    const uint32_t pix = (uint32_t)(pix0 + cy * S.W + cx);
    const bool first = c < S.C0 || S.C1 == 0;
    const float *base = first ? S.p0 : S.p1;
    const uint32_t Cs = (uint32_t)(first ? S.C0 : S.C1), cc = (uint32_t)(ok ? (c < S.C0 ? c : c - S.C0) : 0);
    const float4 t = *reinterpret_cast<const float4 *>(base + ((uint64_t)pix * Cs + cc));
    v = make_float4(ok ? t.x : 0.f, ok ? t.y : 0.f, ok ? t.z : 0.f, ok ? t.w : 0.f);
  } else {
    if ((unsigned)iy >= (unsigned)S.H || (unsigned)ix >= (unsigned)S.W) return v;
    const uint32_t pix = (uint32_t)(pix0 + iy * S.W + ix);
    float e[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ci = c + i;
      e[i] = ci < S.C0 ? S.p0[(uint64_t)pix * (uint32_t)S.C0 + (uint32_t)ci]
                       : (ci < S.C0 + S.C1 ? S.p1[(uint64_t)pix * (uint32_t)S.C1 + (uint32_t)(ci - S.C0)] : 0.f);
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
