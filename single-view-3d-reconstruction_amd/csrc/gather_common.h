// Sample geometry shared by the f32 and the bf16-storage gather kernels (gather.hip, bf16_path.hip): coordinate prep
// of model/ifnet.py:156-161 and ATen's grid_sampler_unnormalize / floor / corner weights
// (torch/include/ATen/native/GridSampler.h:27-36).  Every file that includes this must be compiled with
// -ffp-contract=off: the index arithmetic has to round after every operation like ATen's, or the bit-exact
// corner-index gate fails.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

struct Corner {
  float ix, iy, iz;     // f32 source index
  float x0f, y0f, z0f;  // floor()
};

__device__ __forceinline__ float unnormalize(float g, int S, int align_corners) {
  if (align_corners) return ((g + 1.0f) / 2.0f) * (float)(S - 1);
  return ((g + 1.0f) * (float)S - 1.0f) / 2.0f;
}

// j: 0 centre, 1/2 -+d on grid-x, 3/4 on grid-y, 5/6 on grid-z (model/ifnet.py:144-153).
__device__ __forceinline__ Corner sample_corner(const float *pt, int j, float disp,
                                                int D, int H, int W, int ac) {
  float gx = 2.0f * pt[2], gy = 2.0f * pt[1], gz = 2.0f * pt[0];
  float dj = (j & 1) ? -disp : disp;
  if (j == 1 || j == 2) gx = gx + dj;
  if (j == 3 || j == 4) gy = gy + dj;
  if (j == 5 || j == 6) gz = gz + dj;
  Corner c;
  c.ix = unnormalize(gx, W, ac);
  c.iy = unnormalize(gy, H, ac);
  c.iz = unnormalize(gz, D, ac);
  c.x0f = floorf(c.ix);
  c.y0f = floorf(c.iy);
  c.z0f = floorf(c.iz);
  return c;
}

__device__ __forceinline__ int clamp_int(float f) {
  f = fminf(fmaxf(f, -1.0e9f), 1.0e9f);
  return (f != f) ? -1000000000 : (int)f;
}

// weights of corner (a,b,c): wx[a]*wy[b]*wz[c] with w[0] = (i0+1) - i, w[1] = i - i0.
struct Weights {
  float wx[2], wy[2], wz[2];
  int x0, y0, z0;
};

__device__ __forceinline__ Weights corner_weights(const Corner &c) {
  Weights w;
  w.wx[0] = (c.x0f + 1.0f) - c.ix;
  w.wx[1] = c.ix - c.x0f;
  w.wy[0] = (c.y0f + 1.0f) - c.iy;
  w.wy[1] = c.iy - c.y0f;
  w.wz[0] = (c.z0f + 1.0f) - c.iz;
  w.wz[1] = c.iz - c.z0f;
  w.x0 = clamp_int(c.x0f);
  w.y0 = clamp_int(c.y0f);
  w.z0 = clamp_int(c.z0f);
  return w;
}

}  // namespace
