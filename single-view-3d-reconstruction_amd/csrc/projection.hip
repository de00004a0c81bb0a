// Depth map -> camera points -> grid space -> trilinear splat -> separable Gaussian blur, gfx950.
//
// Replaces the reference's `project` module (model/projection.py):
//   unproject  : depth_to_camera :199-206 + the 4x4 camera->frustum affine :160-161 (+ the
//                grid-space normalisation :124-132), one elementwise pass;
//   splat      : pc_voxels :39-80 (validity test, integer base voxel, 8 weighted index_put_
//                accumulations) with f32 atomics;  the x8 aliasing quirk and the clamp are
//                svr_scale_clamp01 (scale 8);
//   blur       : voxels_smooth :100-117, one pass per axis, taps from the learnable sigma.
// All are HBM / atomic bound elementwise kernels (no MFMA).  Compiled with -ffp-contract=off
// so the base-voxel arithmetic ((p+0.5)*(dims-1), floor) rounds exactly like torch's separate ops
// and the integer voxel indices are bit-exact.
#include "common.h"

using namespace svr;

namespace {

struct UnprojConsts {
  float f, cx, cy, s00, t0, s11, t1, s22, t2, d0, d1, d2;
};

__global__ void unproject_fwd_kernel(const float *__restrict__ depth, float *__restrict__ pc, int64_t total, int Hi,
                                     int Wi, UnprojConsts c, int normalize) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  int u = (int)(i % Wi), v = (int)((i / Wi) % Hi);
  float z = depth[i];
  float X = ((float)u * z - c.cx * z) / c.f;
  float Y = -(((float)v * z - c.cy * z) / c.f);
  float gx = c.s00 * X + c.t0, gy = c.s11 * Y + c.t1, gz = c.s22 * z + c.t2;
  if (normalize) {
    gx = (gx - c.d0 / 2.f) / c.d0;
    gy = (gy - c.d1 / 2.f) / c.d1;
    gz = (gz - c.d2 / 2.f) / c.d2;
  }
  pc[i * 3 + 0] = gx;
  pc[i * 3 + 1] = gy;
  pc[i * 3 + 2] = gz;
}

__global__ void unproject_bwd_kernel(const float *__restrict__ gpc, float *__restrict__ gdepth, int64_t total, int Hi,
                                     int Wi, UnprojConsts c, int normalize) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  int u = (int)(i % Wi), v = (int)((i / Wi) % Hi);
  float a0 = c.s00 * (((float)u - c.cx) / c.f);
  float a1 = -c.s11 * (((float)v - c.cy) / c.f);
  float a2 = c.s22;
  if (normalize) { a0 /= c.d0; a1 /= c.d1; a2 /= c.d2; }
  gdepth[i] = gpc[i * 3] * a0 + gpc[i * 3 + 1] * a1 + gpc[i * 3 + 2] * a2;
}

struct Splat {
  bool valid;
  int i0, i1, i2;
  float r0, r1, r2;
};

__device__ __forceinline__ Splat splat_of(const float *__restrict__ p, int D0, int D1, int D2) {
  const float hi = (float)(0.5 - 1e-6), lo = (float)(-0.5 + 1e-6);  // torch compares in float32
  Splat s;
  float a = p[0], b = p[1], c = p[2];
  s.valid = (a < hi && a > lo) && (b < hi && b > lo) && (c < hi && c > lo);
  float g0 = (a + 0.5f) * (float)(D0 - 1), g1 = (b + 0.5f) * (float)(D1 - 1), g2 = (c + 0.5f) * (float)(D2 - 1);
  float f0 = floorf(g0), f1 = floorf(g1), f2 = floorf(g2);
  s.r0 = g0 - f0; s.r1 = g1 - f1; s.r2 = g2 - f2;
  float cl = 2.0e9f;
  s.i0 = (int)fminf(fmaxf(f0, -cl), cl);
  s.i1 = (int)fminf(fmaxf(f1, -cl), cl);
  s.i2 = (int)fminf(fmaxf(f2, -cl), cl);
  if (a != a || b != b || c != c) { s.valid = false; s.i0 = s.i1 = s.i2 = 0; }
  return s;
}

__global__ void splat_fwd_kernel(const float *__restrict__ pts, float *__restrict__ acc, int32_t *__restrict__ base,
                                 uint8_t *__restrict__ valid, int64_t total, int N, int D0, int D1, int D2) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  Splat s = splat_of(pts + i * 3, D0, D1, D2);
  if (base) { base[i * 3] = s.i0; base[i * 3 + 1] = s.i1; base[i * 3 + 2] = s.i2; }
  if (valid) valid[i] = s.valid ? 1 : 0;
  if (!s.valid) return;
  int64_t b = i / N;
  float w0[2] = {1.f - s.r0, s.r0}, w1[2] = {1.f - s.r1, s.r1}, w2[2] = {1.f - s.r2, s.r2};
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int l = 0; l < 2; ++l) {
        int z = s.i0 + k, y = s.i1 + j, x = s.i2 + l;
        if (z >= 0 && z < D0 && y >= 0 && y < D1 && x >= 0 && x < D2)
          atomicAdd(acc + ((b * D0 + z) * D1 + y) * (int64_t)D2 + x, (w0[k] * w1[j]) * w2[l]);
      }
}

__global__ void splat_bwd_kernel(const float *__restrict__ pts, const float *__restrict__ gacc,
                                 float *__restrict__ gpts, int64_t total, int N, int D0, int D1, int D2) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  Splat s = splat_of(pts + i * 3, D0, D1, D2);
  float g0 = 0.f, g1 = 0.f, g2 = 0.f;
  if (s.valid) {
    int64_t b = i / N;
    float w0[2] = {1.f - s.r0, s.r0}, w1[2] = {1.f - s.r1, s.r1}, w2[2] = {1.f - s.r2, s.r2};
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int l = 0; l < 2; ++l) {
          int z = s.i0 + k, y = s.i1 + j, x = s.i2 + l;
          if (z >= 0 && z < D0 && y >= 0 && y < D1 && x >= 0 && x < D2) {
            float g = gacc[((b * D0 + z) * D1 + y) * (int64_t)D2 + x];
            g0 += (k ? g : -g) * w1[j] * w2[l];
            g1 += (j ? g : -g) * w0[k] * w2[l];
            g2 += (l ? g : -g) * w0[k] * w1[j];
          }
        }
    g0 *= (float)(D0 - 1); g1 *= (float)(D1 - 1); g2 *= (float)(D2 - 1);
  }
  gpts[i * 3] = g0; gpts[i * 3 + 1] = g1; gpts[i * 3 + 2] = g2;
}

__global__ void scale_clamp_fwd_kernel(const float *__restrict__ in, float *__restrict__ out, int64_t n, float scale) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = fminf(fmaxf(in[i] * scale, 0.f), 1.f);
}

__global__ void scale_clamp_bwd_kernel(const float *__restrict__ in, const float *__restrict__ gout,
                                       float *__restrict__ gin, int64_t n, float scale) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v = in[i] * scale;
  gin[i] = (v >= 0.f && v <= 1.f) ? gout[i] * scale : 0.f;
}

constexpr int KMAX = 15;

// out[i] = sum_t in[i + (t - K/2)*stride_axis] * taps[t]   (ADJ: gin = sum_t gout[i - (t-K/2)] taps[t])
template <bool ADJ>
__global__ void blur_axis_kernel(const float *__restrict__ in, const float *__restrict__ taps, float *__restrict__ out,
                                 int64_t total, int len, int64_t stride, int K) {
  __shared__ float tp[KMAX];
  if ((int)threadIdx.x < K) tp[threadIdx.x] = taps[threadIdx.x];
  __syncthreads();
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  int pos = (int)((i / stride) % len);
  float s = 0.f;
  for (int t = 0; t < K; ++t) {
    int o = ADJ ? (K / 2 - t) : (t - K / 2);
    int p = pos + o;
    if (p >= 0 && p < len) s += in[i + (int64_t)o * stride] * tp[t];
  }
  out[i] = s;
}

// gtaps[t] += sum_i gout[i] * in[i + (t-K/2)]
__global__ __launch_bounds__(256) void blur_taps_grad_kernel(const float *__restrict__ in,
                                                             const float *__restrict__ gout, double *__restrict__ gtaps,
                                                             int64_t total, int len, int64_t stride, int K) {
  float s[KMAX];
#pragma unroll
  for (int t = 0; t < KMAX; ++t) s[t] = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int pos = (int)((i / stride) % len);
    float g = gout[i];
#pragma unroll
    for (int t = 0; t < KMAX; ++t) {
      if (t < K) {
        int p = pos + t - K / 2;
        if (p >= 0 && p < len) s[t] += g * in[i + (int64_t)(t - K / 2) * stride];
      }
    }
  }
  __shared__ double red[256];
  for (int t = 0; t < K; ++t) {
    red[threadIdx.x] = (double)s[t];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) atomicAdd(gtaps + t, red[0]);
    __syncthreads();
  }
}

UnprojConsts load_consts(const float *c) { return UnprojConsts{c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7], c[8], c[9], c[10], c[11]}; }

int axis_geometry(int D0, int D1, int D2, int axis, int *len, int64_t *stride) {
  SVR_CHECK(axis >= 0 && axis <= 2, SVR_E_BADARG, "blur: axis %d", axis);
  *len = axis == 0 ? D0 : (axis == 1 ? D1 : D2);
  *stride = axis == 0 ? (int64_t)D1 * D2 : (axis == 1 ? D2 : 1);
  return SVR_OK;
}

}  // namespace

extern "C" int svr_unproject_fwd(const float *depth, float *pc, int32_t B, int32_t Hi, int32_t Wi, const float *consts,
                                 int normalize, void *stream) {
  SVR_CHECK(depth && pc && consts, SVR_E_BADARG, "unproject_fwd: null pointer");
  int64_t total = (int64_t)B * Hi * Wi;
  if (total <= 0) return SVR_OK;
  hipLaunchKernelGGL(unproject_fwd_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, depth, pc,
                     total, Hi, Wi, load_consts(consts), normalize);
  return launch_status("unproject_fwd");
}

extern "C" int svr_unproject_bwd(const float *depth, const float *gpc, float *gdepth, int32_t B, int32_t Hi, int32_t Wi,
                                 const float *consts, int normalize, void *stream) {
  (void)depth;
  SVR_CHECK(gpc && gdepth && consts, SVR_E_BADARG, "unproject_bwd: null pointer");
  int64_t total = (int64_t)B * Hi * Wi;
  if (total <= 0) return SVR_OK;
  hipLaunchKernelGGL(unproject_bwd_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, gpc,
                     gdepth, total, Hi, Wi, load_consts(consts), normalize);
  return launch_status("unproject_bwd");
}

extern "C" int svr_voxelize_splat_fwd(const float *pts, float *acc, int32_t *base, uint8_t *valid, int32_t B, int32_t N,
                                      int32_t D0, int32_t D1, int32_t D2, void *stream) {
  int64_t total = (int64_t)B * N;
  if (total <= 0) return SVR_OK;  // empty point cloud: nothing to add
  SVR_CHECK(pts && acc, SVR_E_BADARG, "splat_fwd: null pointer");
  SVR_CHECK(D0 > 0 && D1 > 0 && D2 > 0, SVR_E_BADSHAPE, "splat_fwd: dims %dx%dx%d", D0, D1, D2);
  hipLaunchKernelGGL(splat_fwd_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, pts, acc, base,
                     valid, total, N, D0, D1, D2);
  return launch_status("splat_fwd");
}

extern "C" int svr_voxelize_splat_bwd(const float *pts, const float *gacc, float *gpts, int32_t B, int32_t N, int32_t D0,
                                      int32_t D1, int32_t D2, void *stream) {
  int64_t total = (int64_t)B * N;
  if (total <= 0) return SVR_OK;
  SVR_CHECK(pts && gacc && gpts, SVR_E_BADARG, "splat_bwd: null pointer");
  hipLaunchKernelGGL(splat_bwd_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, pts, gacc, gpts,
                     total, N, D0, D1, D2);
  return launch_status("splat_bwd");
}

extern "C" int svr_scale_clamp01_fwd(const float *in, float *out, int64_t n, float scale, void *stream) {
  SVR_CHECK(in && out, SVR_E_BADARG, "scale_clamp01_fwd: null pointer");
  if (n <= 0) return SVR_OK;
  hipLaunchKernelGGL(scale_clamp_fwd_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, in, out, n, scale);
  return launch_status("scale_clamp01_fwd");
}

extern "C" int svr_scale_clamp01_bwd(const float *in, const float *gout, float *gin, int64_t n, float scale, void *stream) {
  SVR_CHECK(in && gout && gin, SVR_E_BADARG, "scale_clamp01_bwd: null pointer");
  if (n <= 0) return SVR_OK;
  hipLaunchKernelGGL(scale_clamp_bwd_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, in, gout, gin, n, scale);
  return launch_status("scale_clamp01_bwd");
}

extern "C" int svr_blur_axis_fwd(const float *in, const float *taps, float *out, int32_t B, int32_t D0, int32_t D1,
                                 int32_t D2, int32_t axis, int32_t K, void *stream) {
  SVR_CHECK(in && taps && out, SVR_E_BADARG, "blur_fwd: null pointer");
  SVR_CHECK(K >= 1 && K <= KMAX && (K & 1), SVR_E_UNSUPPORTED, "blur_fwd: K=%d (odd, <= %d)", K, KMAX);
  int len;
  int64_t stride;
  if (int rc = axis_geometry(D0, D1, D2, axis, &len, &stride)) return rc;
  int64_t total = (int64_t)B * D0 * D1 * D2;
  if (total <= 0) return SVR_OK;
  hipLaunchKernelGGL(blur_axis_kernel<false>, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, in, taps,
                     out, total, len, stride, K);
  return launch_status("blur_fwd");
}

extern "C" int svr_blur_axis_bwd(const float *in, const float *taps, const float *gout, float *gin, double *gtaps,
                                 int32_t B, int32_t D0, int32_t D1, int32_t D2, int32_t axis, int32_t K, void *stream) {
  SVR_CHECK(in && taps && gout, SVR_E_BADARG, "blur_bwd: null pointer");
  SVR_CHECK(K >= 1 && K <= KMAX && (K & 1), SVR_E_UNSUPPORTED, "blur_bwd: K=%d (odd, <= %d)", K, KMAX);
  int len;
  int64_t stride;
  if (int rc = axis_geometry(D0, D1, D2, axis, &len, &stride)) return rc;
  int64_t total = (int64_t)B * D0 * D1 * D2;
  if (total <= 0) return SVR_OK;
  hipStream_t s = (hipStream_t)stream;
  if (gin)
    hipLaunchKernelGGL(blur_axis_kernel<true>, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, s, gout, taps, gin, total,
                       len, stride, K);
  if (gtaps) {
    int64_t blocks = cdiv(total, 256 * 8);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(blur_taps_grad_kernel, dim3((unsigned)blocks), dim3(256), 0, s, in, gout, gtaps, total, len, stride, K);
  }
  return launch_status("blur_bwd");
}
