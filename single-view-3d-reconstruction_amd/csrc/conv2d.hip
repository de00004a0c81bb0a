// 2-D convolution blocks of the UNet depth regressor (reference model/unet.py:15-118,
// :121-186), gfx950.  Channels-last activations (B, H, W, C) float32.
//
// A UNet layer is   [LeakyReLU(0.2) | ReLU] -> [bilinear x2 upsample] -> Conv2d(k4 s2 p1 | k3 s1 p1) (+ bias)
// on an input that is either one tensor or the channel concatenation of two (decoder: torch.cat((bn_out, skip), 1),
// model/unet.py:91-110) -- the concatenation is never materialised.  The layers are small and many (16 convolutions,
// 21 GFLOP forward at batch 4): stock MIOpen runs them at 15-50 ms per step on gfx950 (naive_conv_* fall-backs).  Here
// a layer is an explicit im2col (activation, upsample and concat fused into the gather: one pass, channel-contiguous
// float4 traffic) + the split-precision MFMA GEMMs of the point MLP (gemm_f16x3.hip forward, gemm_bf16x3.hip backward),
// and the backward is gather-form throughout (col2im sums the <= k*k taps that touch a pixel; the upsample adjoint sums
// the <= 16 virtual pixels that read a source pixel): no atomics, deterministic.
//
// Bilinear upsample = at::upsample_bilinear2d, align_corners = False, scale 2: src = max(0, 0.5 * (dst + 0.5) - 0.5),
// i0 = (int)src, i1 = i0 + (i0 < n - 1), l1 = src - i0, l0 = 1 - l1; value = l0y*(l0x*v00 + l1x*v01) + l1y*(l0x*v10 + l1x*v11).
#include "common.h"

using namespace svr;

namespace {

struct Src2 {
  const float *p0, *p1;  // (B, H, W, C0) and (B, H, W, C1) or null
  int C0, C1;
};

__device__ __forceinline__ float act_f(float v, int act) { return act == 1 ? (v > 0.f ? v : 0.2f * v) : (act == 2 ? fmaxf(v, 0.f) : v); }
__device__ __forceinline__ float act_g(float v, int act) { return act == 1 ? (v > 0.f ? 1.f : 0.2f) : (act == 2 ? (v > 0.f ? 1.f : 0.f) : 1.f); }

__device__ __forceinline__ void up_axis(int dst, int n, int &i0, int &i1, float &l0, float &l1) {
  const float s = fmaxf(0.5f * ((float)dst + 0.5f) - 0.5f, 0.f);
  i0 = (int)s;
  i1 = i0 + (i0 < n - 1 ? 1 : 0);
  l1 = s - (float)i0;
  l0 = 1.f - l1;
}

// VEC channels (1 or 4) of the virtual input at (b, y, x): activation, optional x2 upsample, concat of two sources
template <int VEC>
__device__ __forceinline__ void load_virtual(const Src2 &S, int b, int y, int x, int c, int H, int W, int act, int up, float *out) {
  const float *p = c < S.C0 ? S.p0 : S.p1;
  const int C = c < S.C0 ? S.C0 : S.C1, cc = c < S.C0 ? c : c - S.C0;
  auto at = [&](int yy, int xx, float *v) {
    const float *q = p + (((int64_t)b * H + yy) * W + xx) * C + cc;
    if constexpr (VEC == 4) {
      const float4 t = *reinterpret_cast<const float4 *>(q);
      v[0] = act_f(t.x, act); v[1] = act_f(t.y, act); v[2] = act_f(t.z, act); v[3] = act_f(t.w, act);
    } else {
      v[0] = act_f(q[0], act);
    }
  };
  if (!up) {
    at(y, x, out);
    return;
  }
  int y0, y1, x0, x1;
  float ly0, ly1, lx0, lx1;
  up_axis(y, H, y0, y1, ly0, ly1);
  up_axis(x, W, x0, x1, lx0, lx1);
  float v00[VEC], v01[VEC], v10[VEC], v11[VEC];
  at(y0, x0, v00); at(y0, x1, v01); at(y1, x0, v10); at(y1, x1, v11);
#pragma unroll
  for (int i = 0; i < VEC; ++i) out[i] = ly0 * (lx0 * v00[i] + lx1 * v01[i]) + ly1 * (lx0 * v10[i] + lx1 * v11[i]);
}

// col[(b, oy, ox)][(ky * k + kx) * C + c] = virtual_input[b][oy * s - p + ky][ox * s - p + kx][c]  (zero outside)
template <int VEC>
__global__ __launch_bounds__(256) void im2col2d_kernel(Src2 S, float *__restrict__ col, int B, int H, int W, int k, int s, int p,
                                                       int Ho, int Wo, int act, int up, int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int C = S.C0 + S.C1, CV = C / VEC;
  const int c = (int)(idx % CV) * VEC;
  int64_t r = idx / CV;
  const int tap = (int)(r % (k * k));
  r /= k * k;
  const int ox = (int)(r % Wo), oy = (int)((r / Wo) % Ho), b = (int)(r / ((int64_t)Wo * Ho));
  const int Hv = up ? 2 * H : H, Wv = up ? 2 * W : W;
  const int y = oy * s - p + tap / k, x = ox * s - p + tap % k;
  float v[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) v[i] = 0.f;
  if (y >= 0 && y < Hv && x >= 0 && x < Wv) load_virtual<VEC>(S, b, y, x, c, H, W, act, up, v);
  float *o = col + r * ((int64_t)k * k * C) + (int64_t)tap * C + c;
  if constexpr (VEC == 4) *reinterpret_cast<float4 *>(o) = make_float4(v[0], v[1], v[2], v[3]);
  else o[0] = v[0];
}

// dvirt[b][y][x][c] = sum over the taps (ky, kx) with (y + p - ky) % s == 0 ... of dcol[(b, oy, ox)][(ky*k+kx)*C + c]
template <int VEC>
__global__ __launch_bounds__(256) void col2im2d_kernel(const float *__restrict__ dcol, float *__restrict__ dvirt, int B, int Hv, int Wv,
                                                       int C, int k, int s, int p, int Ho, int Wo, int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int CV = C / VEC;
  const int c = (int)(idx % CV) * VEC;
  int64_t r = idx / CV;
  const int x = (int)(r % Wv), y = (int)((r / Wv) % Hv), b = (int)(r / ((int64_t)Wv * Hv));
  float acc[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
  for (int ky = 0; ky < k; ++ky) {
    const int ty = y + p - ky;
    if (ty < 0 || ty % s != 0 || ty / s >= Ho) continue;
    for (int kx = 0; kx < k; ++kx) {
      const int tx = x + p - kx;
      if (tx < 0 || tx % s != 0 || tx / s >= Wo) continue;
      const float *q = dcol + (((int64_t)b * Ho + ty / s) * Wo + tx / s) * ((int64_t)k * k * C) + (int64_t)(ky * k + kx) * C + c;
      if constexpr (VEC == 4) {
        const float4 t = *reinterpret_cast<const float4 *>(q);
        acc[0] += t.x; acc[1] += t.y; acc[2] += t.z; acc[3] += t.w;
      } else {
        acc[0] += q[0];
      }
    }
  }
  float *o = dvirt + r * C + c;
  if constexpr (VEC == 4) *reinterpret_cast<float4 *>(o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  else o[0] = acc[0];
}

// gradient wrt the two sources: activation mask (and the upsample adjoint, gather form) applied to dvirt
template <int VEC>
__global__ __launch_bounds__(256) void conv2d_finish_bwd_kernel(const float *__restrict__ dvirt, Src2 S, float *__restrict__ d0,
                                                                float *__restrict__ d1, int B, int H, int W, int act, int up,
                                                                int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int C = S.C0 + S.C1, CV = C / VEC;
  const int c = (int)(idx % CV) * VEC;
  int64_t r = idx / CV;
  const int x = (int)(r % W), y = (int)((r / W) % H), b = (int)(r / ((int64_t)W * H));
  const bool first = c < S.C0;
  float *dst = first ? d0 : d1;
  if (dst == nullptr) return;
  const float *src = first ? S.p0 : S.p1;
  const int Cs = first ? S.C0 : S.C1, cc = first ? c : c - S.C0;
  float g[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) g[i] = 0.f;
  if (!up) {
    const float *q = dvirt + r * C + c;
#pragma unroll
    for (int i = 0; i < VEC; ++i) g[i] = q[i];
  } else {
    const int Hv = 2 * H, Wv = 2 * W;
    for (int yv = max(2 * y - 2, 0); yv <= min(2 * y + 2, Hv - 1); ++yv) {
      int y0, y1;
      float ly0, ly1;
      up_axis(yv, H, y0, y1, ly0, ly1);
      const float wy = (y0 == y ? ly0 : 0.f) + (y1 == y ? ly1 : 0.f);
      if (wy == 0.f) continue;
      for (int xv = max(2 * x - 2, 0); xv <= min(2 * x + 2, Wv - 1); ++xv) {
        int x0, x1;
        float lx0, lx1;
        up_axis(xv, W, x0, x1, lx0, lx1);
        const float wx = (x0 == x ? lx0 : 0.f) + (x1 == x ? lx1 : 0.f);
        if (wx == 0.f) continue;
        const float *q = dvirt + (((int64_t)b * Hv + yv) * Wv + xv) * C + c;
#pragma unroll
        for (int i = 0; i < VEC; ++i) g[i] += (wy * wx) * q[i];
      }
    }
  }
  const float *sp = src + r * Cs + cc;
  float *o = dst + r * Cs + cc;
#pragma unroll
  for (int i = 0; i < VEC; ++i) o[i] = g[i] * act_g(sp[i], act);
}

int check_block(const svr_conv2d_desc *d, const char *what) {
  SVR_CHECK(d && d->src0 && d->B > 0 && d->H > 0 && d->W > 0 && d->C0 > 0 && d->C1 >= 0, SVR_E_BADARG, "%s: bad descriptor", what);
  SVR_CHECK(d->C1 == 0 || d->src1, SVR_E_BADARG, "%s: C1 = %d without a second source", what, d->C1);
  SVR_CHECK((d->k == 4 && d->stride == 2) || (d->k == 3 && d->stride == 1), SVR_E_UNSUPPORTED, "%s: k=%d stride=%d (k4 s2 / k3 s1)", what,
            d->k, d->stride);
  SVR_CHECK(d->act >= 0 && d->act <= 2, SVR_E_BADARG, "%s: act %d", what, d->act);
  return SVR_OK;
}

void out_dims(const svr_conv2d_desc *d, int &Hv, int &Wv, int &Ho, int &Wo) {
  Hv = d->upsample ? 2 * d->H : d->H;
  Wv = d->upsample ? 2 * d->W : d->W;
  Ho = (Hv + 2 - d->k) / d->stride + 1;
  Wo = (Wv + 2 - d->k) / d->stride + 1;
}

}  // namespace

extern "C" int svr_conv2d_im2col(const svr_conv2d_desc *d, float *col, void *stream) {
  if (int rc = check_block(d, "conv2d_im2col")) return rc;
  SVR_CHECK(col, SVR_E_BADARG, "conv2d_im2col: null output");
  int Hv, Wv, Ho, Wo;
  out_dims(d, Hv, Wv, Ho, Wo);
  const int C = d->C0 + d->C1;
  const Src2 S{d->src0, d->src1, d->C0, d->C1};
  const bool vec = C % 4 == 0 && d->C0 % 4 == 0 && (((uintptr_t)d->src0 | (uintptr_t)d->src1 | (uintptr_t)col) & 15) == 0;
  const int64_t total = (int64_t)d->B * Ho * Wo * d->k * d->k * (vec ? C / 4 : C);
  SVR_CHECK(cdiv(total, 256) < (1LL << 31), SVR_E_UNSUPPORTED, "conv2d_im2col: %ld work items", (long)total);
  if (vec)
    hipLaunchKernelGGL(im2col2d_kernel<4>, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, S, col, d->B, d->H, d->W,
                       d->k, d->stride, 1, Ho, Wo, d->act, d->upsample, total);
  else
    hipLaunchKernelGGL(im2col2d_kernel<1>, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, S, col, d->B, d->H, d->W,
                       d->k, d->stride, 1, Ho, Wo, d->act, d->upsample, total);
  return launch_status("conv2d_im2col");
}

extern "C" int svr_conv2d_col2im(const svr_conv2d_desc *d, const float *dcol, float *dvirt, float *dsrc0, float *dsrc1, void *stream) {
  if (int rc = check_block(d, "conv2d_col2im")) return rc;
  SVR_CHECK(dcol && dvirt && (dsrc0 || dsrc1), SVR_E_BADARG, "conv2d_col2im: null pointer");
  int Hv, Wv, Ho, Wo;
  out_dims(d, Hv, Wv, Ho, Wo);
  const int C = d->C0 + d->C1;
  const Src2 S{d->src0, d->src1, d->C0, d->C1};
  hipStream_t s = (hipStream_t)stream;
  const bool vec = C % 4 == 0 && d->C0 % 4 == 0 &&
                   (((uintptr_t)d->src0 | (uintptr_t)d->src1 | (uintptr_t)dcol | (uintptr_t)dvirt | (uintptr_t)dsrc0 | (uintptr_t)dsrc1) & 15) == 0;
  const int64_t tv = (int64_t)d->B * Hv * Wv * (vec ? C / 4 : C), ts = (int64_t)d->B * d->H * d->W * (vec ? C / 4 : C);
  if (vec) {
    hipLaunchKernelGGL(col2im2d_kernel<4>, dim3((unsigned)cdiv(tv, 256)), dim3(256), 0, s, dcol, dvirt, d->B, Hv, Wv, C, d->k, d->stride, 1,
                       Ho, Wo, tv);
    hipLaunchKernelGGL(conv2d_finish_bwd_kernel<4>, dim3((unsigned)cdiv(ts, 256)), dim3(256), 0, s, (const float *)dvirt, S, dsrc0, dsrc1,
                       d->B, d->H, d->W, d->act, d->upsample, ts);
  } else {
    hipLaunchKernelGGL(col2im2d_kernel<1>, dim3((unsigned)cdiv(tv, 256)), dim3(256), 0, s, dcol, dvirt, d->B, Hv, Wv, C, d->k, d->stride, 1,
                       Ho, Wo, tv);
    hipLaunchKernelGGL(conv2d_finish_bwd_kernel<1>, dim3((unsigned)cdiv(ts, 256)), dim3(256), 0, s, (const float *)dvirt, S, dsrc0, dsrc1,
                       d->B, d->H, d->W, d->act, d->upsample, ts);
  }
  return launch_status("conv2d_col2im");
}

// ---- pieces of the implicit-GEMM path (conv2d_igemm.hip) ---------------------------------------------------------------
// V (B, Hv, Wv, C0 + C1) = [x2 bilinear upsample]( act( cat(src0, src1) ) ): the input of a decoder convolution, written once
// (the im2col gather with a 1 x 1 window)
extern "C" int svr_conv2d_virtual(const svr_conv2d_desc *d, float *V, void *stream) {
  SVR_CHECK(d && d->src0 && V && d->B > 0 && d->H > 0 && d->W > 0 && d->C0 > 0 && d->C1 >= 0, SVR_E_BADARG, "conv2d_virtual: bad descriptor");
  SVR_CHECK(d->C1 == 0 || d->src1, SVR_E_BADARG, "conv2d_virtual: C1 = %d without a second source", d->C1);
  SVR_CHECK(d->act >= 0 && d->act <= 2, SVR_E_BADARG, "conv2d_virtual: act %d", d->act);
  const int Hv = d->upsample ? 2 * d->H : d->H, Wv = d->upsample ? 2 * d->W : d->W, C = d->C0 + d->C1;
  const Src2 S{d->src0, d->src1, d->C0, d->C1};
  const bool vec = C % 4 == 0 && d->C0 % 4 == 0 && (((uintptr_t)d->src0 | (uintptr_t)d->src1 | (uintptr_t)V) & 15) == 0;
  const int64_t total = (int64_t)d->B * Hv * Wv * (vec ? C / 4 : C);
  SVR_CHECK(cdiv(total, 256) < (1LL << 31), SVR_E_UNSUPPORTED, "conv2d_virtual: %ld work items", (long)total);
  if (vec)
    hipLaunchKernelGGL(im2col2d_kernel<4>, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, S, V, d->B, d->H, d->W, 1, 1, 0,
                       Hv, Wv, d->act, d->upsample, total);
  else
    hipLaunchKernelGGL(im2col2d_kernel<1>, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, S, V, d->B, d->H, d->W, 1, 1, 0,
                       Hv, Wv, d->act, d->upsample, total);
  return launch_status("conv2d_virtual");
}

// gradients with respect to the two sources from dvirt (B, Hv, Wv, C0 + C1), the gradient of the convolution's (activated,
// upsampled, concatenated) input: upsample adjoint in gather form, activation derivative, split over the sources
extern "C" int svr_conv2d_finish_bwd(const svr_conv2d_desc *d, const float *dvirt, float *dsrc0, float *dsrc1, void *stream) {
  SVR_CHECK(d && d->src0 && dvirt && (dsrc0 || dsrc1) && d->B > 0 && d->H > 0 && d->W > 0 && d->C0 > 0 && d->C1 >= 0, SVR_E_BADARG,
            "conv2d_finish_bwd: bad descriptor");
  SVR_CHECK(d->C1 == 0 || d->src1, SVR_E_BADARG, "conv2d_finish_bwd: C1 = %d without a second source", d->C1);
  const int C = d->C0 + d->C1;
  const Src2 S{d->src0, d->src1, d->C0, d->C1};
  const bool vec = C % 4 == 0 && d->C0 % 4 == 0 &&
                   (((uintptr_t)d->src0 | (uintptr_t)d->src1 | (uintptr_t)dvirt | (uintptr_t)dsrc0 | (uintptr_t)dsrc1) & 15) == 0;
  const int64_t ts = (int64_t)d->B * d->H * d->W * (vec ? C / 4 : C);
  if (vec)
    hipLaunchKernelGGL(conv2d_finish_bwd_kernel<4>, dim3((unsigned)cdiv(ts, 256)), dim3(256), 0, (hipStream_t)stream, dvirt, S, dsrc0, dsrc1,
                       d->B, d->H, d->W, d->act, d->upsample, ts);
  else
    hipLaunchKernelGGL(conv2d_finish_bwd_kernel<1>, dim3((unsigned)cdiv(ts, 256)), dim3(256), 0, (hipStream_t)stream, dvirt, S, dsrc0, dsrc1,
                       d->B, d->H, d->W, d->act, d->upsample, ts);
  return launch_status("conv2d_finish_bwd");
}
