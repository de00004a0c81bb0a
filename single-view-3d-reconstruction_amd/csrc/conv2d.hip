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
#include "conv2d_virt.h"
#include <algorithm>

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

// ---- convolutions with 1..4 output channels (the UNet's last layer, model/unet.py:60: 64 -> channels_out) ------------------
// A GEMM tile would be 1/64 .. 1/32 full, so these run on the vector ALUs in plain f32 FMAs (no split, no amax): 16 lanes per
// pixel, one channel quad each.  Weights are staged in LDS as [tap][c][co]; limit CO * C * k * k <= 8192 floats.
namespace {
constexpr int SM_MAXW = 8192;

template <int CO>
__device__ __forceinline__ void sm_stage_w(const float *__restrict__ W, float *wl, int C, int kk) {
  for (int i = threadIdx.x; i < CO * C * kk; i += blockDim.x) {   // W[co][c][tap] -> wl[(tap C + c) CO + co]
    const int tap = i % kk, c = (i / kk) % C, co = i / (kk * C);
    wl[(tap * C + c) * CO + co] = W[i];
  }
  __syncthreads();
}

// y[(b,oy,ox)][co] = bias[co] + sum_{tap,c} act(in)[b][oy s - 1 + ky][ox s - 1 + kx][c] W[co][c][ky][kx]     (K = 3: s = 1, K = 4: s = 2)
template <int CO, int K, bool VEC4>
__global__ __launch_bounds__(256) void conv2d_small_fwd_kernel(const CvSrc S, const float *__restrict__ W, const float *__restrict__ bias,
                                                               float *__restrict__ Y, int B, int Ho, int Wo) {
  constexpr int ST = K == 4 ? 2 : 1;
  __shared__ float wl[SM_MAXW];
  const int C = S.C0 + S.C1;
  sm_stage_w<CO>(W, wl, C, K * K);
  const int64_t M = (int64_t)B * Ho * Wo;
  const int l16 = threadIdx.x & 15;
  // (a workgroup stages the weights once and then walks pixel groups: with one group per workgroup the staging was the kernel)
  for (int64_t pix0 = (int64_t)blockIdx.x * 16; pix0 < M; pix0 += (int64_t)gridDim.x * 16) {
  const int64_t pix = pix0 + (threadIdx.x >> 4);
  float acc[CO];
#pragma unroll
  for (int o = 0; o < CO; ++o) acc[o] = 0.f;
  if (pix < M) {
    const int ox = (int)(pix % Wo), oy = (int)((pix / Wo) % Ho), b = (int)(pix / ((int64_t)Wo * Ho));
    const int p0 = b * S.H * S.W;
    for (int c = l16 * 4; c < C; c += 64) {
      float4 v[K * K];
#pragma unroll
      for (int ky = 0; ky < K; ++ky)
#pragma unroll
        for (int kx = 0; kx < K; ++kx) v[ky * K + kx] = cv_load4<VEC4>(S, p0, oy * ST - 1 + ky, ox * ST - 1 + kx, c);   // all taps in flight
#pragma unroll
      for (int tap = 0; tap < K * K; ++tap) {
        const float4 a = cv_act4(v[tap], S.act);
        const float *w = wl + (tap * C + c) * CO;
        const float e[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (c + u < C) {
#pragma unroll
            for (int o = 0; o < CO; ++o) acc[o] = fmaf(e[u], w[u * CO + o], acc[o]);
          }
      }
    }
  }
#pragma unroll
  for (int o = 0; o < CO; ++o) {
#pragma unroll
    for (int d = 8; d > 0; d >>= 1) acc[o] += __shfl_xor(acc[o], d);
  }
  if (pix < M && l16 == 0) {
#pragma unroll
    for (int o = 0; o < CO; ++o) Y[pix * CO + o] = acc[o] + (bias ? bias[o] : 0.f);
  }
  }
}

// dIn[(b,y,x)][c] = sum over the taps that reach the pixel of dY[b][(y + 1 - ky) / s][(x + 1 - kx) / s][co] W[co][c][ky][kx]
template <int CO, int K>
__global__ __launch_bounds__(256) void conv2d_small_bwd_data_kernel(const float *__restrict__ dY, const float *__restrict__ W,
                                                                    float *__restrict__ dIn, int B, int H, int Wd, int C, int Ho, int Wo) {
  constexpr int ST = K == 4 ? 2 : 1;
  __shared__ float wl[SM_MAXW];
  const int CQ = (C + 3) / 4;
  sm_stage_w<CO>(W, wl, C, K * K);
  const int64_t total = (int64_t)B * H * Wd * CQ;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
  const int c = (int)(idx % CQ) * 4;
  const int64_t r = idx / CQ;
  const int x = (int)(r % Wd), y = (int)((r / Wd) % H), b = (int)(r / ((int64_t)Wd * H));
  float g[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ky = 0; ky < K; ++ky) {
    const int ty = y + 1 - ky;
    if (ty < 0 || (ST == 2 && (ty & 1)) || ty / ST >= Ho) continue;
#pragma unroll
    for (int kx = 0; kx < K; ++kx) {
      const int tx = x + 1 - kx;
      if (tx < 0 || (ST == 2 && (tx & 1)) || tx / ST >= Wo) continue;
      const float *q = dY + (((int64_t)b * Ho + ty / ST) * Wo + tx / ST) * CO;
      const float *w = wl + ((ky * K + kx) * C + c) * CO;
#pragma unroll
      for (int o = 0; o < CO; ++o) {
        const float d = q[o];
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (c + u < C) g[u] = fmaf(d, w[u * CO + o], g[u]);
      }
    }
  }
  float *o = dIn + r * C + c;
  if (c + 4 <= C && (C & 3) == 0) *reinterpret_cast<float4 *>(o) = make_float4(g[0], g[1], g[2], g[3]);
  else
    for (int u = 0; u < 4 && c + u < C; ++u) o[u] = g[u];
  }
}

// partial[block][(tap C + c) CO + co] = sum over the block's pixels of dY[m][co] act(in)[m + tap][c]; partial db behind it.
// A thread owns one (tap, channel quad) item; the 1024 threads of a workgroup are G groups of `ipg` items that take the
// block's pixels in turns, four at a time (four gathers in flight per thread), and are summed through LDS in group order.
template <int CO, int K, bool VEC4>
__global__ __launch_bounds__(1024) void conv2d_small_bwd_weight_kernel(const CvSrc S, const float *__restrict__ dY, float *__restrict__ part,
                                                                       int B, int Ho, int Wo, float rWo, float rHoWo, int rows_per_block,
                                                                       int ipg, int G) {
  constexpr int ST = K == 4 ? 2 : 1, U = 4;
  __shared__ float red[1024 * (4 * CO + 1)];
  const int C = S.C0 + S.C1, CQ = (C + 3) / 4, items = K * K * CQ;
  const int g = threadIdx.x / ipg, il = threadIdx.x % ipg, it = blockIdx.y * ipg + il;
  const int64_t M = (int64_t)B * Ho * Wo, m0 = (int64_t)blockIdx.x * rows_per_block, m1 = min(M, m0 + rows_per_block);
  const int tap = it / CQ, c = (it % CQ) * 4;
  const bool live = g < G && it < items;
  const int ty = tap / K - 1, tx = tap % K - 1, HoWo = Ho * Wo;
  float acc[CO][4];
  float dbs[CO];
#pragma unroll
  for (int o = 0; o < CO; ++o) {
    dbs[o] = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[o][u] = 0.f;
  }
  if (live) {
    for (int64_t mb = m0 + (int64_t)g * U; mb < m1; mb += (int64_t)G * U) {
      float d[U][CO];
      float4 v[U];
#pragma unroll
      for (int j = 0; j < U; ++j) {
        const int64_t m = mb + j;
        const bool ok = m < m1;
        int b, rem, oy, ox;
        cv_divmod((int)(ok ? m : m0), HoWo, rHoWo, b, rem);
        cv_divmod(rem, Wo, rWo, oy, ox);
#pragma unroll
        for (int o = 0; o < CO; ++o) d[j][o] = ok ? dY[m * CO + o] : 0.f;
        v[j] = ok ? cv_load4<VEC4>(S, b * S.H * S.W, oy * ST + ty, ox * ST + tx, c) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int j = 0; j < U; ++j) {
        const float4 a = cv_act4(v[j], S.act);
#pragma unroll
        for (int o = 0; o < CO; ++o) {
          acc[o][0] = fmaf(d[j][o], a.x, acc[o][0]); acc[o][1] = fmaf(d[j][o], a.y, acc[o][1]);
          acc[o][2] = fmaf(d[j][o], a.z, acc[o][2]); acc[o][3] = fmaf(d[j][o], a.w, acc[o][3]);
          dbs[o] += d[j][o];
        }
      }
    }
  }
  float *mine = red + threadIdx.x * (4 * CO + 1);       // (odd stride: conflict-free column walks)
#pragma unroll
  for (int o = 0; o < CO; ++o) {
#pragma unroll
    for (int u = 0; u < 4; ++u) mine[o * 4 + u] = acc[o][u];
  }
  mine[4 * CO] = 0.f;
  __syncthreads();
  const int64_t stride_p = (int64_t)K * K * C * CO + CO;
  float *p = part + blockIdx.x * stride_p;
  if (g == 0 && it < items) {
#pragma unroll
    for (int o = 0; o < CO; ++o)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float sum = 0.f;
        for (int q = 0; q < G; ++q) sum += red[(q * ipg + il) * (4 * CO + 1) + o * 4 + u];
        if (c + u < C) p[(tap * C + c + u) * CO + o] = sum;
      }
  }
  __syncthreads();
  if (il == 0 && g < G && blockIdx.y == 0) {   // db: every group summed dY over its own pixels
#pragma unroll
    for (int o = 0; o < CO; ++o) red[g * CO + o] = dbs[o];
  }
  __syncthreads();
  if (threadIdx.x < CO && blockIdx.y == 0) {
    float sum = 0.f;
    for (int q = 0; q < G; ++q) sum += red[q * CO + threadIdx.x];
    p[(int64_t)K * K * C * CO + threadIdx.x] = sum;
  }
}
// dW (CO, C, k, k) and db from the per-block partials, fixed order: 16 outputs per workgroup, sixteen lane groups walk the parts
__global__ __launch_bounds__(256) void conv2d_small_wgrad_reduce_kernel(const float *__restrict__ part, float *__restrict__ dW,
                                                                        float *__restrict__ db, int CO, int C, int kk, int parts) {
  __shared__ float red[256];
  const int total = kk * C * CO + CO, e = blockIdx.x * 16 + (threadIdx.x & 15), g = threadIdx.x >> 4;
  float s = 0.f;
  if (e < total)
    for (int p = g; p < parts; p += 16) s += part[(int64_t)p * total + e];
  red[threadIdx.x] = s;
  __syncthreads();
  if (g == 0 && e < total) {
    float v = 0.f;
    for (int q = 0; q < 16; ++q) v += red[q * 16 + (threadIdx.x & 15)];
    if (e >= kk * C * CO) {
      if (db) db[e - kk * C * CO] = v;
    } else {
      const int co = e % CO, c = (e / CO) % C, tap = e / (CO * C);
      dW[((int64_t)co * C + c) * kk + tap] = v;
    }
  }
}

int sm_check(const svr_conv2d_desc *d, int Cout, const char *what) {
  SVR_CHECK(d && d->src0 && d->B > 0 && d->H > 0 && d->W > 0 && d->C0 > 0 && d->C1 >= 0, SVR_E_BADARG, "%s: bad descriptor", what);
  SVR_CHECK(d->C1 == 0 || d->src1, SVR_E_BADARG, "%s: C1 = %d without a second source", what, d->C1);
  SVR_CHECK((d->k == 4 && d->stride == 2) || (d->k == 3 && d->stride == 1), SVR_E_UNSUPPORTED, "%s: k=%d stride=%d", what, d->k, d->stride);
  SVR_CHECK(!d->upsample, SVR_E_UNSUPPORTED, "%s: the x2 upsample is materialised first (svr_conv2d_virtual)", what);
  SVR_CHECK((int64_t)d->B * d->H * d->W < (1 << 24), SVR_E_UNSUPPORTED, "%s: %ld pixels (row decode limit 2^24)", what, (long)d->B * d->H * d->W);
  SVR_CHECK(Cout >= 1 && Cout <= 4 && (int64_t)Cout * (d->C0 + d->C1) * d->k * d->k <= SM_MAXW, SVR_E_UNSUPPORTED,
            "%s: Cout=%d C=%d (1..4 output channels, Cout*C*k*k <= %d)", what, Cout, d->C0 + d->C1, SM_MAXW);
  return SVR_OK;
}
bool sm_vec4(const svr_conv2d_desc *d) {
  return d->C0 % 4 == 0 && d->C1 % 4 == 0 && (((uintptr_t)d->src0 | (uintptr_t)d->src1) & 15) == 0;
}
int sm_rows_per_block(int64_t M) { return (int)std::max<int64_t>(64, cdiv(cdiv(M, 1024), 4) * 4); }
}  // namespace

extern "C" int svr_conv2d_small_supported(int32_t Cout, int32_t C, int32_t k) { return Cout >= 1 && Cout <= 4 && (int64_t)Cout * C * k * k <= SM_MAXW; }

#define SM_CO(CALL)              \
  switch (Cout) {                \
    case 1: { CALL(1); } break;  \
    case 2: { CALL(2); } break;  \
    case 3: { CALL(3); } break;  \
    default: { CALL(4); } break; \
  }

extern "C" int svr_conv2d_small_fwd(const svr_conv2d_desc *d, const float *W, const float *bias, float *Y, int32_t Cout, void *stream) {
  if (int rc = sm_check(d, Cout, "conv2d_small_fwd")) return rc;
  SVR_CHECK(W && Y, SVR_E_BADARG, "conv2d_small_fwd: null pointer");
  int Hv, Wv, Ho, Wo;
  out_dims(d, Hv, Wv, Ho, Wo);
  const CvSrc S{d->src0, d->src1, d->C0, d->C1, d->H, d->W, d->act};
  const int64_t M = (int64_t)d->B * Ho * Wo;
  const dim3 grid((unsigned)std::min<int64_t>(cdiv(M, 16), 2048));
  hipStream_t s = (hipStream_t)stream;
  const bool v4 = sm_vec4(d);
#define CALL(N)                                                                                                          \
  if (d->k == 3) {                                                                                                       \
    if (v4) hipLaunchKernelGGL((conv2d_small_fwd_kernel<N, 3, true>), grid, dim3(256), 0, s, S, W, bias, Y, d->B, Ho, Wo);   \
    else hipLaunchKernelGGL((conv2d_small_fwd_kernel<N, 3, false>), grid, dim3(256), 0, s, S, W, bias, Y, d->B, Ho, Wo);     \
  } else {                                                                                                               \
    if (v4) hipLaunchKernelGGL((conv2d_small_fwd_kernel<N, 4, true>), grid, dim3(256), 0, s, S, W, bias, Y, d->B, Ho, Wo);   \
    else hipLaunchKernelGGL((conv2d_small_fwd_kernel<N, 4, false>), grid, dim3(256), 0, s, S, W, bias, Y, d->B, Ho, Wo);     \
  }
  SM_CO(CALL)
#undef CALL
  return launch_status("conv2d_small_fwd");
}

extern "C" int svr_conv2d_small_bwd_data(const svr_conv2d_desc *d, const float *W, const float *dY, int32_t Cout, float *dIn, void *stream) {
  if (int rc = sm_check(d, Cout, "conv2d_small_bwd_data")) return rc;
  SVR_CHECK(W && dY && dIn, SVR_E_BADARG, "conv2d_small_bwd_data: null pointer");
  int Hv, Wv, Ho, Wo;
  out_dims(d, Hv, Wv, Ho, Wo);
  const int C = d->C0 + d->C1;
  const int64_t total = (int64_t)d->B * d->H * d->W * ((C + 3) / 4);
  SVR_CHECK((((uintptr_t)dIn) & 15) == 0, SVR_E_ALIGN, "conv2d_small_bwd_data: dIn must be 16-byte aligned");
  const unsigned gridx = (unsigned)std::min<int64_t>(cdiv(total, 256), 4096);
  hipStream_t s = (hipStream_t)stream;
#define CALL(N)                                                                                                                              \
  if (d->k == 3) hipLaunchKernelGGL((conv2d_small_bwd_data_kernel<N, 3>), dim3(gridx), dim3(256), 0, s, dY, W, dIn, d->B, d->H, d->W, C, Ho, Wo); \
  else hipLaunchKernelGGL((conv2d_small_bwd_data_kernel<N, 4>), dim3(gridx), dim3(256), 0, s, dY, W, dIn, d->B, d->H, d->W, C, Ho, Wo)
  SM_CO(CALL)
#undef CALL
  return launch_status("conv2d_small_bwd_data");
}

extern "C" int64_t svr_conv2d_small_bwd_weight_workspace(const svr_conv2d_desc *d, int32_t Cout) {
  if (!d) return 0;
  int Hv, Wv, Ho, Wo;
  out_dims(d, Hv, Wv, Ho, Wo);
  const int64_t M = (int64_t)d->B * Ho * Wo;
  const int64_t parts = cdiv(M, sm_rows_per_block(M));
  return parts * ((int64_t)d->k * d->k * (d->C0 + d->C1) * Cout + Cout) * 4 + 256;
}

extern "C" int svr_conv2d_small_bwd_weight(const svr_conv2d_desc *d, const float *dY, int32_t Cout, float *dW, float *db, void *workspace,
                                           void *stream) {
  if (int rc = sm_check(d, Cout, "conv2d_small_bwd_weight")) return rc;
  SVR_CHECK(dY && dW && workspace, SVR_E_BADARG, "conv2d_small_bwd_weight: null pointer");
  int Hv, Wv, Ho, Wo;
  out_dims(d, Hv, Wv, Ho, Wo);
  const int C = d->C0 + d->C1, kk = d->k * d->k;
  const CvSrc S{d->src0, d->src1, d->C0, d->C1, d->H, d->W, d->act};
  const int64_t M = (int64_t)d->B * Ho * Wo;
  const int rpb = sm_rows_per_block(M);
  const int parts = (int)cdiv(M, rpb);
  float *part = (float *)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  const int items = kk * ((C + 3) / 4);
  const int ipg = std::min(1024, (items + 63) / 64 * 64), G = std::max(1, 1024 / ipg);   // items per group, groups per workgroup
  const dim3 grid((unsigned)parts, (unsigned)cdiv(items, ipg));
  hipStream_t s = (hipStream_t)stream;
  const bool v4 = sm_vec4(d);
  const float rWo = 1.0f / (float)Wo, rHW = 1.0f / (float)(Ho * Wo);
#define CALL(N)                                                                                                                                  \
  if (d->k == 3) {                                                                                                                               \
    if (v4) hipLaunchKernelGGL((conv2d_small_bwd_weight_kernel<N, 3, true>), grid, dim3(1024), 0, s, S, dY, part, d->B, Ho, Wo, rWo, rHW, rpb, ipg, G);    \
    else hipLaunchKernelGGL((conv2d_small_bwd_weight_kernel<N, 3, false>), grid, dim3(1024), 0, s, S, dY, part, d->B, Ho, Wo, rWo, rHW, rpb, ipg, G);      \
  } else {                                                                                                                                       \
    if (v4) hipLaunchKernelGGL((conv2d_small_bwd_weight_kernel<N, 4, true>), grid, dim3(1024), 0, s, S, dY, part, d->B, Ho, Wo, rWo, rHW, rpb, ipg, G);    \
    else hipLaunchKernelGGL((conv2d_small_bwd_weight_kernel<N, 4, false>), grid, dim3(1024), 0, s, S, dY, part, d->B, Ho, Wo, rWo, rHW, rpb, ipg, G);      \
  }
  SM_CO(CALL)
#undef CALL
  const int total = kk * C * Cout + Cout;
  hipLaunchKernelGGL(conv2d_small_wgrad_reduce_kernel, dim3((unsigned)cdiv(total, 16)), dim3(256), 0, s, (const float *)part, dW, db, Cout, C, kk,
                     parts);
  return launch_status("conv2d_small_bwd_weight");
}
