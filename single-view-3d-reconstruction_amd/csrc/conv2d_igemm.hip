// UNet convolutions as IMPLICIT GEMMs (reference model/unet.py:15-118, :121-186), gfx950.
//
// conv2d.hip (rounds 2-3) wrote the k*k-times larger patch matrix of every layer to HBM (twice per step: the backward
// recomputed it) and ran the point MLP's GEMMs on it.  Here the A operand of the GEMM is gathered straight from the
// channels-last activation: a k-step of the reduction is 16 channels of ONE tap, so a tile row's operand is one 64-byte piece
// of a pixel's channel vector (zero outside the image), LeakyReLU / ReLU and the decoder's channel concatenation are applied
// on the way into the f16 split -- the arithmetic and the tile / LDS / MFMA schedule are gemm_f16x3.hip's:
//
//   forward        Y[(b,oy,ox)][co]   = sum_{tap,c}  act(in)[b][oy s - p + ky][ox s - p + kx][c]  W[co][c][ky][kx] + bias[co]
//   backward-data  dIn[(b,y,x)][c]    = sum_{tap,co} dY[b][(y + p - ky) / s][(x + p - kx) / s][co] W[co][c][ky][kx]
//
// Backward-data of the stride-2 layers (k4 s2 p1) is FOUR stride-1 problems, one per parity class of the input pixel: a pixel
// (2 gy + py, 2 gx + px) is reached by the 2 x 2 taps ky = ((py + 1) & 1) + 2 jy from output pixel gy + py - jy -- no zero
// taps, no dilated tensor; grid.y is the class and the weight planes are stored per class.  (dY is the "scaled" operand of
// the f16x3s split: multiplied by 2^sx from its |max| on the way in, ops.py.)
//
// Deep layers have a handful of output tiles and a reduction of 2 304 - 4 608 (8 workgroups walking 290 k-steps: latency
// bound): grid.z splits the reduction, the partial tiles go to a workspace in the output's own layout and one elementwise
// pass sums them in a fixed order (+ bias) -- deterministic, no atomics.
//
// The x2 bilinear upsample in front of the decoder convolutions is NOT folded into the operand gather (4 loads + a lerp per
// operand, 9 times per virtual pixel): svr_conv2d_virtual writes the activated, upsampled, concatenated input once (1x the
// activation, where the patch matrix was 9x) and the convolution reads that.
#include "common.h"
#include "conv2d_virt.h"
#include "f16x3.h"
#include <algorithm>

using namespace svr;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int GK = 16;           // reduction elements per step (one MFMA depth)
constexpr int GLW = 8;           // dwords per LDS row of a plane: 16 halves
constexpr int GTM = 128;         // tile rows
constexpr int GAPLANE = GTM * GLW;

// (row, half h) -> dword offset of the 16-byte slot; see gemm_f16x3.hip (conflict-free ds_read_b128 on unpadded rows)
__device__ __forceinline__ int g_slot(int row, int h) { return (row * 2 + (h ^ ((row >> 3) & 1))) * 4; }
__device__ __forceinline__ f16x8 g_frag(const uint32_t *plane, int row, int lh) {
  union { uint4 q; f16x8 v; } f;
  f.q = *reinterpret_cast<const uint4 *>(plane + g_slot(row, lh));
  return f.v;
}

struct IgGeom {
  CvSrc S;            // the gathered operand: forward = the block's input, backward-data = dY (B, Ho, Wo, Cout)
  int B;
  int mode;           // 0 forward, 1 backward-data of a stride-1 layer, 2 backward-data of a stride-2 layer (4 classes)
  int k, stride, pad;
  int Qh, Qw;         // image the rows of the GEMM live on: forward = output pixels, backward-data = input pixels
  int Cpad;           // channels per tap in the reduction order (multiple of 16)
  int N;              // output channels of the GEMM
  int ksteps;         // k-steps per class = taps * Cpad / 16
  int ksplit;         // k-steps per reduction split (grid.z)
  int64_t class_stride;  // halves between the weight planes of two classes
  int64_t split_stride;  // floats between two partial outputs
};

struct IgClass {   // geometry of one row class (scalars)
  int Gh, Gw, sy, by, bx, step, tw, qs, qoy, qox;
};
__device__ __forceinline__ IgClass ig_class(const IgGeom &G, int cls) {
  IgClass c;
  if (G.mode == 0) {
    c = {G.Qh, G.Qw, G.stride, -G.pad, -G.pad, 1, G.k, 1, 0, 0};
  } else if (G.mode == 1) {
    c = {G.Qh, G.Qw, 1, G.pad, G.pad, -1, G.k, 1, 0, 0};
  } else {
    const int py = cls >> 1, px = cls & 1;
    c = {(G.Qh - py + 1) / 2, (G.Qw - px + 1) / 2, 1, py, px, -1, 2, 2, py, px};
  }
  return c;
}

// Y (or a partial of it) = A_gathered W^T.  TN = 128: 4 waves (2 x 2 of 64 x 64), TN = 64: 2 waves; k-step 16, two LDS stages,
// operands prefetched two k-steps ahead (gemm_f16x3.hip's schedule).
template <int TN, bool VEC4>
__global__ __launch_bounds__(2 * TN, TN == 128 ? 3 : 2) void conv2d_igemm_kernel(
    const IgGeom G, const uint16_t *__restrict__ W0, const uint32_t *__restrict__ amax_w, const uint32_t *__restrict__ amax_x,
    const float *__restrict__ bias, float *__restrict__ Y, float *__restrict__ partial, uint32_t *__restrict__ amax_y) {
  constexpr int NT = 2 * TN, BPLANE = TN * GLW, XPT = 512 / NT, STAGE = 2 * GAPLANE + 2 * BPLANE;
  __shared__ __attribute__((aligned(16))) uint32_t lds[2 * STAGE];
  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wr = wave / (TN / 64), wc = wave % (TN / 64);
  const int l31 = lane & 31, lh = lane >> 5;
  const int cls = blockIdx.y, split = blockIdx.z;
  const IgClass Cg = ig_class(G, cls);
  const int HW = Cg.Gh * Cg.Gw;
  const int64_t Mc = (int64_t)G.B * HW;
  const int64_t ntn = cdiv(G.N, TN), lidx = xcd_logical(blockIdx.x, gridDim.x);
  if (lidx >= ntn * cdiv(Mc, GTM)) return;
  const int64_t n0 = (lidx % ntn) * TN, m0 = (lidx / ntn) * GTM;
  const int kbeg = split * G.ksplit, kend = min(G.ksteps, kbeg + G.ksplit);
  const int cps = G.Cpad / GK;     // k-steps per tap
  const int Ctot = G.S.C0 + G.S.C1;

  // rows of this thread: pixel base of the sample and the input coordinates of tap (0, 0); rows past the class get
  // coordinates that fail every bounds test (their outputs are never stored)
  int rpix[XPT];
  int riy[XPT], rix[XPT];
#pragma unroll
  for (int i = 0; i < XPT; ++i) {
    const int64_t m = m0 + (t >> 2) + (NT / 4) * i;
    const int b = (int)(m / HW), rem = (int)(m % HW);
    const int gy = rem / Cg.Gw, gx = rem % Cg.Gw;
    rpix[i] = m < Mc ? b * G.S.H * G.S.W : 0;      // (host: B H W < 2^24)
    riy[i] = m < Mc ? gy * Cg.sy + Cg.by : -(1 << 24);
    rix[i] = gx * Cg.sy + Cg.bx;
  }
  int64_t wrow = n0 + (t >> 1);
  wrow = wrow < G.N ? wrow : G.N - 1;
  const uint16_t *wbase = W0 + cls * G.class_stride + wrow * 16 + (t & 1) * 8;
  const int64_t wstep = 2 * (int64_t)G.N * 16;    // halves per k-step (hi plane, lo plane)

  struct Regs {
    float4 x[XPT];
    uint4 w[2];
  };
  Regs ra, rb;
  // position of the NEXT load in the reduction order: channel step, tap column, tap row
  int lk = kbeg, lc = kbeg % cps, ljx = (kbeg / cps) % Cg.tw, ljy = (kbeg / cps) / Cg.tw;
  auto load = [&](Regs &r) {
    const int c = lc * GK + (t & 3) * 4;
    const bool live = lk < kend && c < Ctot;
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
      const int iy = live ? riy[i] + Cg.step * ljy : -1, ix = rix[i] + Cg.step * ljx;
      r.x[i] = cv_load4<VEC4>(G.S, rpix[i], iy, ix, c);
    }
    const uint16_t *wk = wbase + (int64_t)min(lk, kend - 1) * wstep;
    r.w[0] = *reinterpret_cast<const uint4 *>(wk);
    r.w[1] = *reinterpret_cast<const uint4 *>(wk + (int64_t)G.N * 16);
    ++lk;
    if (++lc == cps) {
      lc = 0;
      if (++ljx == Cg.tw) { ljx = 0; ++ljy; }
    }
  };
  const float sx = amax_x ? w_scale(amax_x[0], false) : 1.f;
  const int act = G.S.act;
  auto store = [&](const Regs &r, uint32_t *st) {
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
      const float4 v = cv_act4(r.x[i], act);
      uint32_t h0, l0, h1, l1;
      split_x(v.x * sx, v.y * sx, h0, l0);
      split_x(v.z * sx, v.w * sx, h1, l1);
      const int off = g_slot((t >> 2) + (NT / 4) * i, (t & 3) >> 1) + (t & 1) * 2;
      *reinterpret_cast<uint2 *>(st + off) = make_uint2(h0, h1);
      *reinterpret_cast<uint2 *>(st + GAPLANE + off) = make_uint2(l0, l1);
    }
    uint32_t *sb = st + 2 * GAPLANE;
    const int offb = g_slot(t >> 1, t & 1);
#pragma unroll
    for (int p = 0; p < 2; ++p) *reinterpret_cast<uint4 *>(sb + p * BPLANE + offb) = r.w[p];
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto step = [&](int cur, Regs &nxt, Regs &fre) {
    load(fre);
    const uint32_t *pa = lds + cur * STAGE, *pb = pa + 2 * GAPLANE;
    f16x8 a[2][2], b[3][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      a[0][i] = g_frag(pa, wr * 64 + i * 32 + l31, lh);
      a[1][i] = g_frag(pa + GAPLANE, wr * 64 + i * 32 + l31, lh);
      b[0][i] = g_frag(pb, wc * 64 + i * 32 + l31, lh);
      b[1][i] = g_frag(pb + BPLANE, wc * 64 + i * 32 + l31, lh);
      b[2][i] = scale_2m11(b[0][i]);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1][i], b[2][j], acc[i][j], 0, 0, 0);  // lo(x) hi(w)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][i], b[1][j], acc[i][j], 0, 0, 0);  // hi(x) lo(w)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][i], b[0][j], acc[i][j], 0, 0, 0);  // hi(x) hi(w)
      }
    store(nxt, lds + (cur ^ 1) * STAGE);
#pragma unroll
    for (int g = 0; g < 12; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
      __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);  // VALU
      __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // DS write
    }
    __syncthreads();
  };

  if (kbeg < kend) {
    load(ra);
    load(rb);
    store(ra, lds);
    __syncthreads();
    for (int k0 = kbeg; k0 < kend; k0 += 2) {
      step(0, rb, ra);
      if (k0 + 1 < kend) step(1, ra, rb);
    }
  }

  float inv = w_scale(amax_w[0], true);
  if (amax_x) inv *= w_scale(amax_x[0], true);
  float *out = partial ? partial + split * G.split_stride : Y;
  const float *bp = partial ? nullptr : bias;
  float vmax = 0.f;    // |max| of what this workgroup stores (amax_y: the next consumer's f16x3s scale; unsplit launches only)
  // address of tile row r's output pixel (class rows are strided over the image in mode 2)
  auto row_out = [&](int r, bool &ok) -> float * {
    const int64_t m = m0 + r;
    ok = m < Mc;
    const int b = (int)(m / HW), rem = (int)(m % HW);
    const int qy = (rem / Cg.Gw) * Cg.qs + Cg.qoy, qx = (rem % Cg.Gw) * Cg.qs + Cg.qox;
    return out + (((int64_t)b * G.Qh + qy) * G.Qw + qx) * G.N;
  };
  const bool coal = TN == 128 && G.N % 4 == 0 && (((uintptr_t)out) & 15) == 0;
  if (coal) {   // 128 x 128 tile through LDS in two passes of 64 rows: full 512-byte row segments per store instruction
    float *ct = reinterpret_cast<float *>(lds);
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      if (wr == pass) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              ct[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * 128 + wc * 64 + j * 32 + l31] = acc[i][j][r];
      }
      __syncthreads();
      const int c4 = (t & 31) * 4;
      const int64_t c = n0 + c4;
      float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (bp && c < G.N) b4 = *reinterpret_cast<const float4 *>(bp + c);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = (t >> 5) + 8 * i;
        bool ok;
        float *o = row_out(64 * pass + row, ok);
        if (ok && c < G.N) {
          float4 v = *reinterpret_cast<const float4 *>(ct + row * 128 + c4);
          v.x = v.x * inv + b4.x; v.y = v.y * inv + b4.y; v.z = v.z * inv + b4.z; v.w = v.w * inv + b4.w;
          vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
          *reinterpret_cast<float4 *>(o + c) = v;
        }
      }
      __syncthreads();
    }
  } else {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        bool ok;
        float *o = row_out(wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, ok);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int64_t n = n0 + wc * 64 + j * 32 + l31;
          if (ok && n < G.N) {
            const float v = acc[i][j][r] * inv + (bp ? bp[n] : 0.f);
            vmax = fmaxf(vmax, fabsf(v));
            o[n] = v;
          }
        }
      }
  }
  if (amax_y && !partial) svr_amax_publish(amax_y, vmax);   // (uniform)
}

// Y = sum over the splits of the partial outputs (+ bias), fixed order.  Grid-stride over float4 groups with at most 1 024
// workgroups: one |max| publication per wave at the END (with one workgroup per 256 outputs the publication -- tens of thousands
// of waves reading and raising one word -- was the kernel: 114 us for 4 M outputs).  total % 4 == 0 or the scalar tail below.
__global__ __launch_bounds__(256) void ig_reduce_kernel(const float *__restrict__ partial, int64_t split_stride, int splits,
                                                        const float *__restrict__ bias, float *__restrict__ Y, int64_t total, int N,
                                                        uint32_t *__restrict__ amax_y) {
  float vmax = 0.f;
  const bool v4 = (N & 3) == 0 && (split_stride & 3) == 0;    // (uniform) a float4 group stays inside one output row
  if (v4) {
    const int64_t groups = total >> 2;
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < groups; g += (int64_t)gridDim.x * 256) {
      float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int p = 0; p < splits; ++p) {
        const float4 t = *reinterpret_cast<const float4 *>(partial + p * split_stride + g * 4);
        s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
      }
      if (bias) {
        const float4 b = *reinterpret_cast<const float4 *>(bias + (g * 4) % N);
        s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
      }
      *reinterpret_cast<float4 *>(Y + g * 4) = s;
      vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(s.x), fabsf(s.y))), fmaxf(fabsf(s.z), fabsf(s.w)));
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
      float s = 0.f;
      for (int p = 0; p < splits; ++p) s += partial[p * split_stride + i];
      s += bias ? bias[i % N] : 0.f;
      Y[i] = s;
      vmax = fmaxf(vmax, fabsf(s));
    }
  }
  if (amax_y) svr_amax_publish(amax_y, vmax);   // (uniform; every wave arrives)
}

// max |W| of a contiguous weight -> amax[0] (zeroed first); one wave-level reduction and at most one atomic per wave
__global__ __launch_bounds__(256) void ig_amax_kernel(const float *__restrict__ W, int64_t n, uint32_t *__restrict__ amax) {
  float m = 0.f;
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4 *>(W)[i];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) m = fmaxf(m, fabsf(W[(n4 << 2) + threadIdx.x]));
  svr_amax_publish(amax, m);
}

// ---- weight planes ------------------------------------------------------------------------------------------------------
// forward: rows n = co, reduction k' = (ky k + kx) Cpad + c; [k-step][hi / lo][n][16 halves] (gemm_f16x3.hip's layout)
__device__ __forceinline__ void ig_split_fwd_item(const float *__restrict__ W, const uint32_t *__restrict__ amax, uint16_t *__restrict__ p0,
                                                  int Cout, int C, int k, int Cpad, int64_t idx) {
  const int64_t K = (int64_t)k * k * Cpad;   // idx over (n, k'/2)
  if (idx >= Cout * (K / 2)) return;
  const int64_t n = idx / (K / 2), kk = (idx % (K / 2)) * 2;
  const int tap = (int)(kk / Cpad), c = (int)(kk % Cpad);
  const float sc = w_scale(amax[0], false);
  const int64_t kk2 = (int64_t)k * k;
  const float w0 = c < C ? W[(n * C + c) * kk2 + tap] * sc : 0.f, w1 = c + 1 < C ? W[(n * C + c + 1) * kk2 + tap] * sc : 0.f;
  const uint32_t hi = pack_f16(w0, w1);
  const f32x2 h = unpack_f16(hi);
  const int64_t o = (((kk >> 4) * 2) * Cout + n) * 16 + (kk & 15);
  *reinterpret_cast<uint32_t *>(p0 + o) = hi;
  *reinterpret_cast<uint32_t *>(p0 + o + (int64_t)Cout * 16) = pack_f16(w0 - h.x, w1 - h.y);
}
__global__ __launch_bounds__(256) void ig_split_fwd_kernel(const float *__restrict__ W, const uint32_t *__restrict__ amax,
                                                           uint16_t *__restrict__ p0, int Cout, int C, int k, int Cpad) {
  ig_split_fwd_item(W, amax, p0, Cout, C, k, Cpad, (int64_t)blockIdx.x * 256 + threadIdx.x);
}
// backward-data: rows n = c, reduction k'' = j Copad + co over the taps j of a class (mode 1: all k*k taps, one class; mode 2:
// 2 x 2 taps per parity class, ky = ((py + 1) & 1) + 2 jy)
__device__ __forceinline__ void ig_split_bwd_item(const float *__restrict__ W, const uint32_t *__restrict__ amax, uint16_t *__restrict__ p0,
                                                  int Cout, int C, int k, int Copad, int mode, int64_t idx) {
  const int tw = mode == 2 ? 2 : k, ncls = mode == 2 ? 4 : 1;
  const int64_t K = (int64_t)tw * tw * Copad, per = C * (K / 2);
  if (idx >= ncls * per) return;
  const int cls = (int)(idx / per);
  const int64_t rem = idx % per, n = rem / (K / 2), kk = (rem % (K / 2)) * 2;
  const int j = (int)(kk / Copad), co = (int)(kk % Copad), jy = j / tw, jx = j % tw;
  const int ky = mode == 2 ? (((cls >> 1) + 1) & 1) + 2 * jy : jy, kx = mode == 2 ? (((cls & 1) + 1) & 1) + 2 * jx : jx;
  const float sc = w_scale(amax[0], false);
  const int64_t kk2 = (int64_t)k * k, tap = ky * k + kx;
  const float w0 = co < Cout ? W[((int64_t)co * C + n) * kk2 + tap] * sc : 0.f;
  const float w1 = co + 1 < Cout ? W[((int64_t)(co + 1) * C + n) * kk2 + tap] * sc : 0.f;
  const uint32_t hi = pack_f16(w0, w1);
  const f32x2 h = unpack_f16(hi);
  const int64_t o = cls * (2 * K * C) + (((kk >> 4) * 2) * C + n) * 16 + (kk & 15);
  *reinterpret_cast<uint32_t *>(p0 + o) = hi;
  *reinterpret_cast<uint32_t *>(p0 + o + (int64_t)C * 16) = pack_f16(w0 - h.x, w1 - h.y);
}
__global__ __launch_bounds__(256) void ig_split_bwd_kernel(const float *__restrict__ W, const uint32_t *__restrict__ amax,
                                                           uint16_t *__restrict__ p0, int Cout, int C, int k, int Copad, int mode) {
  ig_split_bwd_item(W, amax, p0, Cout, C, k, Copad, mode, (int64_t)blockIdx.x * 256 + threadIdx.x);
}

// ---- all layers of a network in three launches (one memset of the amax words, one |max| pass, one split pass): a UNet step
// spent 0.36 ms of GPU time and 64 launches in 16 x (memset, amax, forward split, backward split) -------------------------------
constexpr int IG_MAXL = 16;
struct IgPrepItem {
  const float *W;
  uint16_t *pf, *pb;     // forward / backward-data planes (pb null: no backward planes)
  int Cout, C, k, stride;
};
struct IgPrepTable {
  IgPrepItem it[IG_MAXL];
  int n;
};
__global__ __launch_bounds__(256) void ig_amax_many_kernel(const IgPrepTable T, uint32_t *__restrict__ amax) {
  const IgPrepItem L = T.it[blockIdx.y];
  const int64_t n = (int64_t)L.Cout * L.C * L.k * L.k, n4 = n >> 2;
  float m = 0.f;
  if ((((uintptr_t)L.W) & 15) == 0) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
      const float4 v = reinterpret_cast<const float4 *>(L.W)[i];
      m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) m = fmaxf(m, fabsf(L.W[(n4 << 2) + threadIdx.x]));
  } else {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) m = fmaxf(m, fabsf(L.W[i]));
  }
  svr_amax_publish(amax + blockIdx.y * 64, m);     // (one word per layer, 256 bytes apart)
}
__global__ __launch_bounds__(256) void ig_split_many_kernel(const IgPrepTable T, const uint32_t *__restrict__ amax) {
  const IgPrepItem L = T.it[blockIdx.y];
  const int Cpad = (L.C + 15) / 16 * 16, Copad = (L.Cout + 15) / 16 * 16;
  const int64_t fwd = (int64_t)L.Cout * L.k * L.k * Cpad / 2, bwd = L.pb ? (int64_t)L.C * L.k * L.k * Copad / 2 : 0;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < fwd + bwd; idx += (int64_t)gridDim.x * 256) {
    if (idx < fwd) ig_split_fwd_item(L.W, amax + blockIdx.y * 64, L.pf, L.Cout, L.C, L.k, Cpad, idx);
    else ig_split_bwd_item(L.W, amax + blockIdx.y * 64, L.pb, L.Cout, L.C, L.k, Copad, L.stride == 2 ? 2 : 1, idx - fwd);
  }
}

int pad16(int c) { return (c + 15) / 16 * 16; }
int64_t a256(int64_t x) { return (x + 255) / 256 * 256; }

int ig_check(const svr_conv2d_desc *d, int Cout, const char *what) {
  SVR_CHECK(d && d->src0 && d->B > 0 && d->H > 0 && d->W > 0 && d->C0 > 0 && d->C1 >= 0 && Cout > 0, SVR_E_BADARG, "%s: bad descriptor", what);
  SVR_CHECK(d->C1 == 0 || d->src1, SVR_E_BADARG, "%s: C1 = %d without a second source", what, d->C1);
  SVR_CHECK((d->k == 4 && d->stride == 2) || (d->k == 3 && d->stride == 1) || (d->k == 1 && d->stride == 1), SVR_E_UNSUPPORTED,
            "%s: k=%d stride=%d (k4 s2 p1 / k3 s1 p1 / k1 s1 p0)", what, d->k, d->stride);
  SVR_CHECK(d->act >= 0 && d->act <= 2, SVR_E_BADARG, "%s: act %d", what, d->act);
  SVR_CHECK(!d->upsample, SVR_E_UNSUPPORTED, "%s: the x2 upsample is materialised first (svr_conv2d_virtual)", what);
  SVR_CHECK((int64_t)d->B * d->H * d->W < (1 << 24), SVR_E_UNSUPPORTED, "%s: %ld pixels (row decode limit 2^24)", what,
            (long)d->B * d->H * d->W);
  return SVR_OK;
}

// reduction splits: enough workgroups to fill the chip (512 four-wave or 1 024 two-wave ones: two waves per SIMD -- a k-step is a
// dependent chain load -> split -> LDS -> MFMA, and with one wave per SIMD the 512-tile layers ran at 2.8 us per k-step;
// tools/exp/unet_target.sh), at least 8 k-steps each
int ig_splits(int64_t tiles, int ksteps, int n_out, int *ksplit) {
  static const int forced_target = getenv("SVR_IG_TARGET") ? atoi(getenv("SVR_IG_TARGET")) : 0;   // measurement switch
  const int target = forced_target > 0 ? forced_target : (n_out <= 64 ? 1024 : 512);
  int splits = (int)std::min<int64_t>(std::max<int64_t>(1, target / std::max<int64_t>(tiles, 1)), std::max(1, ksteps / 8));
  splits = std::min(splits, 64);
  static const int forced = getenv("SVR_IG_SPLITS") ? atoi(getenv("SVR_IG_SPLITS")) : 0;   // measurement switch
  if (forced > 0) splits = std::min(forced, ksteps);
  *ksplit = (int)cdiv(ksteps, splits);
  return (int)cdiv(ksteps, *ksplit);
}

template <bool VEC4>
void ig_launch(const IgGeom &G, int ncls, int64_t mtiles, int splits, const uint16_t *planes, const uint32_t *amax_w,
               const uint32_t *amax_x, const float *bias, float *Y, float *partial, uint32_t *amax_y, hipStream_t s) {
  if (G.N <= 64) {
    dim3 grid(xcd_grid(cdiv(G.N, 64) * mtiles), ncls, splits);
    hipLaunchKernelGGL((conv2d_igemm_kernel<64, VEC4>), grid, dim3(128), 0, s, G, planes, amax_w, amax_x, bias, Y, partial, amax_y);
  } else {
    dim3 grid(xcd_grid(cdiv(G.N, 128) * mtiles), ncls, splits);
    hipLaunchKernelGGL((conv2d_igemm_kernel<128, VEC4>), grid, dim3(256), 0, s, G, planes, amax_w, amax_x, bias, Y, partial, amax_y);
  }
}

bool vec4_ok(const CvSrc &S) {
  return S.C0 % 4 == 0 && S.C1 % 4 == 0 && (((uintptr_t)S.p0 | (uintptr_t)S.p1) & 15) == 0;
}

}  // namespace

// ---- C ABI --------------------------------------------------------------------------------------------------------------
extern "C" int64_t svr_conv2d_planes_bytes(int32_t Cout, int32_t C, int32_t k) {
  const int64_t fwd = 4LL * Cout * k * k * pad16(C), bwd = 4LL * C * k * k * pad16(Cout);
  return a256(fwd) + a256(bwd) + 256;
}

extern "C" int svr_conv2d_prepare(const float *W, int32_t Cout, int32_t C, int32_t k, int32_t stride, int32_t want_bwd,
                                  uint32_t *amax, void *planes, void *stream) {
  SVR_CHECK(W && amax && planes && Cout > 0 && C > 0, SVR_E_BADARG, "conv2d_prepare: null pointer / empty weight");
  SVR_CHECK((k == 4 && stride == 2) || (k == 3 && stride == 1) || (k == 1 && stride == 1 && !want_bwd), SVR_E_UNSUPPORTED,
            "conv2d_prepare: k=%d stride=%d (k = 1: forward planes only)", k, stride);
  hipStream_t s = (hipStream_t)stream;
  const int64_t numel = (int64_t)Cout * C * k * k;
  uint16_t *pf = (uint16_t *)(((uintptr_t)planes + 255) & ~(uintptr_t)255);
  uint16_t *pb = (uint16_t *)((char *)pf + a256(4LL * Cout * k * k * pad16(C)));
  (void)hipMemsetAsync(amax, 0, sizeof(uint32_t), s);
  if ((((uintptr_t)W) & 15) == 0)
    hipLaunchKernelGGL(ig_amax_kernel, dim3((unsigned)std::min<int64_t>(cdiv(numel, 4096), 256)), dim3(256), 0, s, W, numel, amax);
  else
    hipLaunchKernelGGL(w_amax_kernel, dim3((unsigned)std::min<int64_t>(cdiv(numel, 1024), 1024)), dim3(256), 0, s, W, numel, (int64_t)1,
                       numel, amax);
  const int Cpad = pad16(C), Copad = pad16(Cout);
  hipLaunchKernelGGL(ig_split_fwd_kernel, dim3((unsigned)cdiv((int64_t)Cout * k * k * Cpad / 2, 256)), dim3(256), 0, s, W, amax, pf,
                     Cout, C, k, Cpad);
  if (want_bwd)
    hipLaunchKernelGGL(ig_split_bwd_kernel, dim3((unsigned)cdiv((int64_t)C * k * k * Copad / 2, 256)), dim3(256), 0, s, W, amax, pb,
                       Cout, C, k, Copad, stride == 2 ? 2 : 1);
  return launch_status("conv2d_prepare");
}

// n <= 16 layers at once: W[i] (Cout[i], C[i], k[i], k[i]); planes[i] as for svr_conv2d_prepare; amax_base: n words 256 BYTES
// apart (word i at amax_base + 64 i), zeroed here.  Same planes, bit for bit, as n calls of svr_conv2d_prepare.
extern "C" int svr_conv2d_prepare_many(int32_t n, const float *const *W, const int32_t *Cout, const int32_t *C, const int32_t *k,
                                       const int32_t *stride, const int32_t *want_bwd, uint32_t *amax_base, void *const *planes,
                                       void *stream) {
  SVR_CHECK(n >= 1 && n <= IG_MAXL && W && Cout && C && k && stride && want_bwd && amax_base && planes, SVR_E_BADARG,
            "conv2d_prepare_many: 1..%d layers, no null pointers", IG_MAXL);
  IgPrepTable T{};
  T.n = n;
  int64_t maxitems = 0, maxnumel = 0;
  for (int i = 0; i < n; ++i) {
    SVR_CHECK(W[i] && planes[i] && Cout[i] > 0 && C[i] > 0, SVR_E_BADARG, "conv2d_prepare_many: layer %d", i);
    SVR_CHECK((k[i] == 4 && stride[i] == 2) || (k[i] == 3 && stride[i] == 1), SVR_E_UNSUPPORTED, "conv2d_prepare_many: layer %d k=%d stride=%d",
              i, k[i], stride[i]);
    uint16_t *pf = (uint16_t *)(((uintptr_t)planes[i] + 255) & ~(uintptr_t)255);
    uint16_t *pb = (uint16_t *)((char *)pf + a256(4LL * Cout[i] * k[i] * k[i] * pad16(C[i])));
    T.it[i] = IgPrepItem{W[i], pf, want_bwd[i] ? pb : nullptr, Cout[i], C[i], k[i], stride[i]};
    const int64_t items = (int64_t)Cout[i] * k[i] * k[i] * pad16(C[i]) / 2 + (want_bwd[i] ? (int64_t)C[i] * k[i] * k[i] * pad16(Cout[i]) / 2 : 0);
    maxitems = std::max(maxitems, items);
    maxnumel = std::max(maxnumel, (int64_t)Cout[i] * C[i] * k[i] * k[i]);
  }
  hipStream_t s = (hipStream_t)stream;
  (void)hipMemsetAsync(amax_base, 0, (size_t)n * 256, s);
  hipLaunchKernelGGL(ig_amax_many_kernel, dim3((unsigned)std::min<int64_t>(cdiv(maxnumel, 4096), 128), (unsigned)n), dim3(256), 0, s, T, amax_base);
  hipLaunchKernelGGL(ig_split_many_kernel, dim3((unsigned)std::min<int64_t>(cdiv(maxitems, 256), 1024), (unsigned)n), dim3(256), 0, s, T,
                     (const uint32_t *)amax_base);
  return launch_status("conv2d_prepare_many");
}

extern "C" int64_t svr_conv2d_workspace_bytes(const svr_conv2d_desc *d, int32_t Cout) {
  // partial outputs of the reduction splits, forward (B Ho Wo Cout) or backward-data (B H W C): at most 64 splits, and only
  // where there are fewer than 512 tiles -- bound by 64 * 512 tiles * 128 * 128 floats; sized exactly instead:
  if (!d) return 0;
  const int C = d->C0 + d->C1;
  const int pad = d->k == 1 ? 0 : 1, Ho = (d->H + 2 * pad - d->k) / d->stride + 1, Wo = (d->W + 2 * pad - d->k) / d->stride + 1;
  int ks;
  const int64_t mf = cdiv((int64_t)d->B * Ho * Wo, GTM) * cdiv(Cout, Cout <= 64 ? 64 : 128);
  const int sf = ig_splits(mf, d->k * d->k * pad16(C) / GK, Cout, &ks);
  const int64_t fwd = sf > 1 ? (int64_t)sf * d->B * Ho * Wo * Cout * 4 : 0;
  const int ncls = d->stride == 2 ? 4 : 1;
  const int64_t mb = cdiv((int64_t)d->B * cdiv(d->H, d->stride) * cdiv(d->W, d->stride), GTM) * cdiv(C, C <= 64 ? 64 : 128);
  const int sb = ig_splits(mb * ncls, (d->stride == 2 ? 4 : 9) * pad16(Cout) / GK, C, &ks);
  const int64_t bwd = (sb > 1 && d->k != 1) ? (int64_t)sb * d->B * d->H * d->W * C * 4 : 0;   // (k = 1: forward only)
  return a256(std::max(fwd, bwd)) + 256;
}

extern "C" int svr_conv2d_fwd(const svr_conv2d_desc *d, const void *planes, const uint32_t *amax_w, const float *bias, float *Y,
                              int32_t Cout, const uint32_t *amax_x, uint32_t *amax_y, void *workspace, void *stream) {
  if (int rc = ig_check(d, Cout, "conv2d_fwd")) return rc;
  SVR_CHECK(planes && amax_w && Y, SVR_E_BADARG, "conv2d_fwd: null pointer");
  const int pad = d->k == 1 ? 0 : 1;
  const int C = d->C0 + d->C1, Ho = (d->H + 2 * pad - d->k) / d->stride + 1, Wo = (d->W + 2 * pad - d->k) / d->stride + 1;
  SVR_CHECK(Ho > 0 && Wo > 0, SVR_E_BADSHAPE, "conv2d_fwd: empty output");
  IgGeom G{};
  G.S = CvSrc{d->src0, d->src1, d->C0, d->C1, d->H, d->W, d->act};
  G.B = d->B; G.mode = 0; G.k = d->k; G.stride = d->stride; G.pad = pad; G.Qh = Ho; G.Qw = Wo;
  G.Cpad = pad16(C); G.N = Cout; G.ksteps = d->k * d->k * G.Cpad / GK; G.class_stride = 0;
  const int64_t mtiles = cdiv((int64_t)d->B * Ho * Wo, GTM);
  const int splits = ig_splits(mtiles * cdiv(Cout, Cout <= 64 ? 64 : 128), G.ksteps, Cout, &G.ksplit);
  G.split_stride = (int64_t)d->B * Ho * Wo * Cout;
  float *partial = nullptr;
  if (splits > 1) {
    SVR_CHECK(workspace, SVR_E_BADARG, "conv2d_fwd: %d reduction splits need the workspace", splits);
    partial = (float *)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  }
  const uint16_t *pf = (const uint16_t *)(((uintptr_t)planes + 255) & ~(uintptr_t)255);
  hipStream_t s = (hipStream_t)stream;
  if (vec4_ok(G.S)) ig_launch<true>(G, 1, mtiles, splits, pf, amax_w, amax_x, bias, Y, partial, amax_y, s);
  else ig_launch<false>(G, 1, mtiles, splits, pf, amax_w, amax_x, bias, Y, partial, amax_y, s);
  if (splits > 1)
    hipLaunchKernelGGL(ig_reduce_kernel, dim3((unsigned)std::min<int64_t>(cdiv(G.split_stride, 1024), 1024)), dim3(256), 0, s,
                       (const float *)partial, G.split_stride, splits, bias, Y, G.split_stride, Cout, amax_y);
  return launch_status("conv2d_fwd");
}

// dIn (B, H, W, C0 + C1) = gradient with respect to the ACTIVATED (concatenated) input; svr_conv2d_finish_bwd applies the
// activation's derivative (and the upsample adjoint) and splits it over the two sources.
extern "C" int svr_conv2d_bwd_data(const svr_conv2d_desc *d, const void *planes, const uint32_t *amax_w, const float *dY,
                                   const uint32_t *amax_dy, int32_t Cout, float *dIn, void *workspace, void *stream) {
  if (int rc = ig_check(d, Cout, "conv2d_bwd_data")) return rc;
  SVR_CHECK(planes && amax_w && dY && dIn, SVR_E_BADARG, "conv2d_bwd_data: null pointer");
  const int C = d->C0 + d->C1, Ho = (d->H + 2 - d->k) / d->stride + 1, Wo = (d->W + 2 - d->k) / d->stride + 1;
  IgGeom G{};
  G.S = CvSrc{dY, nullptr, Cout, 0, Ho, Wo, 0};
  G.B = d->B; G.mode = d->stride == 2 ? 2 : 1; G.k = d->k; G.stride = d->stride; G.pad = 1; G.Qh = d->H; G.Qw = d->W;
  G.Cpad = pad16(Cout); G.N = C;
  const int taps = G.mode == 2 ? 4 : d->k * d->k, ncls = G.mode == 2 ? 4 : 1;
  G.ksteps = taps * G.Cpad / GK;
  G.class_stride = 2LL * taps * G.Cpad * C;
  const int64_t mtiles = cdiv((int64_t)d->B * cdiv(d->H, G.mode == 2 ? 2 : 1) * cdiv(d->W, G.mode == 2 ? 2 : 1), GTM);
  const int splits = ig_splits(mtiles * ncls * cdiv(C, C <= 64 ? 64 : 128), G.ksteps, C, &G.ksplit);
  G.split_stride = (int64_t)d->B * d->H * d->W * C;
  float *partial = nullptr;
  if (splits > 1) {
    SVR_CHECK(workspace, SVR_E_BADARG, "conv2d_bwd_data: %d reduction splits need the workspace", splits);
    partial = (float *)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  }
  const uint16_t *pf = (const uint16_t *)(((uintptr_t)planes + 255) & ~(uintptr_t)255);
  const uint16_t *pb = (const uint16_t *)((const char *)pf + a256(4LL * Cout * d->k * d->k * pad16(C)));
  hipStream_t s = (hipStream_t)stream;
  if (vec4_ok(G.S)) ig_launch<true>(G, ncls, mtiles, splits, pb, amax_w, amax_dy, nullptr, dIn, partial, nullptr, s);
  else ig_launch<false>(G, ncls, mtiles, splits, pb, amax_w, amax_dy, nullptr, dIn, partial, nullptr, s);
  if (splits > 1)
    hipLaunchKernelGGL(ig_reduce_kernel, dim3((unsigned)std::min<int64_t>(cdiv(G.split_stride, 1024), 1024)), dim3(256), 0, s,
                       (const float *)partial, G.split_stride, splits, (const float *)nullptr, dIn, G.split_stride, C, (uint32_t *)nullptr);
  return launch_status("conv2d_bwd_data");
}
