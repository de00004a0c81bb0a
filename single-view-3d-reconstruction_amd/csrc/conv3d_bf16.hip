// 3x3x3 convolution on the bf16 matrix cores, gfx950: backward-data with the 3-term split ("bf16x3") and
// forward with the 6-product split ("bf16x6", f32-equivalent accuracy: x = hi + mid + lo, see gemm_bf16x6.hip).
//
// Same brick structure as conv3d_brick_kernel (conv3d.hip): one workgroup = one 4x4x8 output brick, wave w =
// z-slice w (32 voxels = one MFMA row tile); per 32-channel chunk the input halo tile (6x6x10 voxels) is staged
// in LDS once -- here split on the way in into hi/mid bf16 planes [voxel][ci] -- and serves all 27 taps; the
// tap's weight slice [co][ci] (pre-split planes, packed once per step) is double buffered in LDS.
//   x = hi + mid, products hi*hi + hi*mid + mid*hi, f32 accumulation in v_mfma_f32_32x32x16_bf16:
//   ~1.5e-5 relative per product (see gemm_bf16x3.hip for why that is harmless in the BACKWARD pass), at a
//   fraction of the exact-f32 MFMA cycles (6 x 32 cycles per 32x32x32 block instead of 16 x 64).
// bf16x3 is used only for dIn = conv^T(dOut).  The production forward is the f16 split (F16 = true: f16x3.h /
// gemm_f16x3.hip, f32-level accuracy for |x| < 65504 at three products); bf16x6 (three planes, six products, any
// f32 range) stays selectable.  Both forwards are held to the exact-f32 kernel's gate.
#include "common.h"
#include <cstdlib>
#include "f16x3.h"

using namespace svr;

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvShape {
  int B, D, H, W, Ci, Co;
};

constexpr int BRZ = 4, BRY = 4, BRX = 8;
constexpr int HLZ = BRZ + 2, HLY = BRY + 2, HLX = BRX + 2, HLV = HLZ * HLY * HLX;

__device__ __forceinline__ void split2(float x0, float x1, uint32_t &hi, uint32_t &mid) {
  f32x2 v = {x0, x1};
  bf16x2 h = __builtin_convertvector(v, bf16x2);
  hi = __builtin_bit_cast(uint32_t, h);
  f32x2 r = {x0 - __uint_as_float(hi << 16), x1 - __uint_as_float(hi & 0xffff0000u)};
  bf16x2 m = __builtin_convertvector(r, bf16x2);
  mid = __builtin_bit_cast(uint32_t, m);
}

__device__ __forceinline__ void split3(float x0, float x1, uint32_t &hi, uint32_t &mid, uint32_t &lo) {
  split2(x0, x1, hi, mid);
  f32x2 r = {(x0 - __uint_as_float(hi << 16)) - __uint_as_float(mid << 16),
             (x1 - __uint_as_float(hi & 0xffff0000u)) - __uint_as_float(mid & 0xffff0000u)};
  bf16x2 l = __builtin_convertvector(r, bf16x2);
  lo = __builtin_bit_cast(uint32_t, l);
}

// LDS layout of the operand tiles (round 3).  A fragment = 8 consecutive k of one row = 16 bytes, and it has to come out
// of ONE ds_read_b128 (256 B/clk; the 8-byte aligned rows of the first version made it a ds_read2_b64: 128 B/clk, and at
// one fragment KB per MFMA these kernels saturated exactly that).  ds_read_b128 serves a wave in four groups of 16 lanes,
// {0-3,12-15,20-27}, {4-11,16-19,28-31} and the same + 32 (MI355X_MICROARCH.md "LDS"), conflict free when the 16 lanes hit
// 16 different 16-byte slots mod 16.  Rows are SL = CK/8 slots wide without padding, slot = SL*row + (sub ^ g(row)) with
// g = bit 3 of the row (SL = 2) or bits 2-3 (SL = 4):
//   weight tile: row = output column = lane & 31 -> the rows of a group pair up at distances 8 and 24;
//   halo tile:   x pitch HP = 12 voxels and the lane -> voxel map of tile_vy / tile_vx below: a group holds two complete
//                x-rows of 8 voxels, y and y + 2, i.e. rows r .. r+7 and r+24 .. r+31 for EVERY tap shift (the shift adds a
//                constant to all rows; an odd multiple of 8 flips bit 3 whatever the constant is).
constexpr int HP = 12;
template <int SL>
__device__ __forceinline__ int slot_dw(int row, int sub) {  // dword offset of slot `sub` of a row
  static_assert(SL == 2 || SL == 4, "16- or 32-channel chunks");
  const int g = SL == 2 ? (row >> 3) & 1 : (row >> 2) & 3;
  return (row * SL + (sub ^ g)) * 4;
}
// voxel (y, x) of the 4 x 8 tile held by MFMA row m (= lane & 31 of the A operand, = the accumulator's row index)
__device__ __forceinline__ int tile_vy(int m) { return (0x32230110u >> ((m >> 2) * 4)) & 3; }
__device__ __forceinline__ int tile_vx(int m) { return ((m >> 3) & 1) * 4 + (m & 3); }

__device__ __forceinline__ bf16x8 read_frag(const uint32_t *p) {  // 8 bf16 at a 16-byte aligned address
  union { uint4 q; bf16x8 v; } f;
  f.q = *reinterpret_cast<const uint4 *>(p);
  return f.v;
}

__device__ __forceinline__ f16x8 read_frag_h(const uint32_t *p) {  // the same 16 bytes as 8 halves
  union { uint4 q; f16x8 v; } f;
  f.q = *reinterpret_cast<const uint4 *>(p);
  return f.v;
}

// W (Co,Ci,3,3,3) f32 -> forward planes of the f16 split [2][27][Co][Ci]: hi(W 2^s), lo(W 2^s)  (f16x3.h / gemm_f16x3.hip)
__global__ void pack_fwd_planes_f16_kernel(const float *__restrict__ W, const uint32_t *__restrict__ amax,
                                           uint16_t *__restrict__ p0, uint16_t *__restrict__ p1, int Ci, int Co) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;  // over (tap, co, ci/2)
  if (idx >= 27 * Co * (Ci / 2)) return;
  const int ci = (idx % (Ci / 2)) * 2;
  const int co = (idx / (Ci / 2)) % Co;
  const int tap = idx / ((Ci / 2) * Co);
  const float sc = w_scale(amax[0], false);
  const float w0 = W[((size_t)co * Ci + ci) * 27 + tap] * sc, w1 = W[((size_t)co * Ci + ci + 1) * 27 + tap] * sc;
  const uint32_t hi = pack_f16(w0, w1);
  const f32x2 h = unpack_f16(hi);
  const size_t o = ((size_t)tap * Co + co) * Ci + ci;
  *reinterpret_cast<uint32_t *>(p0 + o) = hi;
  *reinterpret_cast<uint32_t *>(p1 + o) = pack_f16(w0 - h.x, w1 - h.y);
}

// W (Co,Ci,3,3,3) f32 -> backward-data planes [2][27][Ci][Co] bf16: row = ORIGINAL input channel (the output
// channel of the transposed conv), k = original output channel, taps flipped.
__global__ void pack_bwd_planes_kernel(const float *__restrict__ W, uint16_t *__restrict__ hi, uint16_t *__restrict__ mid,
                                       int Ci, int Co) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;  // over (tap', ci, co/2)
  if (idx >= 27 * Ci * (Co / 2)) return;
  const int co = (idx % (Co / 2)) * 2;
  const int ci = (idx / (Co / 2)) % Ci;
  const int tp = idx / ((Co / 2) * Ci);
  uint32_t h, m;
  split2(W[((size_t)co * Ci + ci) * 27 + (26 - tp)], W[((size_t)(co + 1) * Ci + ci) * 27 + (26 - tp)], h, m);
  const size_t o = ((size_t)tp * Ci + ci) * Co + co;
  *reinterpret_cast<uint32_t *>(hi + o) = h;
  *reinterpret_cast<uint32_t *>(mid + o) = m;
}

// The same planes for the scaled f16 split ("f16x3s" backward-data): hi(W 2^s), lo(W 2^s) in f16, [2][27][Ci][Co]
__global__ void pack_bwd_planes_f16_kernel(const float *__restrict__ W, const uint32_t *__restrict__ amax,
                                           uint16_t *__restrict__ p0, uint16_t *__restrict__ p1, int Ci, int Co) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;  // over (tap', ci, co/2)
  if (idx >= 27 * Ci * (Co / 2)) return;
  const int co = (idx % (Co / 2)) * 2;
  const int ci = (idx / (Co / 2)) % Ci;
  const int tp = idx / ((Co / 2) * Ci);
  const float sc = w_scale(amax[0], false);
  const float w0 = W[((size_t)co * Ci + ci) * 27 + (26 - tp)] * sc, w1 = W[((size_t)(co + 1) * Ci + ci) * 27 + (26 - tp)] * sc;
  const uint32_t hi = pack_f16(w0, w1);
  const f32x2 h = unpack_f16(hi);
  const size_t o = ((size_t)tp * Ci + ci) * Co + co;
  *reinterpret_cast<uint32_t *>(p0 + o) = hi;
  *reinterpret_cast<uint32_t *>(p1 + o) = pack_f16(w0 - h.x, w1 - h.y);
}

// W (Co,Ci,3,3,3) f32 -> forward planes [3][27][Co][Ci] bf16 (row = output channel, k = input channel)
__global__ void pack_fwd_planes_kernel(const float *__restrict__ W, uint16_t *__restrict__ p0, uint16_t *__restrict__ p1,
                                       uint16_t *__restrict__ p2, int Ci, int Co) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;  // over (tap, co, ci/2)
  if (idx >= 27 * Co * (Ci / 2)) return;
  const int ci = (idx % (Ci / 2)) * 2;
  const int co = (idx / (Ci / 2)) % Co;
  const int tap = idx / ((Ci / 2) * Co);
  uint32_t h, m, l;
  split3(W[((size_t)co * Ci + ci) * 27 + tap], W[((size_t)co * Ci + ci + 1) * 27 + tap], h, m, l);
  const size_t o = ((size_t)tap * Co + co) * Ci + ci;
  *reinterpret_cast<uint32_t *>(p0 + o) = h;
  *reinterpret_cast<uint32_t *>(p1 + o) = m;
  *reinterpret_cast<uint32_t *>(p2 + o) = l;
}

// out(B,D,H,W,NOUT) = epi( sum_{tap,k} in[voxel+tap][k] * P[tap][n][k] ), K = s.Ci input channels of THIS call,
// NOUT = s.Co.  planes: hi then mid, each [27][NOUT][K] bf16.
// NP = 2: products mid*hi + hi*mid + hi*hi (bf16x3);  NP = 3: the six products of bf16x6;
// F16 (NP = 2): the f16 split of gemm_f16x3.hip -- planes hi / lo*2^11 of the input, hi / lo of the normalised
// weights, products lo'(x) wq + hi(x) lo(w) + hi(x) hi(w) with wq = hi(w) 2^-11 made in registers, epilogue * 2^-s.
// VT = z-slices (32-voxel row tiles) per wave: with VT = 2 the brick is 8x4x8 and every weight fragment read from LDS
// feeds two MFMA tiles (these kernels are LDS-bandwidth bound: 1.33 -> 1.0 KB of fragment reads per MFMA at Co = 32).
template <int CK, int TNB, int NP, bool F16 = false, int VT = 1>
__global__ __launch_bounds__(256) void conv3d_brick_x3_kernel(const float *__restrict__ in,
                                                              const uint16_t *__restrict__ P0,
                                                              int64_t plane_stride, const float *__restrict__ bias,
                                                              float *__restrict__ out, const float *__restrict__ mask,
                                                              ConvShape s, int nbz, int nby, int nbx, int mode,
                                                              const uint32_t *__restrict__ amax = nullptr,
                                                              double *__restrict__ spart = nullptr,
                                                              const uint32_t *__restrict__ amax_in = nullptr,
                                                              uint32_t *__restrict__ amax_out = nullptr) {
  // amax_in (F16 only; the backward-data product of the scaled f16 split, "f16x3s"): |max| of the input (a gradient): it is
  // multiplied by the exact power of two that brings that into [2^13, 2^14) on the way into the split, the epilogue divides
  // again; amax_out: |max| of what this launch stored (the next layer's scale without a pass over the tensor)
  float sin = 1.f;
  if constexpr (F16) sin = amax_in ? w_scale(amax_in[0], false) : 1.f;
  float vmax = 0.f;
  constexpr int BZ = BRZ * VT, HV = (BZ + 2) * HLY * HLX;  // brick depth and halo voxels of this instantiation
  constexpr int SL = CK / 8, XW = SL * 4;    // 16-byte slots / dwords per LDS row (CK 16-bit values, no padding: slot_dw)
  constexpr int HR = (BZ + 2) * HLY * HP;    // halo rows in LDS (x pitch HP)
  constexpr int NC = TNB * 32;               // output columns of this workgroup
  constexpr int TG = TNB == 1 ? 3 : 1;       // taps per barrier
  constexpr int KS = CK / 16;                // MFMA k sub-steps per chunk
  constexpr int PIECES = TG * NP * NC * (CK / 8);  // 16-byte pieces per weight group
  constexpr int WPT = (PIECES + 255) / 256;
  __shared__ __attribute__((aligned(16))) uint32_t sh[NP][HR * XW];        // halo tile, hi / mid (/ lo) planes: [voxel row][k]
  __shared__ __attribute__((aligned(16))) uint32_t sw[2][TG][NP][NC * XW]; // weight slices: [buffer][tap][plane][n][k]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  int64_t q = blockIdx.x;
  const int bx = (int)(q % nbx); q /= nbx;
  const int by = (int)(q % nby); q /= nby;
  const int bz = (int)(q % nbz);
  const int64_t b = q / nbz;
  const int z0 = bz * BZ, y0 = by * BRY, x0 = bx * BRX;
  const int n0 = blockIdx.y * NC;
  const float *inb = in + b * (int64_t)s.D * s.H * s.W * s.Ci;
  const int hrow = ((wave + 1) * HLY + tile_vy(l31) + 1) * HP + tile_vx(l31) + 1;  // this lane's voxel (row) in the halo tile
  f32x16 acc[VT][TNB];  // tile v of this wave = z-slice wave + 4 v
#pragma unroll
  for (int v = 0; v < VT; ++v)
#pragma unroll
    for (int j = 0; j < TNB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[v][j][r] = 0.f;

  // weight slices travel global -> registers -> LDS two steps ahead of their use (two register sets, two LDS
  // buffers): one step of MFMAs (0.2-0.3 us at Co = 32) is shorter than an L2 round trip
  uint2 wregA[WPT][2], wregB[WPT][2];
  auto wload = [&](uint2 (&wreg)[WPT][2], int k0, int tap0) {
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      const int idx = t + 256 * i;
      const int part = idx % (CK / 8), row = (idx / (CK / 8)) % NC, pl = (idx / (CK / 8 * NC)) % NP, tg = idx / (CK / 8 * NC * NP);
      const bool ok = idx < PIECES && n0 + row < s.Co;
      const uint16_t *base = P0 + (ok ? pl : 0) * plane_stride;
      const uint2 *p = reinterpret_cast<const uint2 *>(base + ((size_t)(tap0 + (ok ? tg : 0)) * s.Co + (ok ? n0 + row : 0)) * s.Ci + k0 + part * 8);
#if defined(SVR_CONV_EXP) && SVR_CONV_EXP == 4   // measurement build: no weight loads
      wreg[i][0] = make_uint2((uint32_t)idx, 0x3c003c00u);
      wreg[i][1] = make_uint2(0x3c003c00u, (uint32_t)(uintptr_t)p);
#else
      wreg[i][0] = p[0];  // unconditional (clamped to piece 0 when out of range: those columns are never stored)
      wreg[i][1] = p[1];
#endif
    }
  };
  auto wstore = [&](const uint2 (&wreg)[WPT][2], int buf) {
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      const int idx = t + 256 * i;
      if (idx < PIECES) {
        const int part = idx % (CK / 8), row = (idx / (CK / 8)) % NC, pl = (idx / (CK / 8 * NC)) % NP, tg = idx / (CK / 8 * NC * NP);
        *reinterpret_cast<uint4 *>(&sw[buf][tg][pl][slot_dw<SL>(row, part)]) = make_uint4(wreg[i][0].x, wreg[i][0].y, wreg[i][1].x, wreg[i][1].y);
      }
    }
  };

  constexpr int STEPS = 27 / TG;
  for (int k0 = 0; k0 < s.Ci; k0 += CK) {
    wload(wregA, k0, 0);
    wload(wregB, k0, TG);
    if (k0 > 0) __syncthreads();  // previous chunk's tiles are no longer read
    // Halo tile: all loads of the chunk are issued first, unconditionally and from clamped coordinates (a load inside
    // a branch is waited for at the end of the branch, one full memory latency per iteration), zeroed afterwards.
    constexpr int HIT = (HV * (CK / 4) + 255) / 256;
    float4 hreg[HIT];
    uint64_t hok = 0;
#pragma unroll
    for (int i = 0; i < HIT; ++i) {
      const int idx = min(t + 256 * i, HV * (CK / 4) - 1);
      const int hv = idx / (CK / 4), c4 = (idx % (CK / 4)) * 4;
      const int hx = hv % HLX, hy = (hv / HLX) % HLY, hz = hv / (HLX * HLY);
      const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = x0 + hx - 1;
      if (gz >= 0 && gz < s.D && gy >= 0 && gy < s.H && gx >= 0 && gx < s.W) hok |= 1ull << i;
      const int cz = min(max(gz, 0), s.D - 1), cy = min(max(gy, 0), s.H - 1), cx = min(max(gx, 0), s.W - 1);
#if defined(SVR_CONV_EXP) && SVR_CONV_EXP == 3   // measurement build: no halo loads (a constant instead)
      hreg[i] = make_float4((float)cz, 1.f, 2.f, 3.f);
#else
      hreg[i] = *reinterpret_cast<const float4 *>(inb + (((int64_t)cz * s.H + cy) * s.W + cx) * s.Ci + k0 + c4);
#endif
    }
#pragma unroll
    for (int i = 0; i < HIT; ++i) {
      const int idx = t + 256 * i;
      if (idx >= HV * (CK / 4)) break;
      const int hv = idx / (CK / 4), c4 = (idx % (CK / 4)) * 4;
      const int hd = slot_dw<SL>((hv / HLX) * HP + hv % HLX, c4 / 8) + (c4 % 8) / 2;   // (hz*HLY + hy)*HP + hx; half a slot
      const bool ok = (hok >> i) & 1ull;
#if defined(SVR_CONV_EXP) && SVR_CONV_EXP == 5   // measurement build: the halo loads are issued but nobody waits for them here
      const float4 v = make_float4(ok ? (float)hv : 0.f, 1.f, 2.f, ok ? 3.f : 0.f);
#else
      const float4 v = make_float4(ok ? hreg[i].x : 0.f, ok ? hreg[i].y : 0.f, ok ? hreg[i].z : 0.f, ok ? hreg[i].w : 0.f);
#endif
      uint32_t h0, m0, l0 = 0, h1, m1, l1 = 0;
      if constexpr (NP == 3) {
        split3(v.x, v.y, h0, m0, l0);
        split3(v.z, v.w, h1, m1, l1);
        *reinterpret_cast<uint2 *>(&sh[NP - 1][hd]) = make_uint2(l0, l1);
      } else if constexpr (F16) {
        split_x(v.x * sin, v.y * sin, h0, m0);
        split_x(v.z * sin, v.w * sin, h1, m1);
      } else {
        split2(v.x, v.y, h0, m0);
        split2(v.z, v.w, h1, m1);
      }
      *reinterpret_cast<uint2 *>(&sh[0][hd]) = make_uint2(h0, h1);
      *reinterpret_cast<uint2 *>(&sh[1][hd]) = make_uint2(m0, m1);
    }
    wstore(wregA, 0);
    __syncthreads();
    // step st: MFMAs on LDS buffer st & 1; `nxt` holds step st+1 (in flight since step st-1), `fre` is refilled with st+2
    auto step = [&](int st, uint2 (&nxt)[WPT][2], uint2 (&fre)[WPT][2]) {
      const int buf = st & 1, tap0 = st * TG;
      if (st + 2 < STEPS) wload(fre, k0, tap0 + 2 * TG);
#pragma unroll
      for (int tg = 0; tg < TG; ++tg) {
        const int tap = tap0 + tg;
#if defined(SVR_CONV_EXP) && SVR_CONV_EXP == 1   // measurement build: the three dx taps read ONE fragment (wrong results)
        const int arow = hrow + ((tap / 9 - 1) * HLY + ((tap / 3) % 3 - 1)) * HP;
#else
        const int arow = hrow + ((tap / 9 - 1) * HLY + ((tap / 3) % 3 - 1)) * HP + (tap % 3 - 1);
#endif
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const int sub = ks * 2 + lh;
          int aoff[VT], woff[TNB];
#pragma unroll
          for (int v = 0; v < VT; ++v) aoff[v] = slot_dw<SL>(arow + v * BRZ * HLY * HP, sub);
#pragma unroll
          for (int j = 0; j < TNB; ++j) woff[j] = slot_dw<SL>(j * 32 + l31, sub);
          if constexpr (F16) {
            f16x8 xh[VT], xl[VT];
#pragma unroll
            for (int v = 0; v < VT; ++v) {
              xh[v] = read_frag_h(&sh[0][aoff[v]]);
              xl[v] = read_frag_h(&sh[1][aoff[v]]);
            }
#pragma unroll
            for (int j = 0; j < TNB; ++j) {
              const f16x8 wh = read_frag_h(&sw[buf][tg][0][woff[j]]);
              const f16x8 wl = read_frag_h(&sw[buf][tg][1][woff[j]]);
              const f16x8 wq = scale_2m11(wh);
#pragma unroll
              for (int v = 0; v < VT; ++v) {
#if defined(SVR_CONV_EXP) && SVR_CONV_EXP == 2   // measurement build: no matrix instructions (the fragments are still read)
                acc[v][j][0] += (float)xl[v][0] * (float)wq[0] + (float)xh[v][1] * (float)wl[1] + (float)wh[2];
#else
                acc[v][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl[v], wq, acc[v][j], 0, 0, 0);
                acc[v][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh[v], wl, acc[v][j], 0, 0, 0);
                acc[v][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh[v], wh, acc[v][j], 0, 0, 0);
#endif
              }
            }
            continue;
          }
          bf16x8 ah[VT], am[VT];
#pragma unroll
          for (int v = 0; v < VT; ++v) {
            ah[v] = read_frag(&sh[0][aoff[v]]);
            am[v] = read_frag(&sh[1][aoff[v]]);
          }
#pragma unroll
          for (int j = 0; j < TNB; ++j) {
            const bf16x8 bh = read_frag(&sw[buf][tg][0][woff[j]]);
            const bf16x8 bm = read_frag(&sw[buf][tg][1][woff[j]]);
#pragma unroll
            for (int v = 0; v < VT; ++v) {
              if constexpr (NP == 3) {
                const bf16x8 al = read_frag(&sh[NP - 1][aoff[v]]);
                const bf16x8 bl = read_frag(&sw[buf][tg][NP - 1][woff[j]]);
                acc[v][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[v][j], 0, 0, 0);
                acc[v][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[v], bl, acc[v][j], 0, 0, 0);
                acc[v][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[v], bm, acc[v][j], 0, 0, 0);
              }
              acc[v][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[v], bh, acc[v][j], 0, 0, 0);
              acc[v][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[v], bm, acc[v][j], 0, 0, 0);
              acc[v][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[v], bh, acc[v][j], 0, 0, 0);
            }
          }
        }
      }
      if (st + 1 < STEPS) wstore(nxt, buf ^ 1);
      __syncthreads();
    };
    for (int st = 0; st < STEPS; st += 2) {
      step(st, wregB, wregA);
      if (st + 1 < STEPS) step(st + 1, wregA, wregB);
    }
#if defined(SVR_CONV_EXP) && SVR_CONV_EXP == 5   // ... the loads land behind the chunk's MFMAs
#pragma unroll
    for (int i = 0; i < HIT; ++i) asm volatile("" ::"v"(hreg[i].x), "v"(hreg[i].y), "v"(hreg[i].z), "v"(hreg[i].w));
#endif
  }
  // BatchNorm statistics of what this workgroup stores (spart: per-workgroup partial sums [brick][2][Co] in f64): the
  // stage's last convolution delivers them and the separate statistics pass over its output disappears
  float ssum[TNB], ssq[TNB];
#pragma unroll
  for (int j = 0; j < TNB; ++j) ssum[j] = ssq[j] = 0.f;
#pragma unroll
  for (int v = 0; v < VT; ++v)
#pragma unroll
  for (int j = 0; j < TNB; ++j) {
    const int n = n0 + j * 32 + l31;
    const int nc = min(n, s.Co - 1);
    const float bv = (mode == SVR_EPI_BIAS || mode == SVR_EPI_BIAS_RELU) ? bias[nc] : 0.f;
    const float inv = F16 ? w_scale(amax[0], true) * (amax_in ? w_scale(amax_in[0], true) : 1.f) : 1.f;
    const int gz = z0 + wave + BRZ * v;
    // the ReLU mask of the whole tile is fetched up front from clamped coordinates: loads inside the bounds
    // branch would be waited for one at a time
    float mk[16];
    if (mode == SVR_EPI_MASK) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int cy = min(y0 + tile_vy(i), s.H - 1), cx = min(x0 + tile_vx(i), s.W - 1), cz = min(gz, s.D - 1);
        mk[r] = mask[((((int64_t)b * s.D + cz) * s.H + cy) * s.W + cx) * s.Co + nc];
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = (r & 3) + 8 * (r >> 2) + 4 * lh;
      const int gy = y0 + tile_vy(i), gx = x0 + tile_vx(i);
      if (n < s.Co && gz < s.D && gy < s.H && gx < s.W) {
        const int64_t o = ((((int64_t)b * s.D + gz) * s.H + gy) * s.W + gx) * s.Co + n;
        float val = (F16 ? acc[v][j][r] * inv : acc[v][j][r]) + bv;
        if (mode == SVR_EPI_BIAS_RELU) val = fmaxf(val, 0.f);
        if (mode == SVR_EPI_MASK) val = mk[r] > 0.f ? val : 0.f;
#ifndef SVR_CONV_NO_NT
        // streaming store: the layer's output (268 MB at 64^3 x 32) should not push the halo voxels its neighbours re-read out of
        // L2 (conv forward 1.39 -> 1.37, backward-data 1.54 -> 1.52 ms per step)
        __builtin_nontemporal_store(val, out + o);
#else
        out[o] = val;
#endif
        ssum[j] += val;
        ssq[j] = fmaf(val, val, ssq[j]);
        vmax = fmaxf(vmax, fabsf(val));
      }
    }
  }
  if (amax_out) {  // (uniform)
svr_amax_publish(amax_out, vmax);
  }
  if (spart) {  // (uniform) lanes l31 / l31 + 32 of the four waves hold the same channel: fixed-order f64 sum of the 8 partials
    float *red = reinterpret_cast<float *>(&sw[0][0][0][0]);   // the weight buffers are free now: [thread][TNB][2] floats
    static_assert(sizeof(sw) >= 256 * TNB * 2 * sizeof(float), "weight buffer too small for the statistics reduction");
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TNB; ++j) {
      red[(t * TNB + j) * 2] = ssum[j];
      red[(t * TNB + j) * 2 + 1] = ssq[j];
    }
    __syncthreads();
    if (t < 2 * NC) {
      const int which = t / NC, col = t % NC, j = col / 32, c31 = col % 32;
      double acc = 0.0;
#pragma unroll
      for (int w = 0; w < 4; ++w)
#pragma unroll
        for (int h = 0; h < 2; ++h) acc += (double)red[(((w * 64 + h * 32 + c31) * TNB) + j) * 2 + which];
      if (n0 + col < s.Co) spart[((int64_t)blockIdx.x * 2 + which) * s.Co + n0 + col] = acc;
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// PERSISTENT form of the brick kernel for 32 output columns (TNB = 1), round 3.  Measurement builds of the kernel above
// (tools/exp/conv_variants.sh, SVR_CONV_EXP = 2 / 3 / 5) showed that at 64^3 the matrix instructions are completely hidden
// (no MFMAs: same time), that the halo tile's global loads cost 19-23 % and that half of that is the WAIT for them: a
// workgroup that stages a chunk computes nothing, and with two workgroups per CU the other one then has each SIMD to itself,
// one wave that stalls on every LDS fragment read.  Here a workgroup walks bricks (grid = the resident workgroups) and the
// NEXT work item's halo tile -- the next 16 / 32-channel chunk of the brick or the first chunk of the next brick -- is
// fetched into registers before the current item's MFMA steps and split into LDS behind them; the weight pipeline (two
// register sets, two steps ahead) runs straight through the item boundaries: an item is TEN barrier steps (taps 3 3 3 3 3 3
// 3 2 2 2), an even number, so the two register sets keep their roles from one item to the next.
// ---------------------------------------------------------------------------------------------------------------------
template <int CK, bool F16, int VT>
__global__ __launch_bounds__(256, 2) void conv3d_brick_p_kernel(const float *__restrict__ in, const uint16_t *__restrict__ P0,
                                                             int64_t plane_stride, const float *__restrict__ bias,
                                                             float *__restrict__ out, const float *__restrict__ mask,
                                                             ConvShape s, int nbz, int nby, int nbx, int nbricks, int mode,
                                                             const uint32_t *__restrict__ amax, double *__restrict__ spart,
                                                             const uint32_t *__restrict__ amax_in, uint32_t *__restrict__ amax_out) {
  float sin = 1.f;   // (see conv3d_brick_x3_kernel)
  if constexpr (F16) sin = amax_in ? w_scale(amax_in[0], false) : 1.f;
  float vmax = 0.f;
  constexpr int NP = 2, NC = 32, TG = 3, STEPS = 10;
  constexpr int BZ = BRZ * VT, HV = (BZ + 2) * HLY * HLX;
  constexpr int SL = CK / 8, XW = SL * 4, HR = (BZ + 2) * HLY * HP, KS = CK / 16;
  constexpr int PIECES = TG * NP * NC * (CK / 8), WPT = (PIECES + 255) / 256;
  constexpr int HIT = (HV * (CK / 4) + 255) / 256;
  static_assert(CK == 16, "the piece -> (tap, plane, row, part) split below is written out for 16-channel chunks");
  __shared__ __attribute__((aligned(16))) uint32_t sh[NP][HR * XW];
  __shared__ __attribute__((aligned(16))) uint32_t sw[2][TG][NP][NC * XW];
  static_assert(sizeof(sh) >= 256 * 2 * sizeof(float), "halo buffer too small for the statistics reduction");
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int n0 = blockIdx.y * NC;
  const int hrow = ((wave + 1) * HLY + tile_vy(l31) + 1) * HP + tile_vx(l31) + 1;
  const int nch = s.Ci / CK;
  const int nmy = (nbricks - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;   // bricks of this workgroup (>= 1: grid <= bricks)
  const int nitems = nmy * nch;

  // ADDRESS ARITHMETIC (round 3, SQ counters: this kernel issued 1 520 vector-ALU instructions per item and wave next to 162
  // MFMAs -- 44 % + 38 % of the SIMDs' time, and the two add up; a fifth of them were 32 / 64-bit integer multiplies of the
  // flat index arithmetic).  Everything below is 32-bit: wave-uniform bases (SGPRs) + per-lane offsets inside ONE sample
  // (host: D H W <= 2^24 voxels and D H W max(Ci, Co) < 2^30 elements, so v_mad_u32_u24 is exact and byte offsets fit),
  // and what only depends on the thread is computed once, in front of the item loop.
  const uint32_t HW = (uint32_t)(s.H * s.W);

  f32x16 acc[VT];
#pragma unroll
  for (int v = 0; v < VT; ++v)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[v][r] = 0.f;

  struct Item { int b, z0, y0, x0, k0, brick; };
  auto decode = [&](int it) {   // (wave-uniform: scalar unit)
    Item I;
    const int bi = it / nch;
    I.k0 = (it - bi * nch) * CK;
    I.brick = (int)blockIdx.x + bi * (int)gridDim.x;
    int q = I.brick;
    I.x0 = (q % nbx) * BRX; q /= nbx;
    I.y0 = (q % nby) * BRY; q /= nby;
    I.z0 = (q % nbz) * BZ;
    I.b = q / nbz;
    return I;
  };

  // ---- weights: piece 0 of a thread = (tap tg0 of the step, plane, row, 8-channel part), piece 1 (threads < 128) = the
  // step's third tap; step st of an item covers taps tap0(st) .. + 2 (the last three steps: + 1, no piece 1)
  const int p0part = t & 1, p0row = (t >> 1) & 31, p0pl = (t >> 6) & 1, p0tg = t >> 7;
  const bool rowok = n0 + p0row < s.Co;
  const uint32_t wtap = (uint32_t)(s.Co * s.Ci) * 2u;                             // bytes per tap of one plane
  const uint32_t wrow = (uint32_t)((rowok ? n0 + p0row : 0) * s.Ci + p0part * 8) * 2u;
  const uint32_t wof0 = (uint32_t)p0pl * (uint32_t)plane_stride * 2u + (uint32_t)p0tg * wtap + wrow;   // + (tap0 Co Ci + k0) * 2
  const uint32_t wof1 = (uint32_t)((t >> 6) & 1) * (uint32_t)plane_stride * 2u + 2u * wtap + wrow;      // piece 1: tg = 2, plane = (t >> 6) & 1
  const int wst0 = (int)(&sw[0][p0tg][p0pl][slot_dw<SL>(p0row, p0part)] - &sw[0][0][0][0]);
  const int wst1 = (int)(&sw[0][2][(t >> 6) & 1][slot_dw<SL>(p0row, p0part)] - &sw[0][0][0][0]);
  constexpr int WBUF = TG * NP * NC * XW;   // dwords per weight buffer
  static_assert(PIECES == 384 && WPT == 2, "pieces 0 (256 threads) and 1 (128 threads)");
  // (scalars, not arrays: a two-element array with a conditionally written element was demoted to LDS by the compiler)
  struct WReg { uint4 p0, p1; };
  WReg wregA, wregB;
  wregA.p1 = wregB.p1 = make_uint4(0u, 0u, 0u, 0u);
  const char *Pb = reinterpret_cast<const char *>(P0);
  auto wload = [&](WReg &wreg, int k0, int st) {
    const int tap0 = st < 7 ? 3 * st : 21 + 2 * (st - 7);
    const uint32_t sb = (uint32_t)tap0 * wtap + (uint32_t)k0 * 2u;   // (uniform)
    wreg.p0 = *reinterpret_cast<const uint4 *>(Pb + (wof0 + sb));
    if (st < 7) wreg.p1 = *reinterpret_cast<const uint4 *>(Pb + (wof1 + sb));   // (compile time: the two-tap steps have no third tap)
  };
  auto wstore = [&](const WReg &wreg, int buf, int st_of_data) {
    uint32_t *base = &sw[0][0][0][0] + buf * WBUF;
    *reinterpret_cast<uint4 *>(base + wst0) = wreg.p0;
    if (st_of_data < 7 && t < 128) *reinterpret_cast<uint4 *>(base + wst1) = wreg.p1;
  };

  // ---- halo tile of an item: global -> registers (unconditional, clamped coordinates), later registers -> split -> LDS.
  // Per piece the thread's halo voxel (hz, hy, hx) and its LDS offset are constants: packed once (8 + 8 + 8 + 8 bits would
  // not hold the LDS offset: two packed registers per piece would cost 20 VGPRs; the LDS offset is recomputed, 6 VALU)
  constexpr int C4 = CK / 4;
  const int c4 = (t % C4) * 4;   // (256 % C4 == 0: the same channel quad in every piece)
  float4 hreg[HIT];
  uint32_t hok = 0;
  auto halo_vox = [&](int i, int &hz, int &hy, int &hx) {   // compile-time i: constants folded per piece, t-dependent part small
    const int idx = min(t + 256 * i, HV * C4 - 1);
    const int hv = idx / C4;
    hx = hv % HLX;
    const int q = hv / HLX;
    hy = q % HLY;
    hz = q / HLY;
  };
  // (hz, hy, hx) of every piece, packed: 5 + 3 + 4 bits
  int hpk[HIT];
#pragma unroll
  for (int i = 0; i < HIT; ++i) {
    int hz, hy, hx;
    halo_vox(i, hz, hy, hx);
    hpk[i] = hz | (hy << 5) | (hx << 8);
  }
  auto halo_load = [&](const Item &I, int i0, int i1) {
    const char *inb = reinterpret_cast<const char *>(in + (int64_t)I.b * s.D * s.H * s.W * s.Ci);   // (uniform)
    if (i0 == 0) hok = 0;
#pragma unroll
    for (int i = i0; i < i1; ++i) {
      const int gz = I.z0 - 1 + (hpk[i] & 31), gy = I.y0 - 1 + ((hpk[i] >> 5) & 7), gx = I.x0 - 1 + (hpk[i] >> 8);
      if ((unsigned)gz < (unsigned)s.D && (unsigned)gy < (unsigned)s.H && (unsigned)gx < (unsigned)s.W) hok |= 1u << i;
      const uint32_t cz = (uint32_t)min(max(gz, 0), s.D - 1), cy = (uint32_t)min(max(gy, 0), s.H - 1), cx = (uint32_t)min(max(gx, 0), s.W - 1);
      const uint32_t vox = __umul24(__umul24(cz, (uint32_t)s.H) + cy, (uint32_t)s.W) + cx;
      const uint32_t off = (__umul24(vox, (uint32_t)s.Ci) + (uint32_t)(I.k0 + c4)) * 4u;
      hreg[i] = *reinterpret_cast<const float4 *>(inb + off);
    }
  };
  auto halo_store = [&]() {
#pragma unroll
    for (int i = 0; i < HIT; ++i) {
      if (t + 256 * i >= HV * C4) break;
      const int row = ((hpk[i] & 31) * HLY + ((hpk[i] >> 5) & 7)) * HP + (hpk[i] >> 8);
      const int hd = slot_dw<SL>(row, c4 / 8) + (c4 % 8) / 2;
      const bool ok = (hok >> i) & 1u;
      const float4 v = make_float4(ok ? hreg[i].x : 0.f, ok ? hreg[i].y : 0.f, ok ? hreg[i].z : 0.f, ok ? hreg[i].w : 0.f);
      uint32_t h0, m0, h1, m1;
      if constexpr (F16) {
        split_x(v.x * sin, v.y * sin, h0, m0);
        split_x(v.z * sin, v.w * sin, h1, m1);
      } else {
        split2(v.x, v.y, h0, m0);
        split2(v.z, v.w, h1, m1);
      }
      *reinterpret_cast<uint2 *>(&sh[0][hd]) = make_uint2(h0, h1);
      *reinterpret_cast<uint2 *>(&sh[1][hd]) = make_uint2(m0, m1);
    }
  };

  // ---- epilogue of a finished brick (the kernel above's, for TNB = 1); the accumulators start the next brick at zero.
  // Element offset of result r = (uniform voxel base + lane part) * Co + n: the lane part (vy W + vx) of the 16 results is
  // one of two compile-time tables (lh), the voxel base is scalar
  auto epilogue = [&](const Item &I) {
    float ssum = 0.f, ssq = 0.f;
    const int n = n0 + l31, nc = min(n, s.Co - 1);
    const float bv = (mode == SVR_EPI_BIAS || mode == SVR_EPI_BIAS_RELU) ? bias[nc] : 0.f;
    const float inv = F16 ? w_scale(amax[0], true) * (amax_in ? w_scale(amax_in[0], true) : 1.f) : 1.f;
    const bool interior = I.z0 + BZ <= s.D && I.y0 + BRY <= s.H && I.x0 + BRX <= s.W && n0 + NC <= s.Co;   // (uniform)
    char *outb = reinterpret_cast<char *>(out + (int64_t)I.b * s.D * s.H * s.W * s.Co);
    const char *maskb = reinterpret_cast<const char *>(mask + (mode == SVR_EPI_MASK ? (int64_t)I.b * s.D * s.H * s.W * s.Co : 0));
#pragma unroll
    for (int v = 0; v < VT; ++v) {
      const int gz = I.z0 + wave + BRZ * v;
      const uint32_t vbase = __umul24((uint32_t)min(gz, s.D - 1), HW) + __umul24((uint32_t)I.y0, (uint32_t)s.W) + (uint32_t)I.x0;   // (uniform)
      uint32_t off[16];
      bool okr[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i0 = (r & 3) + 8 * (r >> 2);                 // MFMA row of lanes 0 .. 31; lanes 32 .. 63: + 4
        const int vy = lh ? tile_vy(i0 + 4) : tile_vy(i0), vx = lh ? tile_vx(i0 + 4) : tile_vx(i0);   // two constants, one select each
        const int gy = I.y0 + vy, gx = I.x0 + vx;
        okr[r] = interior || (n < s.Co && gz < s.D && gy < s.H && gx < s.W);
        // (a voxel outside the volume gets the offset of a clamped one: never stored, its mask value never used)
        const uint32_t lv = __umul24((uint32_t)min(vy, s.H - 1 - I.y0), (uint32_t)s.W) + (uint32_t)min(vx, s.W - 1 - I.x0);
        off[r] = (__umul24(vbase + lv, (uint32_t)s.Co) + (uint32_t)nc) * 4u;
      }
      float mk[16];
      if (mode == SVR_EPI_MASK) {
#pragma unroll
        for (int r = 0; r < 16; ++r) mk[r] = *reinterpret_cast<const float *>(maskb + off[r]);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (okr[r]) {
          float val = (F16 ? acc[v][r] * inv : acc[v][r]) + bv;
          if (mode == SVR_EPI_BIAS_RELU) val = fmaxf(val, 0.f);
          if (mode == SVR_EPI_MASK) val = mk[r] > 0.f ? val : 0.f;
#ifndef SVR_CONV_NO_NT
          __builtin_nontemporal_store(val, reinterpret_cast<float *>(outb + off[r]));
#else
          *reinterpret_cast<float *>(outb + off[r]) = val;
#endif
          ssum += val;
          ssq = fmaf(val, val, ssq);
          vmax = fmaxf(vmax, fabsf(val));
        }
        acc[v][r] = 0.f;
      }
    }
    if (spart) {  // (uniform) per-brick partial sums, row = brick id: the same rows as the one-brick-per-workgroup kernel writes
      float *red = reinterpret_cast<float *>(&sh[0][0]);   // the halo tile is free between an item's last step and the next store
      red[t * 2] = ssum;
      red[t * 2 + 1] = ssq;
      __syncthreads();
      if (t < 2 * NC) {
        const int which = t / NC, c31 = t % NC;
        double a2 = 0.0;
#pragma unroll
        for (int w = 0; w < 4; ++w)
#pragma unroll
          for (int h = 0; h < 2; ++h) a2 += (double)red[(w * 64 + h * 32 + c31) * 2 + which];
        if (n0 + c31 < s.Co) spart[((int64_t)I.brick * 2 + which) * s.Co + n0 + c31] = a2;
      }
      __syncthreads();
    }
  };

  // W fragment offset of this lane inside a tap's plane (rows = output columns)
  int woff[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) woff[ks] = slot_dw<SL>(l31, ks * 2 + lh);

  Item cur = decode(0);
  halo_load(cur, 0, HIT);
  wload(wregA, cur.k0, 0);
  wload(wregB, cur.k0, 1);
  halo_store();
  wstore(wregA, 0, 0);
  __syncthreads();
  for (int it = 0; it < nitems; ++it) {
    const Item nxt = decode(it + 1 < nitems ? it + 1 : it);   // (clamped: the last item fetches itself again, never stored)
    // the 27 swizzled fragment addresses are loop invariant; hoisted out of the item loop they cost 27 VGPRs and the kernel
    // its second workgroup per CU.  Opaque per item, each is computed where it is used (4 VALU) and dies there; the second
    // z-slice of the wave sits BRZ HLY HP = 288 rows further -- a multiple of 16, so the swizzle bit is the same and its
    // address is an immediate offset
    int hrow_i = hrow;
    asm volatile("" : "+v"(hrow_i));
    // step st: MFMAs on LDS buffer st & 1; `nx` holds step st + 1 (in flight since st - 1), `fr` is refilled with st + 2
    auto step = [&](int st, WReg &nx, WReg &fr) {
      const int buf = st & 1, tap0 = st < 7 ? 3 * st : 21 + 2 * (st - 7), nt = st < 7 ? 3 : 2;
      if (st + 2 < STEPS) wload(fr, cur.k0, st + 2);
      else wload(fr, nxt.k0, st + 2 - STEPS);
      halo_load(nxt, (st * HIT) / STEPS, ((st + 1) * HIT) / STEPS);   // this step's share of the next item's halo tile
#pragma unroll
      for (int tg = 0; tg < TG; ++tg) {
        if (tg >= nt) break;
        const int tap = tap0 + tg;
        const int arow = hrow_i + ((tap / 9 - 1) * HLY + ((tap / 3) % 3 - 1)) * HP + (tap % 3 - 1);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const int sub = ks * 2 + lh;
          const int aoff = slot_dw<SL>(arow, sub);
          static_assert((BRZ * HLY * HP) % 16 == 0, "second z-slice: same swizzle");
          constexpr int VSTEP = BRZ * HLY * HP * XW;   // dwords between the wave's z-slices
          if constexpr (F16) {
            f16x8 xh[VT], xl[VT];
#pragma unroll
            for (int v = 0; v < VT; ++v) {
              xh[v] = read_frag_h(&sh[0][aoff + v * VSTEP]);
              xl[v] = read_frag_h(&sh[1][aoff + v * VSTEP]);
            }
            const f16x8 wh = read_frag_h(&sw[buf][tg][0][woff[ks]]);
            const f16x8 wl = read_frag_h(&sw[buf][tg][1][woff[ks]]);
            const f16x8 wq = scale_2m11(wh);
#pragma unroll
            for (int v = 0; v < VT; ++v) {
              acc[v] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl[v], wq, acc[v], 0, 0, 0);
              acc[v] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh[v], wl, acc[v], 0, 0, 0);
              acc[v] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh[v], wh, acc[v], 0, 0, 0);
            }
          } else {
            bf16x8 ah[VT], am[VT];
#pragma unroll
            for (int v = 0; v < VT; ++v) {
              ah[v] = read_frag(&sh[0][aoff + v * VSTEP]);
              am[v] = read_frag(&sh[1][aoff + v * VSTEP]);
            }
            const bf16x8 bh = read_frag(&sw[buf][tg][0][woff[ks]]);
            const bf16x8 bm = read_frag(&sw[buf][tg][1][woff[ks]]);
#pragma unroll
            for (int v = 0; v < VT; ++v) {
              acc[v] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[v], bh, acc[v], 0, 0, 0);
              acc[v] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[v], bm, acc[v], 0, 0, 0);
              acc[v] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[v], bh, acc[v], 0, 0, 0);
            }
          }
        }
      }
      wstore(nx, buf ^ 1, (st + 1) % STEPS);
      __syncthreads();
    };
#pragma unroll
    for (int st = 0; st < STEPS; st += 2) {
      step(st, wregB, wregA);
      step(st + 1, wregA, wregB);
    }
    if (cur.k0 + CK >= s.Ci) epilogue(cur);   // (uniform) the brick's last chunk
    halo_store();                              // the next item's tile, fetched during this item's steps
    __syncthreads();
    cur = nxt;
  }
  if (amax_out) {  // (uniform) |max| of everything this workgroup stored
svr_amax_publish(amax_out, vmax);
  }
}

// SVR_CONV_PERSISTENT=0: the one-brick-per-workgroup kernel everywhere (A/B switch, read once)
bool persistent_bricks(const ConvShape &sh) {
  static const bool on = !(getenv("SVR_CONV_PERSISTENT") && getenv("SVR_CONV_PERSISTENT")[0] == '0');
  // the kernel's 24-bit voxel index multiplies and 32-bit byte offsets inside one sample
  const int64_t vox = (int64_t)sh.D * sh.H * sh.W, cmax = sh.Ci > sh.Co ? sh.Ci : sh.Co;
  return on && vox <= (1LL << 24) && vox * cmax < (1LL << 30);
}
// resident workgroups of an instantiation (CUs x occupancy; cached per kernel: immutable after the first call)
template <int CK, bool F16, int VT>
int brick_p_resident() {
  static int resident = 0;
  if (resident == 0) {
    int dev = 0, cus = 256, per_cu = 2;
    if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, conv3d_brick_p_kernel<CK, F16, VT>, 256, 0) != hipSuccess || per_cu < 1) per_cu = 2;
    (void)hipGetLastError();
    resident = cus * per_cu;
  }
  return resident;
}
template <int CK, bool F16, int VT>
void launch_brick_p(const float *in, const uint16_t *P0, int64_t plane_stride, const float *bias, float *out, const float *mask,
                    ConvShape sh, int nbz, int nby, int nbx, int ycols, int mode, const uint32_t *amax, double *spart, hipStream_t s,
                    const uint32_t *amax_in = nullptr, uint32_t *amax_out = nullptr) {
  const int64_t bricks = (int64_t)sh.B * nbz * nby * nbx;
  int gx = brick_p_resident<CK, F16, VT>() / ycols;
  if (gx < 1) gx = 1;
  if (gx > bricks) gx = (int)bricks;
  hipLaunchKernelGGL((conv3d_brick_p_kernel<CK, F16, VT>), dim3((unsigned)gx, (unsigned)ycols), dim3(256), 0, s, in, P0, plane_stride, bias, out,
                     mask, sh, nbz, nby, nbx, (int)bricks, mode, amax, spart, amax_in, amax_out);
}

}  // namespace

extern "C" int64_t svr_conv3d_bwd_data_bf16x3_workspace(int32_t Ci, int32_t Co) { return 2LL * 27 * Ci * Co * (int64_t)sizeof(uint16_t) + 256; }

// dIn(B,D,H,W,Ci) = epi( conv^T(dOut(B,D,H,W,Co), W(Co,Ci,3,3,3)) ), epilogue NONE or MASK (mask shaped like dIn).
extern "C" int svr_conv3d_k3_bwd_data_bf16x3(const float *dout, const float *W, float *din, int32_t B, int32_t D, int32_t H,
                                             int32_t Wd, int32_t Ci, int32_t Co, int epilogue, const float *mask,
                                             void *workspace, void *stream) {
  // dout == NULL: PREPARE only (W -> split planes in the workspace);  W == NULL: RUN on a workspace prepared earlier
  SVR_CHECK((dout || W) && (!dout || din) && workspace, SVR_E_BADARG, "conv3d_bwd_data_bf16x3: null pointer");
  SVR_CHECK(Co % 16 == 0 && Ci % 2 == 0 && Ci >= 2, SVR_E_UNSUPPORTED, "conv3d_bwd_data_bf16x3: need Co %% 16 == 0, Ci even (Ci=%d Co=%d)", Ci, Co);
  hipStream_t s = (hipStream_t)stream;
  uint16_t *hi = (uint16_t *)(((uintptr_t)workspace + 15) & ~(uintptr_t)15);
  uint16_t *mid = hi + (size_t)27 * Ci * Co;
  // the transposed conv reads dOut (Co channels = its K) and writes Ci channels: planes [27][Ci][Co]
  if (W) hipLaunchKernelGGL(pack_bwd_planes_kernel, dim3(cdiv(27 * Ci * (Co / 2), 256)), dim3(256), 0, s, W, hi, mid, Ci, Co);
  if (!dout) return launch_status("conv3d_bwd_data_bf16x3 (prepare)");
  SVR_CHECK(B > 0 && D > 0 && H > 0 && Wd > 0, SVR_E_BADSHAPE, "conv3d_bwd_data_bf16x3: empty volume");
  SVR_CHECK(epilogue == SVR_EPI_NONE || (epilogue == SVR_EPI_MASK && mask), SVR_E_BADARG, "conv3d_bwd_data_bf16x3: epilogue %d", epilogue);
  ConvShape sh{B, D, H, Wd, /*K=*/Co, /*NOUT=*/Ci};
  const int nbz = (int)cdiv(D, BRZ), nby = (int)cdiv(H, BRY), nbx = (int)cdiv(Wd, BRX);
  const unsigned bricks = (unsigned)((int64_t)B * nbz * nby * nbx);
#define LAUNCH_X3(CKV, TNV)                                                                                               \
  hipLaunchKernelGGL((conv3d_brick_x3_kernel<CKV, TNV, 2>), dim3(bricks, (unsigned)cdiv(Ci, TNV * 32)), dim3(256), 0, s, dout, \
                     hi, (int64_t)27 * Ci * Co, (const float *)nullptr, din, mask, sh, nbz, nby, nbx, epilogue)
  // output-column tiles per workgroup: as wide as the layer allows (fragment reuse), but narrower on the small volumes
  // so that the grid still has >= 512 workgroups (the 8^3 / 16^3 layers ran 32 / 256 workgroups of 108 barrier steps)
  int tn = Ci <= 32 ? 1 : (Ci <= 64 ? 2 : 4);
  while (tn > 1 && (int64_t)bricks * cdiv(Ci, tn * 32) < 512) tn /= 2;
  const int nbz2 = (int)cdiv(D, 2 * BRZ);
  if (tn == 1 && Co % 16 == 0 && (int64_t)B * nbz2 * nby * nbx * cdiv(Ci, 32) >= 512) {
    // 32 output columns: two z-slices per wave (8x4x8 bricks, 16-channel chunks)
    if (persistent_bricks(sh))
      launch_brick_p<16, false, 2>(dout, hi, (int64_t)27 * Ci * Co, nullptr, din, mask, sh, nbz2, nby, nbx, (int)cdiv(Ci, 32), epilogue, nullptr, nullptr, s);
    else
    hipLaunchKernelGGL((conv3d_brick_x3_kernel<16, 1, 2, false, 2>), dim3((unsigned)((int64_t)B * nbz2 * nby * nbx), (unsigned)cdiv(Ci, 32)),
                       dim3(256), 0, s, dout, hi, (int64_t)27 * Ci * Co, (const float *)nullptr, din, mask, sh, nbz2, nby, nbx, epilogue);
  } else if (Co % 32 == 0) {
    if (tn == 1) LAUNCH_X3(32, 1); else if (tn == 2) LAUNCH_X3(32, 2); else LAUNCH_X3(32, 4);
  } else {
    if (tn == 1) LAUNCH_X3(16, 1); else if (tn == 2) LAUNCH_X3(16, 2); else LAUNCH_X3(16, 4);
  }
#undef LAUNCH_X3
  return launch_status("conv3d_bwd_data_bf16x3");
}

// ---- backward-data on the scaled f16 split ("f16x3s": f32 level at the bf16x3 cost; see svr_linear_bwd_data_f16x3) -------------
extern "C" int64_t svr_conv3d_bwd_data_f16x3_workspace(int32_t Ci, int32_t Co) { return 2LL * 27 * Ci * Co * (int64_t)sizeof(uint16_t) + 512; }

extern "C" int svr_conv3d_k3_bwd_data_f16x3(const float *dout, const float *W, float *din, int32_t B, int32_t D, int32_t H,
                                            int32_t Wd, int32_t Ci, int32_t Co, int epilogue, const float *mask,
                                            const uint32_t *amax_dout, uint32_t *amax_din, void *workspace, void *stream) {
  // dout == NULL: PREPARE only (W -> scale + split planes in the workspace);  W == NULL: RUN on a workspace prepared earlier
  SVR_CHECK((dout || W) && (!dout || din) && workspace, SVR_E_BADARG, "conv3d_bwd_data_f16x3: null pointer");
  SVR_CHECK(Co % 16 == 0 && Ci % 2 == 0 && Ci >= 2, SVR_E_UNSUPPORTED, "conv3d_bwd_data_f16x3: need Co %% 16 == 0, Ci even (Ci=%d Co=%d)", Ci, Co);
  hipStream_t s = (hipStream_t)stream;
  uint32_t *amax = (uint32_t *)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  uint16_t *hi = (uint16_t *)(amax + 64);
  const int64_t ps = (int64_t)27 * Ci * Co;
  if (W) {
    (void)hipMemsetAsync(amax, 0, sizeof(uint32_t), s);
    hipLaunchKernelGGL(w_amax_kernel, dim3((unsigned)cdiv(ps, 1024)), dim3(256), 0, s, W, ps, (int64_t)1, ps, amax);
    hipLaunchKernelGGL(pack_bwd_planes_f16_kernel, dim3(cdiv(27 * Ci * (Co / 2), 256)), dim3(256), 0, s, W, amax, hi, hi + ps, Ci, Co);
  }
  if (!dout) return launch_status("conv3d_bwd_data_f16x3 (prepare)");
  SVR_CHECK(B > 0 && D > 0 && H > 0 && Wd > 0, SVR_E_BADSHAPE, "conv3d_bwd_data_f16x3: empty volume");
  SVR_CHECK(epilogue == SVR_EPI_NONE || (epilogue == SVR_EPI_MASK && mask), SVR_E_BADARG, "conv3d_bwd_data_f16x3: epilogue %d", epilogue);
  ConvShape sh{B, D, H, Wd, /*K=*/Co, /*NOUT=*/Ci};
  const int nbz = (int)cdiv(D, BRZ), nby = (int)cdiv(H, BRY), nbx = (int)cdiv(Wd, BRX);
  const unsigned bricks = (unsigned)((int64_t)B * nbz * nby * nbx);
#define LAUNCH_B3(CKV, TNV)                                                                                                     \
  hipLaunchKernelGGL((conv3d_brick_x3_kernel<CKV, TNV, 2, true>), dim3(bricks, (unsigned)cdiv(Ci, TNV * 32)), dim3(256), 0, s, dout, \
                     hi, ps, (const float *)nullptr, din, mask, sh, nbz, nby, nbx, epilogue, amax, (double *)nullptr, amax_dout, amax_din)
  int tn = Ci <= 32 ? 1 : (Ci <= 64 ? 2 : 4);                        // (the tile choice of svr_conv3d_k3_bwd_data_bf16x3)
  while (tn > 1 && (int64_t)bricks * cdiv(Ci, tn * 32) < 512) tn /= 2;
  const int nbz2 = (int)cdiv(D, 2 * BRZ);
  if (tn == 1 && (int64_t)B * nbz2 * nby * nbx * cdiv(Ci, 32) >= 512) {
    if (persistent_bricks(sh))
      launch_brick_p<16, true, 2>(dout, hi, ps, nullptr, din, mask, sh, nbz2, nby, nbx, (int)cdiv(Ci, 32), epilogue, amax, nullptr, s, amax_dout, amax_din);
    else
      hipLaunchKernelGGL((conv3d_brick_x3_kernel<16, 1, 2, true, 2>), dim3((unsigned)((int64_t)B * nbz2 * nby * nbx), (unsigned)cdiv(Ci, 32)),
                         dim3(256), 0, s, dout, hi, ps, (const float *)nullptr, din, mask, sh, nbz2, nby, nbx, epilogue, amax, (double *)nullptr,
                         amax_dout, amax_din);
  } else if (Co % 32 == 0) {
    if (tn == 1) LAUNCH_B3(32, 1); else if (tn == 2) LAUNCH_B3(32, 2); else LAUNCH_B3(32, 4);
  } else {
    if (tn == 1) LAUNCH_B3(16, 1); else if (tn == 2) LAUNCH_B3(16, 2); else LAUNCH_B3(16, 4);
  }
#undef LAUNCH_B3
  return launch_status("conv3d_bwd_data_f16x3");
}

extern "C" int64_t svr_conv3d_fwd_bf16x6_workspace(int32_t Ci, int32_t Co) { return 3LL * 27 * Ci * Co * (int64_t)sizeof(uint16_t) + 256; }

// out(B,D,H,W,Co) = epi( conv(in(B,D,H,W,Ci), W(Co,Ci,3,3,3)) ) at f32 accuracy; epilogue NONE / BIAS / BIAS_RELU.
extern "C" int svr_conv3d_k3_fwd_bf16x6(const float *in, const float *W, const float *bias, float *out, int32_t B, int32_t D,
                                        int32_t H, int32_t Wd, int32_t Ci, int32_t Co, int epilogue, void *workspace,
                                        void *stream) {
  SVR_CHECK(in && W && out && workspace, SVR_E_BADARG, "conv3d_fwd_bf16x6: null pointer");
  SVR_CHECK(B > 0 && D > 0 && H > 0 && Wd > 0, SVR_E_BADSHAPE, "conv3d_fwd_bf16x6: empty volume");
  SVR_CHECK(Ci % 16 == 0 && Co >= 1, SVR_E_UNSUPPORTED, "conv3d_fwd_bf16x6: need Ci %% 16 == 0 (Ci=%d Co=%d)", Ci, Co);
  SVR_CHECK(epilogue == SVR_EPI_NONE || ((epilogue == SVR_EPI_BIAS || epilogue == SVR_EPI_BIAS_RELU) && bias), SVR_E_BADARG,
            "conv3d_fwd_bf16x6: epilogue %d", epilogue);
  hipStream_t s = (hipStream_t)stream;
  uint16_t *p0 = (uint16_t *)(((uintptr_t)workspace + 15) & ~(uintptr_t)15);
  const int64_t ps = (int64_t)27 * Ci * Co;
  hipLaunchKernelGGL(pack_fwd_planes_kernel, dim3(cdiv(27 * Co * (Ci / 2), 256)), dim3(256), 0, s, W, p0, p0 + ps, p0 + 2 * ps, Ci, Co);
  ConvShape sh{B, D, H, Wd, Ci, Co};
  const int nbz = (int)cdiv(D, BRZ), nby = (int)cdiv(H, BRY), nbx = (int)cdiv(Wd, BRX);
  const unsigned bricks = (unsigned)((int64_t)B * nbz * nby * nbx);
#define LAUNCH_X6(TNV)                                                                                                    \
  hipLaunchKernelGGL((conv3d_brick_x3_kernel<16, TNV, 3>), dim3(bricks, (unsigned)cdiv(Co, TNV * 32)), dim3(256), 0, s, in, p0, \
                     ps, bias, out, (const float *)nullptr, sh, nbz, nby, nbx, epilogue)
  if (Co <= 32) LAUNCH_X6(1); else if (Co <= 64) LAUNCH_X6(2); else LAUNCH_X6(4);
#undef LAUNCH_X6
  return launch_status("conv3d_fwd_bf16x6");
}

extern "C" int64_t svr_conv3d_fwd_f16x3_workspace(int32_t Ci, int32_t Co) { return 2LL * 27 * Ci * Co * (int64_t)sizeof(uint16_t) + 512; }

// out(B,D,H,W,Co) = epi( conv(in(B,D,H,W,Ci), W(Co,Ci,3,3,3)) ) with the 3-product f16 split: f32-level accuracy for
// |in| < 65504 (see gemm_f16x3.hip); epilogue NONE / BIAS / BIAS_RELU.
namespace {
int fwd_f16x3(const float *in, const float *W, const float *bias, float *out, int32_t B, int32_t D, int32_t H, int32_t Wd, int32_t Ci,
              int32_t Co, int epilogue, void *workspace, void *stream, double *spart, int *blocks);
}

extern "C" int svr_conv3d_k3_fwd_f16x3(const float *in, const float *W, const float *bias, float *out, int32_t B, int32_t D,
                                       int32_t H, int32_t Wd, int32_t Ci, int32_t Co, int epilogue, void *workspace,
                                       void *stream) {
  return fwd_f16x3(in, W, bias, out, B, D, H, Wd, Ci, Co, epilogue, workspace, stream, nullptr, nullptr);
}

// The same convolution + the BatchNorm statistics of its output as per-workgroup partial sums: part[blocks][2][Co] float64
// (sum, sum of squares), blocks = svr_conv3d_fwd_f16x3_stats_blocks(...); svr_bn_finalize_parts turns them into the
// BatchNorm's scale / shift / running statistics -- no statistics pass over the output.
extern "C" int32_t svr_conv3d_fwd_f16x3_stats_blocks(int32_t B, int32_t D, int32_t H, int32_t Wd, int32_t Ci, int32_t Co) {
  int blocks = 0;
  (void)fwd_f16x3(nullptr, nullptr, nullptr, nullptr, B, D, H, Wd, Ci, Co, SVR_EPI_NONE, nullptr, nullptr, nullptr, &blocks);
  return blocks;
}
extern "C" int svr_conv3d_k3_fwd_f16x3_stats(const float *in, const float *W, const float *bias, float *out, double *part, int32_t B,
                                             int32_t D, int32_t H, int32_t Wd, int32_t Ci, int32_t Co, int epilogue,
                                             void *workspace, void *stream) {
  SVR_CHECK(in && part, SVR_E_BADARG, "conv3d_fwd_f16x3_stats: null pointer");
  return fwd_f16x3(in, W, bias, out, B, D, H, Wd, Ci, Co, epilogue, workspace, stream, part, nullptr);
}

namespace {
int fwd_f16x3(const float *in, const float *W, const float *bias, float *out, int32_t B, int32_t D, int32_t H, int32_t Wd, int32_t Ci,
              int32_t Co, int epilogue, void *workspace, void *stream, double *spart, int *blocks) {
  if (blocks) {   // query: the number of workgroup rows (= partial-sum rows) of the variant this shape takes
    const int nbz = (int)cdiv(D, BRZ), nby = (int)cdiv(H, BRY), nbx = (int)cdiv(Wd, BRX);
    const int64_t bricks = (int64_t)B * nbz * nby * nbx;
    int tn = Co <= 32 ? 1 : (Co <= 64 ? 2 : 4);
    while (tn > 1 && bricks * cdiv(Co, tn * 32) < 512) tn /= 2;
    const int nbz2 = (int)cdiv(D, 2 * BRZ);
    *blocks = (tn == 1 && (int64_t)B * nbz2 * nby * nbx * cdiv(Co, 32) >= 512) ? (int)((int64_t)B * nbz2 * nby * nbx) : (int)bricks;
    return SVR_OK;
  }
  // in == NULL: PREPARE only (W -> scale + split planes in the workspace);  W == NULL: RUN on a workspace prepared earlier
  SVR_CHECK((in || W) && (!in || out) && workspace, SVR_E_BADARG, "conv3d_fwd_f16x3: null pointer");
  SVR_CHECK(Ci % 16 == 0 && Co >= 1, SVR_E_UNSUPPORTED, "conv3d_fwd_f16x3: need Ci %% 16 == 0 (Ci=%d Co=%d)", Ci, Co);
  hipStream_t s = (hipStream_t)stream;
  uint32_t *amax = (uint32_t *)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  uint16_t *p0 = (uint16_t *)(amax + 64);
  const int64_t ps = (int64_t)27 * Ci * Co;
  if (W) {
    (void)hipMemsetAsync(amax, 0, sizeof(uint32_t), s);
    hipLaunchKernelGGL(w_amax_kernel, dim3((unsigned)cdiv(ps, 1024)), dim3(256), 0, s, W, ps, (int64_t)1, ps, amax);
    hipLaunchKernelGGL(pack_fwd_planes_f16_kernel, dim3(cdiv(27 * Co * (Ci / 2), 256)), dim3(256), 0, s, W, amax, p0, p0 + ps, Ci, Co);
  }
  if (!in) return launch_status("conv3d_fwd_f16x3 (prepare)");
  SVR_CHECK(B > 0 && D > 0 && H > 0 && Wd > 0, SVR_E_BADSHAPE, "conv3d_fwd_f16x3: empty volume");
  SVR_CHECK(epilogue == SVR_EPI_NONE || ((epilogue == SVR_EPI_BIAS || epilogue == SVR_EPI_BIAS_RELU) && bias), SVR_E_BADARG,
            "conv3d_fwd_f16x3: epilogue %d", epilogue);
  ConvShape sh{B, D, H, Wd, Ci, Co};
  const int nbz = (int)cdiv(D, BRZ), nby = (int)cdiv(H, BRY), nbx = (int)cdiv(Wd, BRX);
  const unsigned bricks = (unsigned)((int64_t)B * nbz * nby * nbx);
#define LAUNCH_H3(CKV, TNV)                                                                                                 \
  hipLaunchKernelGGL((conv3d_brick_x3_kernel<CKV, TNV, 2, true>), dim3(bricks, (unsigned)cdiv(Co, TNV * 32)), dim3(256), 0, s, \
                     in, p0, ps, bias, out, (const float *)nullptr, sh, nbz, nby, nbx, epilogue, amax, spart)
  int tn = Co <= 32 ? 1 : (Co <= 64 ? 2 : 4);
  while (tn > 1 && (int64_t)bricks * cdiv(Co, tn * 32) < 512) tn /= 2;   // see svr_conv3d_k3_bwd_data_bf16x3
  const int nbz2 = (int)cdiv(D, 2 * BRZ);
  if (tn == 1 && (int64_t)B * nbz2 * nby * nbx * cdiv(Co, 32) >= 512) {  // two z-slices per wave (8x4x8 bricks, 16-channel chunks)
    if (persistent_bricks(sh))
      launch_brick_p<16, true, 2>(in, p0, ps, bias, out, nullptr, sh, nbz2, nby, nbx, (int)cdiv(Co, 32), epilogue, amax, spart, s);
    else
    hipLaunchKernelGGL((conv3d_brick_x3_kernel<16, 1, 2, true, 2>), dim3((unsigned)((int64_t)B * nbz2 * nby * nbx), (unsigned)cdiv(Co, 32)),
                       dim3(256), 0, s, in, p0, ps, bias, out, (const float *)nullptr, sh, nbz2, nby, nbx, epilogue, amax, spart);
  } else if (Ci % 32 == 0) {
    if (tn == 1) LAUNCH_H3(32, 1); else if (tn == 2) LAUNCH_H3(32, 2); else LAUNCH_H3(32, 4);
  } else {
    if (tn == 1) LAUNCH_H3(16, 1); else if (tn == 2) LAUNCH_H3(16, 2); else LAUNCH_H3(16, 4);
  }
#undef LAUNCH_H3
  return launch_status("conv3d_fwd_f16x3");
}
}  // namespace
