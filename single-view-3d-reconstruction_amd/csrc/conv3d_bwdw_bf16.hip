// 3x3x3 convolution weight gradient on the bf16 matrix cores with the 3-term split ("bf16x3"), gfx950.
//
//   dW[tap][ci][co] = sum_voxel in[voxel + tap][ci] * dout[voxel][co]
//
// The reduction runs over VOXELS, so both MFMA operands need 8 consecutive voxels of one channel per lane:
// the brick's input halo tile (6x6x10 voxels) and its dout tile (4x4x8) are transposed on the way into LDS,
//   in plane  [ci][halo row (hz,hy)][12 bf16: the 10 voxels of the x-row + pad]      (hi and mid planes)
//   dout plane[co][row (vz,vy)][8 bf16]
// One 24-byte halo row read serves the three dx taps of a (dz,dy) pair: dx = -1 / +1 are dword-aligned
// sub-rows, dx = 0 is a 16-bit funnel shift (v_alignbit).  k of v_mfma_f32_32x32x16_bf16 = 2 x-rows of 8 voxels.
//   x = hi + mid, products mid*hi + hi*mid + hi*hi, f32 accumulation (~1.5e-5 relative per product; the weight
//   gradient sums 10^4..10^6 such products of mixed sign, see gemm_bf16x3.hip for the error argument).
//
// Workgroup = 8 waves, persistent over bricks: wave (g, h) owns the taps 7g .. 7g+6 (g = 3: six taps) for half of the
// brick's voxels (h), its 7 accumulator tiles (32 ci x 32 co) stay in registers over all bricks; the two halves and the
// workgroups land in partial slabs that conv3d.hip's ordered f64 reduce sums (deterministic).  (Until round 3: 6 waves
// = 3 dz x 2 halves with 9 tiles each -- two of the four SIMDs carried two waves, the other two one, and the barrier
// per brick waited for the loaded pair: 6 912 MFMA cycles per brick on the critical SIMD instead of 5 376 now.)
// LDS is double buffered: the next brick's global loads are issued before the MFMA phase and written to the other
// buffer after it, one barrier per brick.  73 KB per buffer -> one workgroup per CU.
#include "common.h"
#include "f16x3.h"

namespace svr {
void colsum_launch(const float *Y, int64_t ldy, float *out, float *part, int64_t M, int64_t N, hipStream_t s);
int64_t colsum_workspace_floats(int64_t M, int64_t N);
void conv3d_bwd_weight_reduce_launch(const float *slab, float *dWp, int Ci, int Co, int cit, int cot, int parts,
                                     hipStream_t s, int param_layout, const float *dbpart, float *db, int dbparts);
}  // namespace svr

using namespace svr;

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvShape {
  int B, D, H, W, Ci, Co;
};

constexpr int BRZ = 4, BRY = 4, BRX = 8;
constexpr int HLY = BRY + 2;
constexpr int HROWS = (BRZ + 2) * (BRY + 2);      // 36 halo x-rows
constexpr int ROWDW = 6;                          // dwords per halo x-row: 10 voxels + 2 pad (bf16)
constexpr int CIS = HROWS * ROWDW + 2;            // 218 dwords per input channel (26 mod 64: b64 reads conflict free)
constexpr int COS = BRZ * BRY * 4 + 4;            // 68 dwords per output channel (b128 reads conflict free)
constexpr int IN_PLANE = 32 * CIS, DO_PLANE = 32 * COS;
constexpr int BUF = 2 * IN_PLANE + 2 * DO_PLANE;  // dwords per stage: 18 304 (73 216 B)
constexpr int NT = 512;
constexpr int IN_ITEMS = HROWS * 5 * 8;           // (halo row, voxel pair, 4-channel group) = 1440
constexpr int DO_ITEMS = BRZ * BRY * 4 * 8;       // (row, voxel pair, 4-channel group)      = 512: one per thread
constexpr int IN_IT = (IN_ITEMS + NT - 1) / NT;   // 3: two travel in the first half of a brick's prefetch, one + the dout item in the second
static_assert(IN_IT == 3 && DO_ITEMS == NT, "prefetch phases below are written for 3 + 1 items per thread");
constexpr int TAPS_PER_WAVE = 7;                  // 27 taps over 4 tap groups: 7 / 7 / 7 / 6

__device__ __forceinline__ void split2(float x0, float x1, uint32_t &hi, uint32_t &mid) {
#if defined(SVR_WG_EXP) && SVR_WG_EXP == 3   // measurement build: no split arithmetic
  hi = __float_as_uint(x0);
  mid = __float_as_uint(x1);
  return;
#endif
  f32x2 v = {x0, x1};
  bf16x2 h = __builtin_convertvector(v, bf16x2);
  hi = __builtin_bit_cast(uint32_t, h);
  f32x2 r = {x0 - __uint_as_float(hi << 16), x1 - __uint_as_float(hi & 0xffff0000u)};
  bf16x2 m = __builtin_convertvector(r, bf16x2);
  mid = __builtin_bit_cast(uint32_t, m);
}

__device__ __forceinline__ bf16x8 frag(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
  union { uint4 q; bf16x8 v; } f;
  f.q = make_uint4(a, b, c, d);
  return f.v;
}

// F16 ("f16x3s", svr_conv3d_k3_bwd_weight_f16x3): the scaled f16 split -- dout * 2^s (its |max| brought to [2^13, 2^14)), both
// operands as hi = rn16(v), lo' = rn16((v - hi) 2^11), products lo'(in) (hi(dout) 2^-11) + (hi(in) 2^-11) lo'(dout) + hi hi with
// the 2^-11 applied to the hi fragments in registers, slabs * 2^-s: 22 mantissa bits per operand, the same three instructions.
// MFMA phase of one x-row pair (kq) for tap group G: rows R = (dz, dy) pairs touched by the taps 7G .. 7G+6
template <int G, bool F16>
__device__ __forceinline__ void wgrad_rows(const uint32_t *__restrict__ ibuf, const uint32_t *__restrict__ dbuf, int l31, int r,
                                           f32x16 (&acc)[TAPS_PER_WAVE]) {
  constexpr int T0 = G * TAPS_PER_WAVE, T1 = (T0 + TAPS_PER_WAVE < 27) ? T0 + TAPS_PER_WAVE : 27;
  const int vz = r >> 2, vy = r & 3;
  const uint4 bh = *reinterpret_cast<const uint4 *>(dbuf + l31 * COS + r * 4);
  const uint4 bm = *reinterpret_cast<const uint4 *>(dbuf + DO_PLANE + l31 * COS + r * 4);
  const bf16x8 b_hi = frag(bh.x, bh.y, bh.z, bh.w), b_mid = frag(bm.x, bm.y, bm.z, bm.w);
#pragma unroll
  for (int R = T0 / 3; R <= (T1 - 1) / 3; ++R) {
    const int dzi = R / 3, dyi = R % 3;
    const uint32_t *ph = ibuf + l31 * CIS + ((vz + dzi) * HLY + vy + dyi) * ROWDW;
    const uint2 m01 = *reinterpret_cast<const uint2 *>(ph + IN_PLANE), m23 = *reinterpret_cast<const uint2 *>(ph + IN_PLANE + 2);
    const uint32_t m4 = ph[IN_PLANE + 4];
    const uint2 h01 = *reinterpret_cast<const uint2 *>(ph), h23 = *reinterpret_cast<const uint2 *>(ph + 2);
    const uint32_t h4 = ph[4];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int tap = R * 3 + dx;
      if (tap < T0 || tap >= T1) continue;
      const int i = tap - T0;
      // one 24-byte halo row serves the three dx taps: dx = 0 / 2 are dword-aligned sub-rows, dx = 1 a 16-bit funnel shift
      const bf16x8 am = dx == 0 ? frag(m01.x, m01.y, m23.x, m23.y)
                      : dx == 2 ? frag(m01.y, m23.x, m23.y, m4)
                                : frag(__builtin_amdgcn_alignbit(m01.y, m01.x, 16), __builtin_amdgcn_alignbit(m23.x, m01.y, 16),
                                       __builtin_amdgcn_alignbit(m23.y, m23.x, 16), __builtin_amdgcn_alignbit(m4, m23.y, 16));
      const bf16x8 ah = dx == 0 ? frag(h01.x, h01.y, h23.x, h23.y)
                      : dx == 2 ? frag(h01.y, h23.x, h23.y, h4)
                                : frag(__builtin_amdgcn_alignbit(h01.y, h01.x, 16), __builtin_amdgcn_alignbit(h23.x, h01.y, 16),
                                       __builtin_amdgcn_alignbit(h23.y, h23.x, 16), __builtin_amdgcn_alignbit(h4, h23.y, 16));
      // mid plane of the input x hi plane of dout first (small terms first)
#if defined(SVR_WG_EXP) && SVR_WG_EXP == 1   // measurement build: no matrix instructions (fragments still read and shifted)
      acc[i][0] += (float)am[0] * (float)b_hi[1] + (float)ah[2] * (float)b_mid[3] + (float)ah[7];
#else
      if constexpr (F16) {
        const f16x8 fam = __builtin_bit_cast(f16x8, am), fah = __builtin_bit_cast(f16x8, ah);
        const f16x8 fbh = __builtin_bit_cast(f16x8, b_hi), fbm = __builtin_bit_cast(f16x8, b_mid);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fam, scale_2m11(fbh), acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(scale_2m11(fah), fbm, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah, fbh, acc[i], 0, 0, 0);
      } else {
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, b_hi, acc[i], 0, 0, 0);
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b_mid, acc[i], 0, 0, 0);
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b_hi, acc[i], 0, 0, 0);
      }
#endif
    }
  }
}

template <bool F16>
__global__ __launch_bounds__(NT) void conv3d_bwd_weight_x3_kernel(const float *__restrict__ in,
                                                                  const float *__restrict__ dout,
                                                                  float *__restrict__ slab, ConvShape s, int nbz, int nby,
                                                                  int nbx, int co_tiles, float *__restrict__ dbpart,
                                                                  const uint32_t *__restrict__ amax_dout) {
  __shared__ uint32_t lds[2 * BUF];
  float sy = 1.f;
  if constexpr (F16) sy = amax_dout ? w_scale(amax_dout[0], false) : 1.f;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int grp = wave & 3, half = wave >> 2;
  const int pair = blockIdx.y;
  const int ci0 = (pair / co_tiles) * 32, co0 = (pair % co_tiles) * 32;
  const int bricks = s.B * nbz * nby * nbx;   // (host: < 2^31)

  // the next brick travels in two phases: phase 0 = input items 0, 1 of this thread, phase 1 = input item 2 + its dout item
  float4 ia[2][2], da[2];
  int iok[2], dok = 0;  // bit v: voxel v of the pair is inside the volume
  // bias gradient: dout passes through this thread's registers exactly once per brick; its 4 channels (cg = t & 7,
  // NT % 8 == 0) are summed here by the workgroups of the first ci tile -> no separate pass over dout
  const bool want_db = dbpart != nullptr && ci0 == 0;
  float4 dbs = make_float4(0.f, 0.f, 0.f, 0.f);

  // (32-bit brick arithmetic: the 64-bit divisions of the first version were ~1 000 scalar instructions per brick and wave,
  // eight waves of them per CU against ~10 000 cycles per brick)
  auto load = [&](int brick, int ph) {
    int q = brick;
    const int bx = q % nbx; q /= nbx;
    const int by = q % nby; q /= nby;
    const int bz = q % nbz;
    const int b = q / nbz;
    const int z0 = bz * BRZ, y0 = by * BRY, x0 = bx * BRX;
    const float *inb = in + (int64_t)b * s.D * s.H * s.W * s.Ci;
#pragma unroll
    for (int j = 0; j < (ph == 0 ? 2 : 1); ++j) {
      const int idx = min(t + NT * (ph * 2 + j), IN_ITEMS - 1);
      const int hrow = idx / 40, rem = idx % 40, pr = rem >> 3, cg = rem & 7;
      const int gz = z0 + hrow / HLY - 1, gy = y0 + hrow % HLY - 1, gx = x0 + 2 * pr - 1;
      // unconditional loads from clamped coordinates (a load inside a branch makes the compiler wait for it at the
      // end of the branch, which serialises the whole prefetch); out-of-range voxels are zeroed in store()
      const bool rowok = gz >= 0 && gz < s.D && gy >= 0 && gy < s.H && ci0 + cg * 4 < s.Ci;
      const int cz = min(max(gz, 0), s.D - 1), cy = min(max(gy, 0), s.H - 1), cc = min(ci0 + cg * 4, s.Ci - 4);
      // (uniform base + 32-bit byte offsets, 24-bit multiplies: the host checks the ranges)
      const uint32_t ro = __umul24(__umul24(__umul24((uint32_t)cz, (uint32_t)s.H) + (uint32_t)cy, (uint32_t)s.W), (uint32_t)s.Ci) + (uint32_t)cc;
      const char *p = reinterpret_cast<const char *>(inb);
#if defined(SVR_WG_EXP) && SVR_WG_EXP == 2   // measurement build: no global loads of the next brick
      ia[j][0] = make_float4((float)idx, 1.f, (float)brick, 3.f);
      ia[j][1] = make_float4((float)(uintptr_t)p, 1.f, 2.f, 3.f);
#else
      ia[j][0] = *reinterpret_cast<const float4 *>(p + (ro + __umul24((uint32_t)min(max(gx, 0), s.W - 1), (uint32_t)s.Ci)) * 4u);
      ia[j][1] = *reinterpret_cast<const float4 *>(p + (ro + __umul24((uint32_t)min(max(gx + 1, 0), s.W - 1), (uint32_t)s.Ci)) * 4u);
#endif
      iok[j] = (rowok && gx >= 0 && gx < s.W ? 1 : 0) | (rowok && gx + 1 >= 0 && gx + 1 < s.W ? 2 : 0);
    }
    if (ph == 1) {
      const float *dob = dout + (int64_t)b * s.D * s.H * s.W * s.Co;
      const int row = t >> 5, pr = (t & 31) >> 3, cg = t & 7;
      const int gz = z0 + (row >> 2), gy = y0 + (row & 3), gx = x0 + 2 * pr;
      const bool rowok = gz < s.D && gy < s.H && co0 + cg * 4 < s.Co;
      const int cz = min(gz, s.D - 1), cy = min(gy, s.H - 1), cc = min(co0 + cg * 4, s.Co - 4);
      const uint32_t ro = __umul24(__umul24(__umul24((uint32_t)cz, (uint32_t)s.H) + (uint32_t)cy, (uint32_t)s.W), (uint32_t)s.Co) + (uint32_t)cc;
      const char *p = reinterpret_cast<const char *>(dob);
#if defined(SVR_WG_EXP) && SVR_WG_EXP == 2
      da[0] = make_float4((float)t, 1.f, (float)brick, 3.f);
      da[1] = make_float4((float)(uintptr_t)p, 1.f, 2.f, 3.f);
#else
      da[0] = *reinterpret_cast<const float4 *>(p + (ro + __umul24((uint32_t)min(gx, s.W - 1), (uint32_t)s.Co)) * 4u);
      da[1] = *reinterpret_cast<const float4 *>(p + (ro + __umul24((uint32_t)min(gx + 1, s.W - 1), (uint32_t)s.Co)) * 4u);
#endif
      dok = (rowok && gx < s.W ? 1 : 0) | (rowok && gx + 1 < s.W ? 2 : 0);
    }
  };

  auto store = [&](uint32_t *buf, int ph) {
#pragma unroll
    for (int j = 0; j < (ph == 0 ? 2 : 1); ++j) {
      const int idx = t + NT * (ph * 2 + j);
      if (idx < IN_ITEMS) {
        const int hrow = idx / 40, rem = idx % 40, pr = rem >> 3, cg = rem & 7;
        uint32_t *d = buf + (cg * 4) * CIS + hrow * ROWDW + pr;
        const float v0[4] = {ia[j][0].x, ia[j][0].y, ia[j][0].z, ia[j][0].w};
        const float v1[4] = {ia[j][1].x, ia[j][1].y, ia[j][1].z, ia[j][1].w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          uint32_t h, m;
          if constexpr (F16) split_x((iok[j] & 1) ? v0[c] : 0.f, (iok[j] & 2) ? v1[c] : 0.f, h, m);
          else split2((iok[j] & 1) ? v0[c] : 0.f, (iok[j] & 2) ? v1[c] : 0.f, h, m);
          d[c * CIS] = h;
          d[IN_PLANE + c * CIS] = m;
        }
      }
    }
    if (ph == 1) {
      uint32_t *dbuf = buf + 2 * IN_PLANE;
      const int row = t >> 5, pr = (t & 31) >> 3, cg = t & 7;
      uint32_t *d = dbuf + (cg * 4) * COS + row * 4 + pr;
      float v0[4] = {da[0].x, da[0].y, da[0].z, da[0].w};
      float v1[4] = {da[1].x, da[1].y, da[1].z, da[1].w};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        v0[c] = (dok & 1) ? v0[c] : 0.f;
        v1[c] = (dok & 2) ? v1[c] : 0.f;
        uint32_t h, m;
        if constexpr (F16) split_x(v0[c] * sy, v1[c] * sy, h, m);
        else split2(v0[c], v1[c], h, m);
        d[c * COS] = h;
        d[DO_PLANE + c * COS] = m;
      }
      if (want_db) {
        dbs.x += v0[0] + v1[0]; dbs.y += v0[1] + v1[1]; dbs.z += v0[2] + v1[2]; dbs.w += v0[3] + v1[3];
      }
    }
  };

  f32x16 acc[TAPS_PER_WAVE];
#pragma unroll
  for (int i = 0; i < TAPS_PER_WAVE; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  int brick = (int)blockIdx.x;
  if (brick < bricks) {
    load(brick, 0);
    store(lds, 0);
    load(brick, 1);
    store(lds, 1);
  }
  __syncthreads();
  int cur = 0;
  for (; brick < bricks; brick += (int)gridDim.x) {
    const int next = brick + (int)gridDim.x;
    const bool more = next < bricks;
    if (more) load(next, 0);
    const uint32_t *ibuf = lds + cur * BUF, *dbuf = ibuf + 2 * IN_PLANE;
#pragma unroll 1
    for (int kq = 0; kq < 4; ++kq) {
      const int r = 2 * (half * 4 + kq) + lh;  // x-row of the brick handled by this half-wave: r = vz*4 + vy
      switch (grp) {  // wave-uniform
        case 0: wgrad_rows<0, F16>(ibuf, dbuf, l31, r, acc); break;
        case 1: wgrad_rows<1, F16>(ibuf, dbuf, l31, r, acc); break;
        case 2: wgrad_rows<2, F16>(ibuf, dbuf, l31, r, acc); break;
        default: wgrad_rows<3, F16>(ibuf, dbuf, l31, r, acc); break;
      }
      if (kq == 1 && more) {  // first part of the next brick has arrived: park it in the other buffer, fetch the rest
        store(lds + (cur ^ 1) * BUF, 0);
        load(next, 1);
      }
    }
    if (more) store(lds + (cur ^ 1) * BUF, 1);
    __syncthreads();
    cur ^= 1;
  }

  if (want_db) {  // fixed-order sum of the 64 threads per channel group through LDS (free after the last barrier)
    float *sred = reinterpret_cast<float *>(lds);
    sred[t * 4 + 0] = dbs.x; sred[t * 4 + 1] = dbs.y; sred[t * 4 + 2] = dbs.z; sred[t * 4 + 3] = dbs.w;
    __syncthreads();
    if (t < 32) {
      const int cg = t >> 2, c = t & 3;
      float sum = 0.f;
      for (int i = 0; i < NT / 8; ++i) sum += sred[(i * 8 + cg) * 4 + c];
      if (co0 + t < s.Co) dbpart[(int64_t)blockIdx.x * s.Co + co0 + t] = sum;
    }
  }
  // slab layout shared with conv3d.hip's reduce kernel: [part][tap][pair][32 ci][32 co]
  float inv = 1.f;
  if constexpr (F16) inv = amax_dout ? w_scale(amax_dout[0], true) : 1.f;
  const int pairs = gridDim.y;
  const int64_t part = (int64_t)blockIdx.x * 2 + half;
#pragma unroll
  for (int i = 0; i < TAPS_PER_WAVE; ++i) {
    const int tap = grp * TAPS_PER_WAVE + i;
    if (tap < 27) {
      float *o = slab + ((part * 27 + tap) * pairs + pair) * 1024;
#pragma unroll
      for (int r = 0; r < 16; ++r) o[((r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + l31] = F16 ? acc[i][r] * inv : acc[i][r];
    }
  }
}

// db[co] = f64 tree (fixed order) over the workgroups' partial bias gradients; one workgroup per channel
__global__ __launch_bounds__(256) void db_reduce_kernel(const float *__restrict__ dbpart, float *__restrict__ db, int Co,
                                                        int parts) {
  __shared__ double red[256];
  const int co = blockIdx.x;
  double sum = 0.0;
  for (int p = threadIdx.x; p < parts; p += 256) sum += (double)dbpart[(int64_t)p * Co + co];
  red[threadIdx.x] = sum;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) db[co] = (float)red[0];
}

// persistent workgroups per channel-tile pair: one workgroup per CU in total (LDS bound)
int x3_parts(int B, int D, int H, int W, int Ci, int Co) {
  int64_t bricks = (int64_t)B * cdiv(D, BRZ) * cdiv(H, BRY) * cdiv(W, BRX);
  int64_t pairs = cdiv(Ci, 32) * cdiv(Co, 32);
  int64_t parts = cdiv(256, pairs);
  if (parts > bricks) parts = bricks;
  return (int)(parts < 1 ? 1 : parts);
}

}  // namespace

extern "C" int64_t svr_conv3d_k3_bwd_weight_bf16x3_workspace(int32_t B, int32_t D, int32_t H, int32_t W, int32_t Ci,
                                                             int32_t Co) {
  int64_t tiles = cdiv(Ci, 32) * cdiv(Co, 32);
  const int64_t parts = x3_parts(B, D, H, W, Ci, Co);
  return (parts * 2 * 27 * tiles * 1024 + parts * Co) * (int64_t)sizeof(float);
}

namespace {
int bwd_weight_x3(const float *in, const float *dout, float *dWp, float *db, int32_t B, int32_t D, int32_t H, int32_t W, int32_t Ci,
                  int32_t Co, void *workspace, void *stream, int param_layout, bool f16 = false, const uint32_t *amax_dout = nullptr);
}

// The weight gradient on the scaled f16 split ("f16x3s": f32 level, see svr_linear_bwd_weight_f16x3); amax_dout: svr_amax_f32 of
// dout (or the amax output of the kernel that produced it).  param_layout != 0: dW(Co,Ci,3,3,3), else the packed [tap][ci][co].
extern "C" int svr_conv3d_k3_bwd_weight_f16x3(const float *in, const float *dout, float *dW, float *db, int32_t B, int32_t D,
                                              int32_t H, int32_t W, int32_t Ci, int32_t Co, int32_t param_layout,
                                              const uint32_t *amax_dout, void *workspace, void *stream) {
  return bwd_weight_x3(in, dout, dW, db, B, D, H, W, Ci, Co, workspace, stream, param_layout ? 1 : 0, true, amax_dout);
}

extern "C" int svr_conv3d_k3_bwd_weight_bf16x3(const float *in, const float *dout, float *dWp, float *db, int32_t B,
                                               int32_t D, int32_t H, int32_t W, int32_t Ci, int32_t Co, void *workspace,
                                               void *stream) {
  return bwd_weight_x3(in, dout, dWp, db, B, D, H, W, Ci, Co, workspace, stream, 0);
}

// The same gradient written straight in the PARAMETER's layout dW(Co,Ci,3,3,3) (what autograd of nn.Conv3d returns):
// no svr_conv3d_unpack_wgrad launch behind it.
extern "C" int svr_conv3d_k3_bwd_weight_bf16x3_param(const float *in, const float *dout, float *dW, float *db, int32_t B,
                                                     int32_t D, int32_t H, int32_t W, int32_t Ci, int32_t Co, void *workspace,
                                                     void *stream) {
  return bwd_weight_x3(in, dout, dW, db, B, D, H, W, Ci, Co, workspace, stream, 1);
}

namespace {
int bwd_weight_x3(const float *in, const float *dout, float *dWp, float *db, int32_t B, int32_t D, int32_t H, int32_t W, int32_t Ci,
                  int32_t Co, void *workspace, void *stream, int param_layout, bool f16, const uint32_t *amax_dout) {
  SVR_CHECK(B > 0 && D > 0 && H > 0 && W > 0, SVR_E_BADSHAPE, "conv3d_bwd_weight_bf16x3: empty volume %dx%dx%dx%d", B, D, H, W);
  SVR_CHECK(in && dout && dWp && workspace, SVR_E_BADARG, "conv3d_bwd_weight_bf16x3: null pointer");
  SVR_CHECK(Ci >= 4 && Ci % 4 == 0 && Co % 4 == 0 && Co >= 4, SVR_E_UNSUPPORTED,
            "conv3d_bwd_weight_bf16x3: need Ci, Co %% 4 == 0 (Ci=%d Co=%d)", Ci, Co);
  SVR_CHECK((((uintptr_t)in | (uintptr_t)dout) & 15) == 0, SVR_E_ALIGN, "conv3d_bwd_weight_bf16x3: 16-byte alignment");
  // the kernel's 24-bit row multiplies and 32-bit byte offsets inside one sample, 32-bit brick ids
  SVR_CHECK((int64_t)D * H * W <= (1LL << 24) && (int64_t)D * H * W * (Ci > Co ? Ci : Co) < (1LL << 30) &&
                (int64_t)B * cdiv(D, BRZ) * cdiv(H, BRY) * cdiv(W, BRX) < (1LL << 31),
            SVR_E_UNSUPPORTED, "conv3d_bwd_weight_bf16x3: volume %dx%dx%dx%d too large for 32-bit offsets", B, D, H, W);
  hipStream_t s = (hipStream_t)stream;
  ConvShape sh{B, D, H, W, Ci, Co};
  const int cit = (int)cdiv(Ci, 32), cot = (int)cdiv(Co, 32);
  const int nbz = (int)cdiv(D, BRZ), nby = (int)cdiv(H, BRY), nbx = (int)cdiv(W, BRX);
  const int parts = x3_parts(B, D, H, W, Ci, Co);
  float *slab = (float *)workspace;
  float *dbpart = db ? slab + (int64_t)parts * 2 * 27 * cit * cot * 1024 : nullptr;
  if (f16)
    hipLaunchKernelGGL(conv3d_bwd_weight_x3_kernel<true>, dim3((unsigned)parts, (unsigned)(cit * cot)), dim3(NT), 0, s, in, dout,
                       slab, sh, nbz, nby, nbx, cot, dbpart, amax_dout);
  else
    hipLaunchKernelGGL(conv3d_bwd_weight_x3_kernel<false>, dim3((unsigned)parts, (unsigned)(cit * cot)), dim3(NT), 0, s, in, dout,
                       slab, sh, nbz, nby, nbx, cot, dbpart, (const uint32_t *)nullptr);
  // one launch: slabs -> dW (either layout) and, in its last Co workgroups, the bias-gradient partials -> db
  conv3d_bwd_weight_reduce_launch(slab, dWp, Ci, Co, cit, cot, parts * 2, s, param_layout, dbpart, db, parts);
  return launch_status("conv3d_bwd_weight_bf16x3");
}
}  // namespace
