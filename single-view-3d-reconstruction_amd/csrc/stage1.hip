// First encoder stage of the 128-architecture as ONE recomputed unit, gfx950:
//     a = relu(conv_in(x) + bias)   x (B,D,H,W,1) -> a (B,D,H,W,16)          reference model/ifnet.py:126,165
//     y = BatchNorm3d(a)            sampled by the gather AND pooled          :138,165-166
//     p = MaxPool3d(2)(y)                                                     :136,169
// and its autograd.  The stage works on the full-resolution grid (1.07 GB per 16-channel tensor at config 3), every pass
// over such a tensor is HBM time (0.2 ms), and `a` is the only tensor of the stage that costs nothing to make again:
// 27 taps x 16 outputs from ONE input channel = 7 MFMAs (16x16x4, exact f32) per 16 voxels from a 4 KB halo tile in LDS.
// So `a` is never stored.  Four passes, all of them the same brick loop with a different tail:
//
//   forward   STATS      a -> per-channel sum / sum of squares (f64 partials)            reads x
//             APPLY      a -> y, pooled, argmax                                           reads x, writes y (+ 1/8)
//   backward  BWD_REDUCE a, dy_total -> sum dy, sum dy*xhat                               reads x, dy, dpooled
//             BWD_APPLY  a, dy_total -> dconv (registers) -> dW (27 x 16), db [, dconv]   reads x, dy, dpooled
//
// against conv (write a) + BN apply (read a, write y) and BN reduce (read a, dy) + BN apply (read a, dy, write dconv) +
// weight gradient (read dconv) before: 3 of 5 + 5 of 7 full-resolution tensor passes are gone, and 2 x 1.07 GB of
// saved / temporary tensors.  The recomputed `a` is bit-identical in all four passes (same MFMA sequence), so the ReLU
// mask and xhat of the backward are exactly the forward's.
//
// Arithmetic of the recomputed convolution: either exact f32 (7 x v_mfma_f32_16x16x4_f32 per tile: 224 matrix cycles) or
// the 3-product f16 split of the other forward convolutions (template flag H3; f16x3.h / gemm_f16x3.hip: f32-level accuracy
// for |x| < 65504): the 27 taps fit ONE k = 32 step of v_mfma_f32_16x16x32_f16, three of them per tile = 48 matrix cycles.
// The halo tile then holds each voxel as a packed pair (f16 hi | f16 lo * 2^11), split once per brick by the threads that
// stage it; a lane fetches its 8 taps and sorts the halves into the two A fragments with 8 v_perm_b32.  The weights are
// normalised by 2^s (amax of the 432 weights, found by the workgroup itself) and the accumulator is scaled back by 2^-s.
//
// Tile = 16 voxels = two x-adjacent 2x2x2 pool cells.  v_mfma_f32_16x16x4_f32 lane mapping (lane = 16 kq + l15):
//   A[i = l15][k = kq], B[k = kq][j = l15], D[i = 4 kq + r][j = l15] in register r.
// conv:  i = voxel, k = tap (7 MFMAs cover taps 0..27), j = output channel; voxel i = 8 cell + 4 dz + 2 dy + dx, so a lane's
//        four results are the (dy, dx) voxels of one z-slice of a cell for ONE channel, and the cell's other z-slice sits
//        16 lanes away (one ds_bpermute for the pool maximum).
// dW:    i = tap (two 16-row tiles), k = voxel, j = output channel; B is the lane's own dconv register r (voxel 4 kq + r),
//        A one LDS read of the halo tile at (voxel + tap).
// v_mfma_f32_16x16x32_f16 (H3): A[i = l15][k = 8 kq + m], B[k = 8 kq + m][j = l15], m = 0..7; D as above.
#include "common.h"
#include "f16x3.h"

#ifndef S1_EXP
#define S1_EXP 0   // measurement switches (build.py: SVR_S1_EXP): 1 no tail, 2 no MFMA
#endif

using namespace svr;

namespace svr {
// bn_pool.hip
void bn_stats_finalize_launch(const double *part, double *stats, int64_t rows, int C, int blocks, const float *gamma,
                              const float *beta, float *rmean, float *rvar, float *ss, float *mean_f32, float eps, float momentum,
                              hipStream_t s);
void bn_sum_parts_launch(const double *part, double *out, int cols, int blocks, hipStream_t s);
// conv3d.hip
void conv3d_c1_wgrad_reduce_launch(const float *slab, float *dWp, int Co, int parts, int param_layout, const float *dbpart,
                                   float *db, hipStream_t s);
}  // namespace svr

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int SB = 8;                      // brick edge (voxels)
constexpr int HE = SB + 2;                 // halo tile edge (10 voxels)
constexpr int RS = 12;                     // halo tile row stride in LDS; plane stride PS = 122, or 123 in the pass that also
                                           // reads the dW operand.  With the natural 10 / 100 nearly every conv operand read
                                           // was a 2-way bank conflict: 864 bank cycles per wave and brick against 448
                                           // without conflicts; 12 / 122: 480 (brute force over the lane map).  The dW operand
                                           // (16 taps x 4 voxels per read) cannot get below 768 of 512; 12 / 123: 576 + 768
constexpr int NH = HE * HE * HE;           // 1000 halo voxels
constexpr int NHP = HE * 123;              // floats of the padded tile
constexpr int HIT = (NH + 255) / 256;      // halo loads per thread
constexpr int CO = 16;
constexpr int S1_MAX_BLOCKS = 2048;        // upper bound of the persistent grid (8 workgroups per CU x 256 CUs): sizes the workspace

enum { S1_STATS = 0, S1_APPLY = 1, S1_BWD_REDUCE = 2, S1_BWD_APPLY = 3 };

struct S1Args {
  const float *x, *Wp, *bias;        // input grid, conv_in weights [27][16], bias [16]
  const float *ss, *mean;            // BatchNorm scale | shift | invstd (3 x 16), mean (16)
  float *y, *pooled;                 // APPLY outputs
  uint8_t *argmax;                   // APPLY output / backward input
  const float *dy, *dpooled;         // backward inputs (either may be null)
  const double *sums;                // BWD_APPLY: sum dy, sum dy*xhat
  double *part;                      // STATS / BWD_REDUCE: per-workgroup partial sums [grid][2][16]
  float *dout, *slab, *dbpart;       // BWD_APPLY: optional dconv, per-workgroup dW [grid][32][32], db [grid][16]
  float *dgamma, *dbeta;             // BWD_APPLY: BatchNorm parameter gradients (= the sums; written by workgroup 0)
  int B, D, H, W, nbz, nby, nbx, nbricks, flags;
};

// Per-brick geometry of a lane's results: everything a tile needs is affine in (tile, r) from here, so the unrolled tile
// loop carries two integers per brick instead of recomputing clamped coordinates per voxel.  Offsets of voxels beyond an
// odd extent land on other (valid) voxels of the sample -- clamped to its last element -- and are masked by ok.
struct S1Brick {
  int lane_base;   // offset (elements) of the lane's voxel (tile 0, r = 0) inside the sample + its channel
  int pool_base;   // offset of the lane's pool cell (tile 0) inside the pooled sample + its channel
  int zl, y0, xl;  // the lane's z, the brick's y origin, the lane's x at tile 0 / dx = 0
  int cz, cy0, cx0;
  int b;
  uint32_t okm;    // bit 4 tt + r: result r of tile tt lies inside the volume
  bool interior;   // (uniform) the whole brick lies inside the volume
};
__device__ __forceinline__ S1Brick s1_brick(const S1Args &a, int brick, int wave, int ciD, int dzD, int l15) {
  S1Brick g;
  int q = brick;
  const int x0 = (q % a.nbx) * SB; q /= a.nbx;
  const int y0 = (q % a.nby) * SB; q /= a.nby;
  const int z0 = (q % a.nbz) * SB;
  g.b = q / a.nbz;
  g.zl = z0 + 2 * wave + dzD;
  g.y0 = y0;
  g.xl = x0 + 2 * ciD;
  g.lane_base = ((min(g.zl, a.D - 1) * a.H + y0) * a.W + g.xl) * CO + l15;
  g.cz = (z0 >> 1) + wave; g.cy0 = y0 >> 1; g.cx0 = (x0 >> 1) + ciD;
  const int Dp = a.D / 2, Hp = a.H / 2, Wp_ = a.W / 2;
  g.pool_base = ((min(g.cz, max(Dp - 1, 0)) * Hp + g.cy0) * Wp_ + g.cx0) * CO + l15;
  // validity of the lane's 32 results of the brick, once per brick: rows y0 + j (j = 2 (tt >> 1) + (r >> 1)) are a
  // wave-uniform prefix, columns xl + {0, 1, 4, 5} (4 (tt & 1) + (r & 1)) and the z-slice are per lane
  g.interior = z0 + SB <= a.D && y0 + SB <= a.H && x0 + SB <= a.W;
  uint32_t my = 0, mx = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {      // bits of the tiles / results in row j: tt >> 1 == j >> 1, r >> 1 == (j & 1)
    const uint32_t rowbits = (0x3u << (2 * (j & 1))) * 0x11u << (8 * (j >> 1));
    if (y0 + j < a.H) my |= rowbits;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {      // bits of column 4 (k >> 1) + (k & 1): tt & 1 == k >> 1, r & 1 == (k & 1)
    const uint32_t colbits = (0x5u << (k & 1)) * 0x01010101u << (4 * (k >> 1));
    if (g.xl + 4 * (k >> 1) + (k & 1) < a.W) mx |= colbits;
  }
  g.okm = g.zl < a.D ? (my & mx) : 0u;
  return g;
}
// tile tt, result r: voxel offset (clamped to the sample), validity
__device__ __forceinline__ uint32_t s1_off(const S1Args &a, const S1Brick &g, int tt, int r) {
  const int o = g.lane_base + (2 * (tt >> 1) + (r >> 1)) * (a.W * CO) + (4 * (tt & 1) + (r & 1)) * CO;
  return (uint32_t)min(o, a.D * a.H * a.W * CO - 1);
}
__device__ __forceinline__ bool s1_ok(const S1Brick &g, int tt, int r) { return (g.okm >> (4 * tt + r)) & 1u; }
// 0xffffffff / 0: and-mask of a result that must not count outside the volume
__device__ __forceinline__ uint32_t s1_okmask(const S1Brick &g, int tt, int r) {
  return (uint32_t)((int32_t)(g.okm << (31 - (4 * tt + r))) >> 31);
}
__device__ __forceinline__ float s1_and(float v, uint32_t m) { return __uint_as_float(__float_as_uint(v) & m); }
__device__ __forceinline__ bool s1_cell_ok(const S1Args &a, const S1Brick &g, int tt) {
  return g.cz < a.D / 2 && g.cy0 + (tt >> 1) < a.H / 2 && g.cx0 + 2 * (tt & 1) < a.W / 2;
}
__device__ __forceinline__ uint32_t s1_po(const S1Args &a, const S1Brick &g, int tt) {
  const int Dp = a.D / 2, Hp = a.H / 2, Wp_ = a.W / 2;
  const int o = g.pool_base + (tt >> 1) * (Wp_ * CO) + (tt & 1) * 2 * CO;
  return (uint32_t)max(min(o, Dp * Hp * Wp_ * CO - 1), 0);
}

// gradient inputs of one tile (backward passes), in flight for S1_PD tiles before they are used
struct S1Grad {
  float g[4];
  float dp;
  uint32_t am;
};
constexpr int S1_PD = 4;   // prefetch distance in tiles: the loads of tile t + 4 are issued when tile t is processed

template <int MODE, bool H3>
__global__ __launch_bounds__(256) void stage1_kernel(const S1Args a) {
  // f32 halo tile: the exact path's conv operand and, in BWD_APPLY, the dW operand of both paths; packed f16 pairs: H3
  __shared__ float tile[(H3 && MODE != S1_BWD_APPLY) ? 1 : NHP];
  __shared__ uint32_t tileh[H3 ? NHP : 1];
  __shared__ __attribute__((aligned(16))) double redd[(MODE == S1_BWD_APPLY) ? 2048 : 512];
  constexpr bool BWD = MODE == S1_BWD_REDUCE || MODE == S1_BWD_APPLY;
  constexpr int PS = MODE == S1_BWD_APPLY ? 123 : 122;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l15 = lane & 15, kq = lane >> 4;
  const int D = a.D, H = a.H, W = a.W;
  const int Dp = D / 2, Hp = H / 2, Wp_ = W / 2;

  // ---- lane constants of the recomputed convolution
  int toffA[7];
  float bw[7];
#pragma unroll
  for (int m = 0; m < 7; ++m) {
    const int tap = 4 * m + kq, tc = tap < 27 ? tap : 26;
    toffA[m] = (tc / 9 - 1) * PS + ((tc / 3) % 3 - 1) * RS + (tc % 3 - 1);
    bw[m] = tap < 27 ? a.Wp[tap * CO + l15] : 0.f;
  }
  // A rows: voxel i = l15 of the tile -> offset in the halo tile (tile origin = brick origin - 1)
  const int vA = (2 * wave + ((l15 >> 2) & 1) + 1) * PS + (((l15 >> 1) & 1) + 1) * RS + (2 * (l15 >> 3) + (l15 & 1) + 1);
  float bv = a.bias ? a.bias[l15] : 0.f;
  // H3: taps 8 kq .. 8 kq + 7 of this lane (28 .. 31 do not exist: weight 0, operand read from tap 26), weight fragments
  int toffH[8];
  f16x8 wh, wl, wq;
  float winv = 1.f, w_scale_up = 1.f;
  if constexpr (H3) {
    float *redf = reinterpret_cast<float *>(redd);
    float m = fabsf(a.Wp[t]);                                      // 27 x 16 = 432 weights
    if (t + 256 < 27 * CO) m = fmaxf(m, fabsf(a.Wp[t + 256]));
    redf[t] = m;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (t < o) redf[t] = fmaxf(redf[t], redf[t + o]);
      __syncthreads();
    }
    const uint32_t amax = __float_as_uint(redf[0]);
    __syncthreads();   // (redd is used again at the end)
    const float wsc = w_scale(amax, false);
    winv = w_scale(amax, true);
    w_scale_up = wsc;
    bv *= wsc;         // the bias rides in the accumulator, which carries the weights' scale
    uint32_t hp[4], lp[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      float w2[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int tap = 8 * kq + 2 * d + e;
        w2[e] = tap < 27 ? a.Wp[tap * CO + l15] * wsc : 0.f;
      }
      hp[d] = pack_f16(w2[0], w2[1]);
      const f32x2 h = unpack_f16(hp[d]);
      lp[d] = pack_f16(w2[0] - h.x, w2[1] - h.y);
    }
    union { uint4 q; f16x8 v; } uh, ul;
    uh.q = make_uint4(hp[0], hp[1], hp[2], hp[3]);
    ul.q = make_uint4(lp[0], lp[1], lp[2], lp[3]);
    wh = uh.v; wl = ul.v; wq = scale_2m11(wh);
#pragma unroll
    for (int m8 = 0; m8 < 8; ++m8) {
      const int tap = 8 * kq + m8, tc = tap < 27 ? tap : 26;
      toffH[m8] = (tc / 9 - 1) * PS + ((tc / 3) % 3 - 1) * RS + (tc % 3 - 1);
    }
  }
  // D rows: this lane's four voxels = cell ciD, slice dzD, (dy, dx) = (r >> 1, r & 1); channel l15
  const int ciD = kq >> 1, dzD = kq & 1;
  float sc = 0.f, sh = 0.f, is = 0.f, mu = 0.f, m1 = 0.f, m2 = 0.f;
  if (MODE != S1_STATS) {
    sc = a.ss[l15]; sh = a.ss[CO + l15]; is = a.ss[2 * CO + l15]; mu = a.mean[l15];
  }
  if (MODE == S1_BWD_APPLY && blockIdx.x == 0 && t < CO) {
    if (a.dbeta) a.dbeta[t] = (float)a.sums[t];
    if (a.dgamma) a.dgamma[t] = (float)a.sums[CO + t];
  }
  if (MODE == S1_BWD_APPLY && !(a.flags & 2)) {
    const double n = (double)a.B * D * H * W;
    m1 = (float)(a.sums[l15] / n);
    m2 = (float)(a.sums[CO + l15] / n);
  }
  // H3: the accumulator carries the weights' scale 2^s.  Instead of scaling every result back (4 VALU per tile), the
  // constants that meet it are scaled once -- exact, powers of two: relu(acc 2^-s) sc + sh = relu(acc) (sc 2^-s) + sh,
  // (relu(acc) 2^-s - mu) is = (relu(acc) - mu 2^s) (is 2^-s); the statistics are scaled where they are written.  `av` below is
  // the activation times 2^s in the H3 passes.
  if constexpr (H3) {
    if (MODE == S1_APPLY) sc *= winv;   // (in the backward passes sc multiplies the gradient, not the activation)
    is *= winv;
    mu *= w_scale_up;
  }
  // dW operands (BWD_APPLY): A rows = taps 16 h + l15, k = voxel 4 kq + r
  int toffW[2];
  uint32_t wmask[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int tap = 16 * h + l15, tc = tap < 27 ? tap : 26;
    toffW[h] = (tc / 9 - 1) * PS + ((tc / 3) % 3 - 1) * RS + (tc % 3 - 1);
    wmask[h] = tap < 27 ? 0xffffffffu : 0u;
  }
  const int vW = (2 * wave + dzD + 1) * PS + RS + (2 * ciD + 1);   // + (2 cy + dy) * RS + 4 cxp + dx per tile and r
  f32x4 wacc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  float dbs = 0.f;
  double d1 = 0.0, d2 = 0.0;

  // gradient loads of one tile: all issued together and UNCONDITIONALLY (a load behind a branch -- even a uniform one --
  // is waited for at the end of the branch, which would drain the whole prefetch queue): a null dy / dpooled reads x[0]
  // instead and is masked to zero WHERE THE VALUE IS USED (a mask applied here is a use of the load inside the iteration
  // that issued it: the loop then ends in s_waitcnt vmcnt(0)); wave-uniform sample base + 32-bit lane offsets
  const uint32_t dymask = a.dy ? 0xffffffffu : 0u, dpmask = a.dpooled ? 0xffffffffu : 0u;
#define S1_GRAD_LOAD(SLOT, G_, TT)                                                                          \
  {                                                                                                         \
    const float *dyb_ = a.dy ? a.dy + (int64_t)(G_).b * D * H * W * CO : a.x;                                \
    _Pragma("unroll") for (int r = 0; r < 4; ++r) SLOT.g[r] = dyb_[s1_off(a, G_, TT, r) & dymask];           \
    const int64_t pb_ = (int64_t)(G_).b * Dp * Hp * Wp_ * CO;                                               \
    const uint32_t po_ = s1_po(a, G_, TT) & dpmask;                                                         \
    SLOT.dp = (a.dpooled ? a.dpooled + pb_ : a.x)[po_];                                                     \
    SLOT.am = (a.dpooled ? a.argmax + pb_ : reinterpret_cast<const uint8_t *>(a.x))[po_];                   \
  }

  // ---- persistent brick loop; the next brick's halo is fetched into registers while this one is processed
  float hv[HIT];
  uint32_t hok = 0;
#define S1_HALO_LOAD(BRICK)                                                                                         \
  {                                                                                                                 \
    int q_ = (BRICK);                                                                                               \
    const int X0_ = (q_ % a.nbx) * SB; q_ /= a.nbx;                                                                 \
    const int Y0_ = (q_ % a.nby) * SB; q_ /= a.nby;                                                                 \
    const int Z0_ = (q_ % a.nbz) * SB;                                                                              \
    const float *inb_ = a.x + (int64_t)(q_ / a.nbz) * D * H * W;                                                    \
    hok = 0;                                                                                                        \
    _Pragma("unroll") for (int i = 0; i < HIT; ++i) {                                                               \
      const int idx = min(t + 256 * i, NH - 1);                                                                     \
      const int gz = Z0_ + idx / (HE * HE) - 1, gy = Y0_ + (idx / HE) % HE - 1, gx = X0_ + idx % HE - 1;            \
      if (gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W) hok |= 1u << i;                              \
      hv[i] = inb_[(uint32_t)((min(max(gz, 0), D - 1) * H + min(max(gy, 0), H - 1)) * W + min(max(gx, 0), W - 1))]; \
    }                                                                                                               \
  }
  int hst[HIT];   // LDS addresses of this thread's halo voxels
#pragma unroll
  for (int i = 0; i < HIT; ++i) {
    const int idx = min(t + 256 * i, NH - 1);
    hst[i] = (idx / (HE * HE)) * PS + ((idx / HE) % HE) * RS + idx % HE;
  }
  int brick = blockIdx.x;
  S1Grad slot[S1_PD];
  S1Brick gn = s1_brick(a, brick, wave, ciD, dzD, l15);      // geometry of the brick whose tiles are prefetched next
  __builtin_amdgcn_sched_barrier(0);   // (the prologue's loads in the loop's order: halo first, then the four tiles)
  S1_HALO_LOAD(brick)
  __builtin_amdgcn_sched_barrier(0);
  if (BWD) {
#pragma unroll
    for (int i = 0; i < S1_PD; ++i) S1_GRAD_LOAD(slot[i], gn, i)
  }
  __builtin_amdgcn_sched_barrier(0);
  for (; brick < a.nbricks; brick += gridDim.x) {
    const S1Brick gc = gn;
    const int nbrick = min(brick + (int)gridDim.x, a.nbricks - 1);   // (clamped: the last iteration re-reads its own brick)
    gn = s1_brick(a, nbrick, wave, ciD, dzD, l15);
    __syncthreads();  // the previous brick's readers are done with the tile
#pragma unroll
    for (int i = 0; i < HIT; ++i)   // unconditional (a predicated store pulls its load into the branch: vmcnt(0) at the join;
      //                               threads past the 1000th voxel repeat voxel 999 with the same value)
    {
      const float v = __uint_as_float(__float_as_uint(hv[i]) & (((hok >> i) & 1u) ? 0xffffffffu : 0u));
      if constexpr (!H3 || MODE == S1_BWD_APPLY) tile[hst[i]] = v;
      if constexpr (H3) {   // (f16 hi | f16 lo * 2^11) of the voxel, split once per brick
        uint32_t hi2, lo2;
        split_x(v, 0.f, hi2, lo2);
        tileh[hst[i]] = (hi2 & 0xffffu) | (lo2 << 16);
      }
    }
    __syncthreads();
    float *yb = a.y + (int64_t)gc.b * D * H * W * CO;                 // this sample's volumes (wave-uniform bases)
    float *doutb = a.dout + (int64_t)gc.b * D * H * W * CO;
    const int64_t pbase = (int64_t)gc.b * Dp * Hp * Wp_ * CO;
    float f1 = 0.f, f2 = 0.f;   // f32 runs of one brick (32 values per lane), carried in f64
    // A operand of tile 0; inside the loop the next tile's seven values are read while this tile's MFMAs run
    float ac[7];
    uint32_t pc[8];
    if constexpr (H3) {
#pragma unroll
      for (int m = 0; m < 8; ++m) pc[m] = tileh[vA + toffH[m]];
    } else {
#pragma unroll
      for (int m = 0; m < 7; ++m) ac[m] = tile[vA + toffA[m]];
    }
#pragma unroll
    for (int tt = 0; tt < 8; ++tt) {
      const int cy = tt >> 1, cxp = tt & 1;
      // the next brick's halo: issued in front of tile 4's prefetch, so that exactly 4 tiles of gradient loads are
      // younger than it when the loop comes round -- the same as behind the prologue (the wait-count pass then emits
      // vmcnt(24) for the halo instead of vmcnt(0), which would drain the prefetch queue once per brick)
      if (tt == 4) S1_HALO_LOAD(nbrick)
      S1Grad cur;
      if (BWD) {
        cur = slot[tt % S1_PD];
        if (tt + S1_PD < 8) S1_GRAD_LOAD(slot[tt % S1_PD], gc, tt + S1_PD)
        else S1_GRAD_LOAD(slot[tt % S1_PD], gn, tt + S1_PD - 8)
      }
      float an[7];
      uint32_t pn[8];
      if (tt < 7) {
        const int va = vA + (2 * ((tt + 1) >> 1)) * RS + 4 * ((tt + 1) & 1);
        if constexpr (H3) {
#pragma unroll
          for (int m = 0; m < 8; ++m) pn[m] = tileh[va + toffH[m]];
        } else {
#pragma unroll
          for (int m = 0; m < 7; ++m) an[m] = tile[va + toffA[m]];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- recompute a = relu(conv + bias) for the tile
      f32x4 acc = {bv, bv, bv, bv};   // the bias rides in the accumulator (every register of a lane is channel l15)
#if S1_EXP == 2     // measurement build: no matrix instructions
      if constexpr (H3) {
#pragma unroll
        for (int m = 0; m < 8; ++m) acc[m & 3] += __uint_as_float(pc[m]) * bw[m & 3];
      } else {
#pragma unroll
        for (int m = 0; m < 7; ++m) acc[m & 3] += ac[m] * bw[m];
      }
#else
      if constexpr (H3) {
        // the lane's 8 taps -> fragments of the hi halves and of the lo * 2^11 halves (one v_perm_b32 per dword)
        union { uint4 q; f16x8 v; } xh, xl;
        xh.q = make_uint4(__builtin_amdgcn_perm(pc[1], pc[0], 0x05040100u), __builtin_amdgcn_perm(pc[3], pc[2], 0x05040100u),
                          __builtin_amdgcn_perm(pc[5], pc[4], 0x05040100u), __builtin_amdgcn_perm(pc[7], pc[6], 0x05040100u));
        xl.q = make_uint4(__builtin_amdgcn_perm(pc[1], pc[0], 0x07060302u), __builtin_amdgcn_perm(pc[3], pc[2], 0x07060302u),
                          __builtin_amdgcn_perm(pc[5], pc[4], 0x07060302u), __builtin_amdgcn_perm(pc[7], pc[6], 0x07060302u));
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl.v, wq, acc, 0, 0, 0);   // lo(x) hi(w)
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh.v, wl, acc, 0, 0, 0);   // hi(x) lo(w)
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh.v, wh, acc, 0, 0, 0);   // hi(x) hi(w)
      } else {
#pragma unroll
        for (int m = 0; m < 7; ++m) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[m], bw[m], acc, 0, 0, 0);
      }
#endif
      float av[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) av[r] = fmaxf(acc[r], 0.f);
#if S1_EXP == 1     // measurement build: no tail (the sums take the raw accumulator)
      if (MODE == S1_STATS) { f1 += acc[0] + acc[1]; f2 += acc[2] + acc[3]; }
      if (false)
#endif

      if (MODE == S1_STATS) {
        if (gc.interior) {   // (uniform) every result counts: no masks
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            f1 += av[r];
            f2 = fmaf(av[r], av[r], f2);
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v = s1_and(av[r], s1_okmask(gc, tt, r));
            f1 += v;
            f2 = fmaf(v, v, f2);
          }
        }
      } else if (MODE == S1_APPLY) {
        float yv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) yv[r] = av[r] * sc + sh;
        if (gc.interior) {  // uniform: plain stores (a predicated store costs an exec-mask round trip each)
#pragma unroll
          for (int r = 0; r < 4; ++r) yb[s1_off(a, gc, tt, r)] = yv[r];
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (s1_ok(gc, tt, r)) yb[s1_off(a, gc, tt, r)] = yv[r];
        }
        if (a.pooled) {  // uniform
          // first maximum in (z,y,x) scan order wins, NaN propagates (ATen max_pool3d: val > max || isnan(val)); this
          // lane scans its slice, the slice dz = 1 then challenges the slice dz = 0 (strictly greater, or NaN)
          float best = yv[0];
          int k = 0;
#pragma unroll
          for (int r = 1; r < 4; ++r)
            if (yv[r] > best || yv[r] != yv[r]) { best = yv[r]; k = r; }
          const float ob = __shfl_xor(best, 16);
          const int okk = __shfl_xor(k, 16);
          if (dzD == 0 && s1_cell_ok(a, gc, tt)) {
            if (ob > best || ob != ob) { best = ob; k = 4 + okk; }
            const uint32_t po = s1_po(a, gc, tt);
            (a.pooled + pbase)[po] = best;
            if (a.argmax) (a.argmax + pbase)[po] = (uint8_t)k;
          }
        }
      } else {
        float dc[4];
        const uint32_t am = cur.am | (s1_cell_ok(a, gc, tt) ? ~dpmask : 0xffu);   // 0xff: no pooled gradient for this cell
        const float dpv = __uint_as_float(__float_as_uint(cur.dp) & dpmask);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float gg = __uint_as_float(__float_as_uint(cur.g[r]) & dymask);
          if (am == (uint32_t)(4 * dzD + r)) gg += dpv;
          const float xh = (av[r] - mu) * is;
          const uint32_t okr = s1_okmask(gc, tt, r);
          if (MODE == S1_BWD_REDUCE) {
            gg = s1_and(gg, okr);
            f1 += gg;
            f2 = fmaf(gg, xh, f2);
          } else {
            float o = sc * (gg - m1 - xh * m2);
            if ((a.flags & 1) && !(av[r] > 0.f)) o = 0.f;
            dc[r] = s1_and(o, okr);
          }
        }
        if (MODE == S1_BWD_APPLY) {
          if (a.dout) {  // uniform: d(loss)/d(conv_in output) is only needed for d(loss)/d(input grid)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (s1_ok(gc, tt, r)) doutb[s1_off(a, gc, tt, r)] = dc[r];
          }
          const int vw = vW + (2 * cy) * RS + 4 * cxp;
          float wa[8];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int vo = vw + (r >> 1) * RS + (r & 1);
            wa[2 * r] = tile[vo + toffW[0]];
            wa[2 * r + 1] = tile[vo + toffW[1]];
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            dbs += dc[r];
            const float a0 = __uint_as_float(__float_as_uint(wa[2 * r]) & wmask[0]);
            const float a1 = __uint_as_float(__float_as_uint(wa[2 * r + 1]) & wmask[1]);
            wacc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, dc[r], wacc[0], 0, 0, 0);
            wacc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, dc[r], wacc[1], 0, 0, 0);
          }
        }
      }
      if (tt < 7) {
        if constexpr (H3) {
#pragma unroll
          for (int m = 0; m < 8; ++m) pc[m] = pn[m];
        } else {
#pragma unroll
          for (int m = 0; m < 7; ++m) ac[m] = an[m];
        }
      }
      // keep the tiles in program order.  The running sums are pure arithmetic with their only use behind the loop: left
      // alone, instruction selection parks every tile's tail behind all loads of the brick (8 accumulators + 8 tiles of
      // operands live: 180 VGPRs, 2 waves per SIMD).  An empty volatile asm that "modifies" them pins them here.
      if (MODE == S1_STATS || MODE == S1_BWD_REDUCE) asm volatile("" : "+v"(f1), "+v"(f2));
      if (MODE == S1_BWD_APPLY) asm volatile("" : "+v"(wacc[0]), "+v"(wacc[1]), "+v"(dbs));
      __builtin_amdgcn_sched_barrier(0);
    }
    if (MODE == S1_STATS || MODE == S1_BWD_REDUCE) { d1 += (double)f1; d2 += (double)f2; }
  }
#undef S1_HALO_LOAD
#undef S1_GRAD_LOAD

  // ---- per-workgroup results (fixed order)
  if (MODE == S1_STATS || MODE == S1_BWD_REDUCE) {
    __syncthreads();
    redd[t * 2] = d1;
    redd[t * 2 + 1] = d2;
    __syncthreads();
    if (t < 2 * CO) {
      const int ch = t % CO, which = t / CO;
      double s = 0.0;
      for (int k = 0; k < 16; ++k) s += redd[(k * 16 + ch) * 2 + which];   // the 16 lane groups (4 waves x 4 kq) of a channel
      // STATS in the H3 arithmetic summed a 2^s and (a 2^s)^2: scaled back here, exactly (BWD_REDUCE sums gradients: no scale)
      if (H3 && MODE == S1_STATS) s *= which ? (double)winv * (double)winv : (double)winv;
      a.part[(int64_t)blockIdx.x * 2 * CO + which * CO + ch] = s;
    }
  }
  if (MODE == S1_BWD_APPLY) {
    float *red = reinterpret_cast<float *>(redd);   // 4096 floats
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave * 1024 + (16 * h + 4 * kq + r) * 32 + l15] = wacc[h][r];
    __syncthreads();
    float *o = a.slab + (int64_t)blockIdx.x * 1024;
    for (int e = t; e < 1024; e += 256) {
      const int col = e & 31;
      o[e] = col < CO ? ((red[e] + red[1024 + e]) + red[2048 + e]) + red[3072 + e] : 0.f;
    }
    __syncthreads();
    red[t] = dbs;
    __syncthreads();
    if (t < CO) {
      float s = 0.f;
      for (int i = 0; i < 16; ++i) s += red[i * 16 + t];
      a.dbpart[(int64_t)blockIdx.x * CO + t] = s;
    }
  }
}

void s1_fill(S1Args &a, int B, int D, int H, int W) {
  a.B = B; a.D = D; a.H = H; a.W = W;
  a.nbz = (int)cdiv(D, SB); a.nby = (int)cdiv(H, SB); a.nbx = (int)cdiv(W, SB);
  a.nbricks = (int)((int64_t)B * a.nbz * a.nby * a.nbx);
}

// Persistent grid of one pass: exactly the workgroups that are resident at once (CUs x occupancy of that kernel), so that
// every workgroup walks the same number of bricks -- a fixed 2048 ran as one full round plus a round at a third of the chip.
// The occupancy query is cached per kernel (immutable after the first call: the only global state of this file).
template <int MODE, bool H3>
int s1_grid(int nbricks) {
  static int resident = 0;
  if (resident == 0) {
    int dev = 0, cus = 256, per_cu = 2;
    if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, stage1_kernel<MODE, H3>, 256, 0) != hipSuccess || per_cu < 1) per_cu = 2;
    (void)hipGetLastError();
    int r = cus * per_cu;
    resident = r > S1_MAX_BLOCKS ? S1_MAX_BLOCKS : (r < 1 ? 1 : r);
  }
  return nbricks < resident ? nbricks : resident;
}

int s1_check(const char *what, int B, int D, int H, int W, int Co) {
  SVR_CHECK(B > 0 && D > 0 && H > 0 && W > 0, SVR_E_BADSHAPE, "%s: empty volume", what);
  SVR_CHECK(Co == CO, SVR_E_UNSUPPORTED, "%s: Co = %d (built for 16 output channels)", what, Co);
  // 32-bit voxel offsets inside one sample, 32-bit brick ids
  SVR_CHECK((int64_t)D * H * W * CO < (1LL << 31) && (int64_t)B * cdiv(D, SB) * cdiv(H, SB) * cdiv(W, SB) < (1LL << 31),
            SVR_E_UNSUPPORTED, "%s: volume %dx%dx%dx%d too large for 32-bit offsets", what, B, D, H, W);
  return SVR_OK;
}

// one pass: grid of the instantiation + launch
template <int MODE>
int s1_launch(const S1Args &a, int f16x3, hipStream_t s) {
  int grid;
  if (f16x3) {
    grid = s1_grid<MODE, true>(a.nbricks);
    hipLaunchKernelGGL((stage1_kernel<MODE, true>), dim3(grid), dim3(256), 0, s, a);
  } else {
    grid = s1_grid<MODE, false>(a.nbricks);
    hipLaunchKernelGGL((stage1_kernel<MODE, false>), dim3(grid), dim3(256), 0, s, a);
  }
  return grid;
}

}  // namespace

extern "C" int32_t svr_stage1_supported(int32_t B, int32_t D, int32_t H, int32_t W, int32_t Co) {
  return Co == CO && B > 0 && D > 0 && H > 0 && W > 0 && (int64_t)D * H * W * CO < (1LL << 31) &&
         (int64_t)B * cdiv(D, SB) * cdiv(H, SB) * cdiv(W, SB) < (1LL << 31);
}

extern "C" int64_t svr_stage1_workspace(int32_t B, int32_t D, int32_t H, int32_t W) {
  (void)B; (void)D; (void)H; (void)W;
  // per-workgroup partials: statistics [2][16] f64, or dW [32][32] f32 + db [16] f32
  return (int64_t)S1_MAX_BLOCKS * (1024 + CO) * (int64_t)sizeof(float) + (int64_t)S1_MAX_BLOCKS * 2 * CO * (int64_t)sizeof(double);
}

extern "C" int svr_stage1_fwd(const float *x, const float *Wp, const float *bias, const float *gamma, const float *beta,
                              float *running_mean, float *running_var, float *y, float *pooled, uint8_t *argmax,
                              float *scale_shift, float *mean_f32, double *stats, int32_t B, int32_t D, int32_t H, int32_t W,
                              int32_t Co, float eps, float momentum, int training, int f16x3, void *workspace, void *stream) {
  if (int rc = s1_check("stage1_fwd", B, D, H, W, Co)) return rc;
  SVR_CHECK(x && Wp && y && scale_shift && mean_f32 && workspace && (!training || stats), SVR_E_BADARG, "stage1_fwd: null pointer");
  SVR_CHECK(!pooled || (D >= 2 && H >= 2 && W >= 2), SVR_E_BADSHAPE, "stage1_fwd: pooled output of a volume thinner than 2");
  hipStream_t s = (hipStream_t)stream;
  S1Args a{};
  s1_fill(a, B, D, H, W);
  a.x = x; a.Wp = Wp; a.bias = bias;
  const int64_t rows = (int64_t)B * D * H * W;
  if (training) {
    a.part = (double *)workspace;
    const int grid = s1_launch<S1_STATS>(a, f16x3, s);
    bn_stats_finalize_launch(a.part, stats, rows, CO, grid, gamma, beta, running_mean, running_var, scale_shift, mean_f32, eps,
                             momentum, s);
  } else if (int rc = svr_bn_finalize(stats, gamma, beta, running_mean, running_var, scale_shift, mean_f32, rows, CO, eps,
                                      momentum, 0, stream)) {
    return rc;
  }
  a.ss = scale_shift; a.mean = mean_f32; a.y = y; a.pooled = pooled; a.argmax = argmax;
  (void)s1_launch<S1_APPLY>(a, f16x3, s);
  return launch_status("stage1_fwd");
}

extern "C" int svr_stage1_bwd(const float *x, const float *Wp, const float *bias, const float *dy, const float *dpooled,
                              const uint8_t *argmax, const float *mean_f32, const float *scale_shift, double *sums,
                              float *dgamma, float *dbeta, float *dWp, float *db, float *dout, int32_t B, int32_t D,
                              int32_t H, int32_t W, int32_t Co, int relu_mask, int f16x3, void *workspace, void *stream) {
  if (int rc = s1_check("stage1_bwd", B, D, H, W, Co)) return rc;
  SVR_CHECK(x && Wp && mean_f32 && scale_shift && sums && dWp && workspace, SVR_E_BADARG, "stage1_bwd: null pointer");
  SVR_CHECK(!dpooled || argmax, SVR_E_BADARG, "stage1_bwd: dpooled needs argmax");
  hipStream_t s = (hipStream_t)stream;
  S1Args a{};
  s1_fill(a, B, D, H, W);
  a.x = x; a.Wp = Wp; a.bias = bias; a.ss = scale_shift; a.mean = mean_f32;
  a.dy = dy; a.dpooled = dpooled; a.argmax = const_cast<uint8_t *>(argmax);
  a.flags = relu_mask;
  float *slab = (float *)workspace;
  float *dbpart = slab + (int64_t)S1_MAX_BLOCKS * 1024;
  a.part = (double *)(dbpart + (int64_t)S1_MAX_BLOCKS * CO);
  const int grid_r = s1_launch<S1_BWD_REDUCE>(a, f16x3, s);
  bn_sum_parts_launch(a.part, sums, 2 * CO, grid_r, s);
  a.sums = sums; a.slab = slab; a.dbpart = dbpart; a.dout = dout; a.dgamma = dgamma; a.dbeta = dbeta;
  const int grid_a = s1_launch<S1_BWD_APPLY>(a, f16x3, s);
  // slabs -> dW in the parameter's layout (16,1,3,3,3), partials -> db: one launch
  conv3d_c1_wgrad_reduce_launch(slab, dWp, CO, grid_a, 1, dbpart, db, s);
  return launch_status("stage1_bwd");
}
