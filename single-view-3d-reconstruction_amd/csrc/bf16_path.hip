// bf16-STORAGE throughput mode of the query path (gather + point MLP forward), gfx950.
//
// north_star / BASELINE configs[1]: "64^3 grid, 10k points, batch 4, bf16 (grid_sample + MLP kernel only)".  The
// reference's hooks are the dtype-generic calls at model/ifnet.py:161,166 (grid_sample / conv in the tensor's dtype) and
// util/arguments.py:30 (--precision).  Here: feature VOLUMES and feature ROWS are stored in bf16 (half the gather's
// bytes, half the fc_0 operand bytes), all arithmetic is f32 -- corner weights, the 8-corner sum, the MFMA accumulators
// (v_mfma_f32_32x32x16_bf16), bias and ReLU -- and each result is rounded to bf16 once (round to nearest even).  The
// sample geometry is the f32 code of gather_common.h, so the voxel-index gather stays BIT-EXACT.  This is a separately
// named mode: it is never the default and never graded against the 1e-4 fp32 gate (bf16 has 8 mantissa bits; the
// reference's own bf16 run differs from its fp32 run by 2e-2 .. 9e-2 on the logits, tests/golden/ifnet_bf16_*.npz).
//
// Build with -ffp-contract=off (gather_common.h).
#include "common.h"
#include "gather_common.h"

using namespace svr;

namespace {

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float bf_lo(uint32_t p) { return __uint_as_float(p << 16); }          // element 0 of a pair
__device__ __forceinline__ float bf_hi(uint32_t p) { return __uint_as_float(p & 0xffff0000u); }  // element 1
__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {  // round to nearest even (v_cvt_pk_bf16_f32), NaN safe
  f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ uint16_t to_bf16(float a) { return (uint16_t)(pack_bf16(a, 0.f) & 0xffffu); }

__global__ __launch_bounds__(256) void cast_bf16_kernel(const float *__restrict__ in, uint16_t *__restrict__ out, int64_t n) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i + 3 < n) {
    const float4 v = *reinterpret_cast<const float4 *>(in + i);
    *reinterpret_cast<uint2 *>(out + i) = make_uint2(pack_bf16(v.x, v.y), pack_bf16(v.z, v.w));
  } else {
    for (int64_t k = i; k < n; ++k) out[k] = to_bf16(in[k]);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// gather: one level per launch.  A wave owns 64 consecutive (point, displacement) items: phase 1, lane l evaluates the
// geometry of item l once (element offset of corner (0,0,0), validity bits, six 1-D weights, output offset); phase 2,
// V = C/8 iterations of 64/V items each: the owner's values are broadcast with ds_bpermute, the eight 16-byte corner
// loads (8 bf16 channels per lane) go out together from clamped offsets, and are summed in ATen's corner order in f32.
// ------------------------------------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(256) void gather_fwd_bf16_kernel(const uint16_t *__restrict__ vol, const float *__restrict__ points,
                                                              uint16_t *__restrict__ feat, const int32_t *__restrict__ order,
                                                              int64_t BN, int N, int D, int H, int W, int col, int row_stride,
                                                              float disp, int ac) {
  constexpr int V = C / 8, IPI = 64 / V;
  const int lane = threadIdx.x & 63;
  const int64_t items = BN * 7, first = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
  if (first >= items) return;
  const int64_t item = min(first + lane, items - 1);
  const int j = (int)(item % 7);
  const int64_t pidx = item / 7;
  const int64_t pn = order ? (int64_t)order[pidx] : pidx;
  const int b = (int)(pn / N);
  const Corner c = sample_corner(points + pn * 3, j, disp, D, H, W, ac);
  const Weights w = corner_weights(c);
  int vmask = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int z = w.z0 + (k >> 2), y = w.y0 + ((k >> 1) & 1), x = w.x0 + (k & 1);
    if (z >= 0 && z < D && y >= 0 && y < H && x >= 0 && x < W) vmask |= 1 << k;
  }
  if (first + lane < items) vmask |= 0x100;  // live
  // 64-bit offsets (elements): in range for every valid corner; the others are never dereferenced
  const int64_t ebase = ((((int64_t)b * D + w.z0) * H + w.y0) * W + w.x0) * C;
  const int64_t rowoff = pn * row_stride + col + j * C;
  const int q8 = (lane % V) * 8;
  const int64_t cz = (int64_t)H * W * C, cy = (int64_t)W * C;
#pragma unroll 1
  for (int it = 0; it < V; ++it) {
    const int src = (it * IPI + lane / V) << 2;  // byte index for ds_bpermute
    const int elo = __builtin_amdgcn_ds_bpermute(src, (int)(ebase & 0xffffffff)), ehi = __builtin_amdgcn_ds_bpermute(src, (int)(ebase >> 32));
    const int rlo = __builtin_amdgcn_ds_bpermute(src, (int)(rowoff & 0xffffffff)), rhi = __builtin_amdgcn_ds_bpermute(src, (int)(rowoff >> 32));
    const int m = __builtin_amdgcn_ds_bpermute(src, vmask);
    const float wx0 = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(w.wx[0])));
    const float wx1 = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(w.wx[1])));
    const float wy0 = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(w.wy[0])));
    const float wy1 = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(w.wy[1])));
    const float wz0 = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(w.wz[0])));
    const float wz1 = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(w.wz[1])));
    const int64_t e = ((int64_t)ehi << 32) | (uint32_t)elo, ro = ((int64_t)rhi << 32) | (uint32_t)rlo;
    uint4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int64_t off = e + (k >> 2) * cz + ((k >> 1) & 1) * cy + (k & 1) * C + q8;
      v[k] = *reinterpret_cast<const uint4 *>(vol + (((m >> k) & 1) ? off : (int64_t)q8));
    }
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if ((m >> k) & 1) {
        const float wt = (((k & 1) ? wx1 : wx0) * (((k >> 1) & 1) ? wy1 : wy0)) * ((k >> 2) ? wz1 : wz0);
        const uint32_t p[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          acc[2 * i] = acc[2 * i] + bf_lo(p[i]) * wt;
          acc[2 * i + 1] = acc[2 * i + 1] + bf_hi(p[i]) * wt;
        }
      }
    }
    if (m & 0x100)
      *reinterpret_cast<uint4 *>(feat + ro + q8) = make_uint4(pack_bf16(acc[0], acc[1]), pack_bf16(acc[2], acc[3]),
                                                              pack_bf16(acc[4], acc[5]), pack_bf16(acc[6], acc[7]));
  }
}

// C == 1 (the raw input grid) and the zero padding columns [pad_start, row_stride) of every row
__global__ __launch_bounds__(256) void gather_fwd_bf16_c1_kernel(const uint16_t *__restrict__ vol, const float *__restrict__ points,
                                                                 uint16_t *__restrict__ feat, const int32_t *__restrict__ order,
                                                                 int64_t BN, int N, int D, int H, int W, int col, int row_stride,
                                                                 float disp, int ac, int pad_start) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= BN * 7) return;
  const int j = (int)(gid % 7);
  const int64_t pidx = gid / 7;
  const int64_t pn = order ? (int64_t)order[pidx] : pidx;
  const int b = (int)(pn / N);
  const Corner c = sample_corner(points + pn * 3, j, disp, D, H, W, ac);
  const Weights w = corner_weights(c);
  const uint16_t *vb = vol + (size_t)b * D * H * W;
  float acc = 0.f;
#pragma unroll
  for (int dz = 0; dz < 2; ++dz)
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const int z = w.z0 + dz, y = w.y0 + dy, x = w.x0 + dx;
        if (z >= 0 && z < D && y >= 0 && y < H && x >= 0 && x < W) {
          const float wt = (w.wx[dx] * w.wy[dy]) * w.wz[dz];
          acc = acc + __uint_as_float((uint32_t)vb[((size_t)z * H + y) * W + x] << 16) * wt;
        }
      }
  uint16_t *row = feat + pn * row_stride;
  row[col + j] = to_bf16(acc);
  if (j == 0)
    for (int cc = pad_start; cc < row_stride; ++cc) row[cc] = 0;
}

// ------------------------------------------------------------------------------------------------------------------
// Y[M,N] (bf16) = epi( X[M,K] (bf16) W[N,K]^T (bf16) ), f32 accumulation.  128 x 128 tile, 4 waves x (2x2) tiles of
// v_mfma_f32_32x32x16_bf16, k-step 32, two LDS stages with the loads two steps ahead (the structure of
// gemm_f16x3.hip's kernel without the split: operands go global -> registers -> LDS as they are).
// LDS rows are 32 bf16 + 16 B pad (80 B): the ds_read_b128 fragment reads of 16 consecutive rows cover all 64 banks.
// ------------------------------------------------------------------------------------------------------------------
constexpr int BK = 32;          // reduction elements per step
constexpr int BLW = 20;         // dwords per LDS row
constexpr int BTM = 128, BTN = 128;
constexpr int BPLANE = BTM * BLW;

__device__ __forceinline__ bf16x8 read_frag_b(const uint32_t *plane, int row, int kk, int lh) {
  const uint4 q = *reinterpret_cast<const uint4 *>(plane + row * BLW + kk * 8 + lh * 4);
  return __builtin_bit_cast(bf16x8, q);
}

__global__ __launch_bounds__(256, 3) void linear_nt_bf16_kernel(const uint16_t *__restrict__ X, int64_t ldx,
                                                                const uint16_t *__restrict__ Wt, int64_t ldw,
                                                                const float *__restrict__ bias, uint16_t *__restrict__ Y,
                                                                int64_t ldy, int64_t M, int64_t N, int64_t K, int relu) {
  __shared__ uint32_t lds[2 * 2 * BPLANE];  // two stages of (X tile, W tile)
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int l31 = lane & 31, lh = lane >> 5;
  const int64_t ntn = cdiv(N, BTN), lidx = xcd_logical(blockIdx.x, gridDim.x);
  if (lidx >= ntn * cdiv(M, BTM)) return;
  const int64_t n0 = (lidx % ntn) * BTN, m0 = (lidx / ntn) * BTM;
  // loaders: 128 rows x 32 bf16 = 512 x 16 B per operand: thread -> rows t/4 and t/4 + 64, 16-byte part t%4
  const uint16_t *xp[2], *wp[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int64_t xr = m0 + (t >> 2) + 64 * i, wrw = n0 + (t >> 2) + 64 * i;
    xr = xr < M ? xr : M - 1;      // clamped rows only feed outputs the guarded epilogue never stores
    wrw = wrw < N ? wrw : N - 1;
    xp[i] = X + xr * ldx + (t & 3) * 8;
    wp[i] = Wt + wrw * ldw + (t & 3) * 8;
  }
  // two register sets (A: even steps, B: odd steps) as plain variables: a struct passed by reference into the step
  // lambdas ended up in scratch memory (144 B per lane, 24 scratch instructions per step)
  uint4 xa0, xa1, wa0, wa1, xb0, xb1, wb0, wb1;
  const int64_t klast = K - BK;
#define SVR_LOAD(x0, x1, w0, w1, kk0)                                      \
  {                                                                        \
    int64_t kq = (kk0);                                                    \
    kq = kq < klast ? kq : klast; /* past the end: re-read the last step */ \
    x0 = *reinterpret_cast<const uint4 *>(xp[0] + kq);                     \
    x1 = *reinterpret_cast<const uint4 *>(xp[1] + kq);                     \
    w0 = *reinterpret_cast<const uint4 *>(wp[0] + kq);                     \
    w1 = *reinterpret_cast<const uint4 *>(wp[1] + kq);                     \
  }
  const int soff = (t >> 2) * BLW + (t & 3) * 4;
#define SVR_STORE(x0, x1, w0, w1, st)                                      \
  {                                                                        \
    uint32_t *sp = (st);                                                   \
    *reinterpret_cast<uint4 *>(sp + soff) = x0;                            \
    *reinterpret_cast<uint4 *>(sp + soff + 64 * BLW) = x1;                 \
    *reinterpret_cast<uint4 *>(sp + BPLANE + soff) = w0;                   \
    *reinterpret_cast<uint4 *>(sp + BPLANE + soff + 64 * BLW) = w1;        \
  }
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  auto mma = [&](const uint32_t *pa) {  // fragments of one LDS stage -> 8 MFMAs
    const uint32_t *pb = pa + BPLANE;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a[i] = read_frag_b(pa, wr * 64 + i * 32 + l31, kk, lh);
        b[i] = read_frag_b(pb, wc * 64 + i * 32 + l31, kk, lh);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  };
  uint32_t *st0 = lds, *st1 = lds + 2 * BPLANE;
  SVR_LOAD(xa0, xa1, wa0, wa1, 0)
  SVR_LOAD(xb0, xb1, wb0, wb1, BK)
  SVR_STORE(xa0, xa1, wa0, wa1, st0)
  __syncthreads();
  // step k0 on stage 0: set A (stored one step ago) is refilled with step k0+2, set B (step k0+1, in flight) goes to
  // stage 1 after the MFMAs; then the same with the roles swapped
  for (int64_t k0 = 0; k0 < K; k0 += 2 * BK) {
    SVR_LOAD(xa0, xa1, wa0, wa1, k0 + 2 * BK)
    mma(st0);
    SVR_STORE(xb0, xb1, wb0, wb1, st1)
    __syncthreads();
    if (k0 + BK < K) {
      SVR_LOAD(xb0, xb1, wb0, wb1, k0 + 3 * BK)
      mma(st1);
      SVR_STORE(xa0, xa1, wa0, wa1, st0)
      __syncthreads();
    }
  }
#undef SVR_LOAD
#undef SVR_STORE
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int64_t n = n0 + wc * 64 + j * 32 + l31;
      const float bv = (bias && n < N) ? bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t m = m0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m < M && n < N) {
          float v = acc[i][j][r] + bv;
          if (relu) v = fmaxf(v, 0.f);
          Y[m * ldy + n] = to_bf16(v);
        }
      }
    }
}

__global__ __launch_bounds__(256) void fc_out_fwd_bf16_kernel(const uint16_t *__restrict__ Hm, int64_t ldh, const float *__restrict__ w,
                                                              const float *__restrict__ b, float *__restrict__ logits,
                                                              const int32_t *__restrict__ row_map, int64_t M, int64_t K) {
  const int64_t row = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
  const int sub = threadIdx.x & 15;
  float s = 0.f;
  if (row < M) {
    const uint16_t *h = Hm + row * ldh;
    for (int64_t k = sub * 8; k < K; k += 128) {
      const uint4 a = *reinterpret_cast<const uint4 *>(h + k);
      const uint32_t p[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) s += bf_lo(p[i]) * w[k + 2 * i] + bf_hi(p[i]) * w[k + 2 * i + 1];
    }
  }
#pragma unroll
  for (int o = 8; o >= 1; o >>= 1) s += __shfl_xor(s, o);
  if (row < M && sub == 0) logits[row_map ? (int64_t)row_map[row] : row] = s + b[0];
}

}  // namespace

extern "C" int svr_cast_f32_to_bf16(const float *in, uint16_t *out, int64_t n, void *stream) {
  if (n <= 0) return SVR_OK;
  SVR_CHECK(in && out, SVR_E_BADARG, "cast_bf16: null pointer");
  SVR_CHECK((((uintptr_t)in) & 15) == 0 && (((uintptr_t)out) & 7) == 0, SVR_E_ALIGN, "cast_bf16: unaligned pointer");
  hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)cdiv(n, 1024)), dim3(256), 0, (hipStream_t)stream, in, out, n);
  return launch_status("cast_bf16");
}

extern "C" int svr_gather_trilinear_fwd_bf16(const svr_gather_desc *d, const float *points, uint16_t *features, void *stream) {
  SVR_CHECK(d != nullptr && d->n_levels >= 1 && d->n_levels <= SVR_MAX_LEVELS, SVR_E_BADARG, "gather_fwd_bf16: bad descriptor");
  SVR_CHECK(d->B >= 0 && d->N >= 0 && d->row_stride % 8 == 0, SVR_E_BADSHAPE, "gather_fwd_bf16: B=%d N=%d row_stride=%d (8 | stride)",
            d->B, d->N, d->row_stride);
  const int64_t BN = (int64_t)d->B * d->N;
  if (BN == 0) return SVR_OK;
  SVR_CHECK(points && features && (((uintptr_t)features) & 15) == 0, SVR_E_BADARG, "gather_fwd_bf16: null / unaligned pointer");
  hipStream_t s = (hipStream_t)stream;
  int pad_start = 0, c1_levels = 0;
  for (int l = 0; l < d->n_levels; ++l) {
    const svr_level &L = d->level[l];
    SVR_CHECK(L.vol && L.D > 0 && L.H > 0 && L.W > 0, SVR_E_BADARG, "gather_fwd_bf16: level %d has no volume", l);
    SVR_CHECK(L.C == 1 || L.C == 16 || L.C == 32 || L.C == 64 || L.C == 128, SVR_E_UNSUPPORTED, "gather_fwd_bf16: level %d: C=%d", l, L.C);
    SVR_CHECK(L.col >= 0 && L.col + 7 * L.C <= d->row_stride && (L.C == 1 || L.col % 8 == 0) && (((uintptr_t)L.vol) & 15) == 0,
              SVR_E_ALIGN, "gather_fwd_bf16: level %d: columns / alignment", l);
    pad_start = pad_start > L.col + 7 * L.C ? pad_start : L.col + 7 * L.C;
    c1_levels += L.C == 1;
  }
  SVR_CHECK(c1_levels >= 1 || pad_start == d->row_stride, SVR_E_UNSUPPORTED,
            "gather_fwd_bf16: the padding columns are written by the C == 1 level's kernel; none given");
  const unsigned wgrid = (unsigned)cdiv(BN * 7, 256);
  bool pad_done = false;
  for (int l = 0; l < d->n_levels; ++l) {
    const svr_level &L = d->level[l];
    const uint16_t *vol = (const uint16_t *)L.vol;
#define SVR_GB(CC)                                                                                                           \
  hipLaunchKernelGGL((gather_fwd_bf16_kernel<CC>), dim3(wgrid), dim3(256), 0, s, vol, points, features, d->order, BN, d->N, L.D, \
                     L.H, L.W, L.col, d->row_stride, d->displacement, d->align_corners)
    switch (L.C) {
      case 16: SVR_GB(16); break;
      case 32: SVR_GB(32); break;
      case 64: SVR_GB(64); break;
      case 128: SVR_GB(128); break;
      default:
        hipLaunchKernelGGL(gather_fwd_bf16_c1_kernel, dim3((unsigned)cdiv(BN * 7, 256)), dim3(256), 0, s, vol, points, features,
                           d->order, BN, d->N, L.D, L.H, L.W, L.col, d->row_stride, d->displacement, d->align_corners,
                           pad_done ? d->row_stride : pad_start);
        pad_done = true;
    }
#undef SVR_GB
  }
  return launch_status("gather_fwd_bf16");
}

extern "C" int svr_linear_fwd_bf16(const uint16_t *X, int64_t ldx, const uint16_t *W, int64_t ldw, const float *bias, uint16_t *Y,
                                   int64_t ldy, int64_t M, int64_t N, int64_t K, int epilogue, void *stream) {
  if (M == 0) return SVR_OK;
  SVR_CHECK(X && W && Y, SVR_E_BADARG, "linear_fwd_bf16: null pointer");
  SVR_CHECK(M > 0 && N > 0 && K > 0 && K % BK == 0, SVR_E_BADSHAPE, "linear_fwd_bf16: M=%ld N=%ld K=%ld (32 | K)", (long)M, (long)N, (long)K);
  SVR_CHECK(ldx % 8 == 0 && ldw % 8 == 0 && (((uintptr_t)X | (uintptr_t)W) & 15) == 0, SVR_E_ALIGN,
            "linear_fwd_bf16: X / W rows must be 16-byte aligned");
  SVR_CHECK(epilogue == SVR_EPI_NONE || ((epilogue == SVR_EPI_BIAS || epilogue == SVR_EPI_BIAS_RELU) && bias), SVR_E_BADARG,
            "linear_fwd_bf16: epilogue %d", epilogue);
  dim3 grid(xcd_grid(cdiv(N, BTN) * cdiv(M, BTM)));
  hipLaunchKernelGGL(linear_nt_bf16_kernel, grid, dim3(256), 0, (hipStream_t)stream, X, ldx, W, ldw,
                     epilogue == SVR_EPI_NONE ? nullptr : bias, Y, ldy, M, N, K, epilogue == SVR_EPI_BIAS_RELU ? 1 : 0);
  return launch_status("linear_fwd_bf16");
}

extern "C" int svr_fc_out_fwd_bf16(const uint16_t *H, int64_t ldh, const float *w, const float *b, float *logits,
                                   const int32_t *row_map, int64_t M, int64_t K, void *stream) {
  if (M <= 0) return SVR_OK;
  SVR_CHECK(H && w && b && logits, SVR_E_BADARG, "fc_out_fwd_bf16: null pointer");
  SVR_CHECK(K > 0 && K % 8 == 0 && ldh % 8 == 0 && (((uintptr_t)H) & 15) == 0, SVR_E_BADSHAPE, "fc_out_fwd_bf16: K=%ld ldh=%ld", (long)K, (long)ldh);
  hipLaunchKernelGGL(fc_out_fwd_bf16_kernel, dim3((unsigned)cdiv(M * 16, 256)), dim3(256), 0, (hipStream_t)stream, H, ldh, w, b,
                     logits, row_map, M, K);
  return launch_status("fc_out_fwd_bf16");
}
