// Backward GEMMs of the point MLP on the bf16 matrix cores with a 3-term split ("bf16x3"), gfx950.
//
//   x = hi + mid (+ 2^-16 |x|),  hi = bf16_rne(x),  mid = bf16_rne(x - hi)
//   a*b ~= a_hi*b_hi + a_hi*b_mid + a_mid*b_hi        (f32 accumulation inside the MFMA)
//
// Each product carries a relative error of ~2^-16 = 1.5e-5 with random sign (round-to-nearest splits), i.e.
// two orders of magnitude below the 1e-3 gradient noise that ReLU-mask flips already put on ANY f32
// implementation of this network (DESIGN.md "Gradient tolerance"), at ~1/5 of the matrix-core cycles of the
// exact-f32 MFMA.  It is used ONLY for the two backward products of the MLP layers,
//     dX = dY W        (svr_linear_bwd_data_bf16x3)
//     dW = dY^T X      (svr_linear_bwd_weight_bf16x3)
// never in the forward pass: logits, ReLU masks and everything the 1e-4 parity gate sees come from the f32-level
// forward kernels (gemm_f16x3.hip / gemm_bf16x6.hip / gemm.hip).
//
// Structure: 128x128 block tile, 4 waves x (2x2) tiles of v_mfma_f32_32x32x16_bf16, k-step 32, ONE LDS stage
// (two barriers per step, three workgroups per CU).  LDS holds hi/mid planes as [row][k] bf16 with an 80-byte
// row stride: rows are 16-byte aligned and 16 consecutive rows cover all 64 banks with their 16-byte pieces,
// so the fragment reads (ds_read_b128, lane <-> row) and the transposing stores (ds_write_b128) are conflict free.
// Tiles that share an operand panel are dispatched to one XCD (common.h: xcd_logical).
#include "common.h"
#include "conv2d_virt.h"
#include "f16x3.h"

using namespace svr;

namespace svr {
void colsum_launch(const float *Y, int64_t ldy, float *out, float *part, int64_t M, int64_t N, hipStream_t s);
int64_t colsum_workspace_floats(int64_t M, int64_t N);
}  // namespace svr

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int XK = 32;    // reduction elements per step
constexpr int XLD = 40;   // bf16 per LDS row (80 B: 16-byte aligned rows, and 16 consecutive rows cover all 64 banks
                          // with their 16-byte pieces -> ds_read_b128 / ds_write_b128 are conflict free)
constexpr int XLW = XLD / 2;  // ... in dwords

// (x0, x1) -> packed bf16 pairs of the hi and mid parts
__device__ __forceinline__ void split2(float x0, float x1, uint32_t &hi, uint32_t &mid) {
  f32x2 v = {x0, x1};
  bf16x2 h = __builtin_convertvector(v, bf16x2);
  hi = __builtin_bit_cast(uint32_t, h);
  f32x2 r = {x0 - __uint_as_float(hi << 16), x1 - __uint_as_float(hi & 0xffff0000u)};
  bf16x2 m = __builtin_convertvector(r, bf16x2);
  mid = __builtin_bit_cast(uint32_t, m);
}

// ---- operand loaders (ROWS tile rows, NT threads) -------------------------------------------------
// A "plane pair" in LDS is uint32_t[2][ROWS * XLW]: hi plane then mid plane, [row][k] bf16.
// Tile rows past the operand's extent are CLAMPED to the last valid row instead of being zero-filled:
// they only feed output rows/columns that the guarded epilogue never stores, and clamping keeps the load
// path free of per-element branches and 64-bit index arithmetic.  Only the reduction tail needs zeros.
//
// RowK: f32 source, tile row = operand row, k contiguous.  thread -> (row = t/8 + (NT/8) i, float4 t%8)
// F16: the scaled 3-product f16 split ("f16x3s"): the operand is multiplied by `sy` (a power of two from its |max|) and split
// into hi = rn16(x), lo' = rn16((x - hi) 2^11) (f16x3.h)
template <int ROWS, int NT, bool F16 = false>
struct RowKLoader {
  static constexpr int R = ROWS * 8 / NT;
  const float *rowp[R];
  float4 v[R];
  float sy = 1.f;
  __device__ __forceinline__ RowKLoader(const float *src, int64_t ld, int64_t row0, int64_t nrows) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < R; ++i) {
      int64_t r = row0 + (t >> 3) + (NT / 8) * i;
      r = r < nrows ? r : nrows - 1;
      rowp[i] = src + r * ld + (t & 7) * 4;
    }
  }
  __device__ __forceinline__ void load(int64_t k0, int64_t) {
#pragma unroll
    for (int i = 0; i < R; ++i) v[i] = *reinterpret_cast<const float4 *>(rowp[i] + k0);
  }
  __device__ __forceinline__ void store(uint32_t *planes) const {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < R; ++i) {
      uint32_t h0, m0, h1, m1;
      if constexpr (F16) {
        split_x(v[i].x * sy, v[i].y * sy, h0, m0);
        split_x(v[i].z * sy, v[i].w * sy, h1, m1);
      } else {
        split2(v[i].x, v[i].y, h0, m0);
        split2(v[i].z, v[i].w, h1, m1);
      }
      const int off = ((t >> 3) + (NT / 8) * i) * XLW + (t & 7) * 2;
      *reinterpret_cast<uint2 *>(planes + off) = make_uint2(h0, h1);
      *reinterpret_cast<uint2 *>(planes + ROWS * XLW + off) = make_uint2(m0, m1);
    }
  }
};

// Planes: pre-split bf16 planes [row][k] in global memory (k contiguous).  thread -> (row = t/4 + (NT/4) i, 16 B part t%4)
// (8-byte pieces: a 16-byte vector here crashes hipcc 7.2's machine copy propagation)
template <int ROWS, int NT>
struct PlaneLoader {
  static constexpr int R = ROWS * 4 / NT;
  const uint16_t *ph[R], *pm[R];
  uint2 vh[R][2], vm[R][2];
  __device__ __forceinline__ PlaneLoader(const uint16_t *hi, const uint16_t *mid, int64_t ld, int64_t row0, int64_t nrows) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < R; ++i) {
      int64_t r = row0 + (t >> 2) + (NT / 4) * i;
      r = r < nrows ? r : nrows - 1;
      ph[i] = hi + r * ld + (t & 3) * 8;
      pm[i] = mid + r * ld + (t & 3) * 8;
    }
  }
  __device__ __forceinline__ void load(int64_t k0, int64_t) {
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const uint2 *qh = reinterpret_cast<const uint2 *>(ph[i] + k0);
      const uint2 *qm = reinterpret_cast<const uint2 *>(pm[i] + k0);
      vh[i][0] = qh[0]; vh[i][1] = qh[1];
      vm[i][0] = qm[0]; vm[i][1] = qm[1];
    }
  }
  __device__ __forceinline__ void store(uint32_t *planes) const {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const int off = ((t >> 2) + (NT / 4) * i) * XLW + (t & 3) * 4;
      *reinterpret_cast<uint2 *>(planes + off) = vh[i][0];
      *reinterpret_cast<uint2 *>(planes + off + 2) = vh[i][1];
      *reinterpret_cast<uint2 *>(planes + ROWS * XLW + off) = vm[i][0];
      *reinterpret_cast<uint2 *>(planes + ROWS * XLW + off + 2) = vm[i][1];
    }
  }
};

// Transposed: f32 source S[k][col] (col contiguous); the tile row is a source COLUMN.
// thread -> (col = t % ROWS, k-pair group t / ROWS): coalesced dword loads, one packed dword store per k pair.
template <int ROWS, int NT>
struct TransLoader {
  static constexpr int KG = NT / ROWS;  // k-pair groups
  static constexpr int P = 16 / KG;     // k pairs per thread
  const float *colp;                    // &S[first k of this thread][col]
  int64_t ld;
  int kfirst;
  float v[2 * P];
  __device__ __forceinline__ TransLoader(const float *src, int64_t ld_, int64_t col0, int64_t ncols) : ld(ld_) {
    const int t = threadIdx.x;
    int64_t c = col0 + (t % ROWS);
    c = c < ncols ? c : ncols - 1;
    kfirst = 2 * (t / ROWS) * P;
    colp = src + (int64_t)kfirst * ld + c;
  }
  // kend: end of the reduction range; a full step (k0 + 32 <= kend) takes the branch-free path
  __device__ __forceinline__ void load(int64_t k0, int64_t kend) {
    const float *q = colp + k0 * ld;
    if (k0 + XK <= kend) {
#pragma unroll
      for (int p = 0; p < 2 * P; ++p) v[p] = q[p * ld];
    } else {
#pragma unroll
      for (int p = 0; p < 2 * P; ++p) v[p] = (k0 + kfirst + p < kend) ? q[p * ld] : 0.f;
    }
  }
  __device__ __forceinline__ void store(uint32_t *planes) const {
    const int t = threadIdx.x;
    uint32_t h[P], m[P];
#pragma unroll
    for (int p = 0; p < P; ++p) split2(v[2 * p], v[2 * p + 1], h[p], m[p]);
    const int off = (t % ROWS) * XLW + (t / ROWS) * P;  // multiple of 4 dwords: 16-byte stores
#pragma unroll
    for (int p = 0; p < P; p += 4) {
      *reinterpret_cast<uint4 *>(planes + off + p) = make_uint4(h[p], h[p + 1], h[p + 2], h[p + 3]);
      *reinterpret_cast<uint4 *>(planes + ROWS * XLW + off + p) = make_uint4(m[p], m[p + 1], m[p + 2], m[p + 3]);
    }
  }
};

__device__ __forceinline__ bf16x8 read_frag(const uint32_t *plane, int row, int kword) {
  // 8 bf16 = 4 dwords at [row][kword .. kword+3]; 16-byte aligned (XLW and kword are multiples of 4)
  union { uint4 q; bf16x8 v; } f;
  f.q = *reinterpret_cast<const uint4 *>(plane + row * XLW + kword);
  return f.v;
}

// Block tile (WR*64) x (WC*64), WR*WC waves, every wave a 64x64 tile = 2x2 MFMA tiles.
// lds: [stage 2][ A planes 2*TM*XLW | B planes 2*TN*XLW ] dwords
template <int WR, int WC>
struct Geo {
  static constexpr int NT = WR * WC * 64, TM = WR * 64, TN = WC * 64;
  static constexpr int STAGE = 2 * (TM + TN) * XLW;  // dwords
  static constexpr int LDS_DWORDS = STAGE;           // ONE stage: see mainloop
};

// F16: A = (hi, lo') of the operand split on the fly, B = planes (wh, wl) of a weight scaled to [2^13, 2^14):
// acc += lo' (wh 2^-11) + hi wl + hi wh  (gemm_f16x3.hip's arithmetic; the caller multiplies by 2^-(s_w + s_x))
template <int WR, int WC, bool F16 = false>
__device__ __forceinline__ void compute_step(const uint32_t *pa, f32x16 (&acc)[2][2]) {
  using G = Geo<WR, WC>;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave / WC, wc = wave % WC;
  const int l31 = lane & 31, lh = lane >> 5;
  const uint32_t *pb = pa + 2 * G::TM * XLW;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    bf16x8 ah[2], am[2], bh[2], bm[2];
    const int kw = ks * 8 + lh * 4;  // dword offset of this lane's 8 k values
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = wr * 64 + i * 32 + l31;
      ah[i] = read_frag(pa, row, kw);
      am[i] = read_frag(pa + G::TM * XLW, row, kw);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = wc * 64 + j * 32 + l31;
      bh[j] = read_frag(pb, row, kw);
      bm[j] = read_frag(pb + G::TN * XLW, row, kw);
    }
    if constexpr (F16) {
      f16x8 fah[2], fal[2], fbh[2], fbl[2], fbq[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        fah[i] = __builtin_bit_cast(f16x8, ah[i]);
        fal[i] = __builtin_bit_cast(f16x8, am[i]);
        fbh[i] = __builtin_bit_cast(f16x8, bh[i]);
        fbl[i] = __builtin_bit_cast(f16x8, bm[i]);
        fbq[i] = scale_2m11(fbh[i]);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fal[i], fbq[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[i], fbl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[i], fbh[j], acc[i][j], 0, 0, 0);
        }
    } else {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bh[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bm[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
      }
    }
  }
}

// Pipeline: the global loads of step s+1 are issued before the MFMAs of step s and converted / written to LDS
// after them.  ONE LDS stage and two barriers per k-step: SQ counters showed these kernels parked on memory
// (SQ_WAIT_ANY 0.55-0.67 of the wave cycles, matrix pipe ~20 % busy), and halving the LDS footprint doubles the
// workgroups per CU, i.e. the loads in flight; the extra barrier is covered by the other workgroups.
// (Two register sets -- loads two steps ahead -- were tried: 200..256 VGPRs, spills, 1.5-4x slower.)
template <int WR, int WC, class AL, class BL, bool F16 = false>
__device__ __forceinline__ void mainloop(AL &al, BL &bl, uint32_t *lds, int64_t kbeg, int64_t kend, f32x16 (&acc)[2][2]) {
  using G = Geo<WR, WC>;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  if (kbeg >= kend) return;
  al.load(kbeg, kend);
  bl.load(kbeg, kend);
  al.store(lds);
  bl.store(lds + 2 * G::TM * XLW);
  __syncthreads();
  for (int64_t k0 = kbeg; k0 < kend; k0 += XK) {
    const bool more = k0 + XK < kend;
    if (more) {
      al.load(k0 + XK, kend);
      bl.load(k0 + XK, kend);
    }
    __builtin_amdgcn_sched_barrier(0);  // the loads stay in front of the MFMAs (the scheduler would sink them to the stores)
    compute_step<WR, WC, F16>(lds, acc);
    __syncthreads();  // every wave has read its fragments
    if (more) {
      al.store(lds);
      bl.store(lds + 2 * G::TM * XLW);
    }
    __syncthreads();
  }
}

template <int WR, int WC, class F>
__device__ __forceinline__ void foreach_acc(f32x16 (&acc)[2][2], F f) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave / WC, wc = wave % WC;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = wc * 64 + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) f(wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col, acc[i][j][r]);
    }
}

// W[n][k] f32 -> planes [k][n] bf16 (hi, mid): the B operand of dX = dY W wants the reduction index n contiguous
__global__ void pack_planes_kernel(const float *__restrict__ W, int64_t ldw, uint16_t *__restrict__ hi,
                                   uint16_t *__restrict__ mid, int64_t N, int64_t K) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // over (k, n/2)
  if (idx >= K * (N / 2)) return;
  int64_t k = idx / (N / 2), n = (idx % (N / 2)) * 2;
  uint32_t h, m;
  split2(W[n * ldw + k], W[(n + 1) * ldw + k], h, m);
  *reinterpret_cast<uint32_t *>(hi + k * N + n) = h;
  *reinterpret_cast<uint32_t *>(mid + k * N + n) = m;
}

// dX[M,K] = dY[M,N] W[N,K] (W given as planes [K][N]); optional ReLU mask epilogue.
// Epilogue: the 128x128 f32 tile goes through LDS so that every wave-instruction writes (and reads the mask
// as) whole 512-byte row pieces with 16 bytes per lane; the MFMA register layout would give 128-byte pieces
// of 64 dword stores per lane.
constexpr int NN_TM = 128, NN_TN = 128, NN_CLD = NN_TN + 4;
// F16 (round 4): the same kernel on the scaled f16 split -- dY scaled by 2^sy from its |max| (amax_dy), W planes from
// pack_planes_f16_kernel (scaled by 2^sw, amax_w), result times 2^-(sw + sy); amax_dx (optional) is left with max|dX|.
template <bool F16>
__global__ __launch_bounds__(256, 3) void linear_nn_x3_kernel(const float *__restrict__ dY, int64_t lddy,
                                                           const uint16_t *__restrict__ Wh,
                                                           const uint16_t *__restrict__ Wm, float *__restrict__ dX,
                                                           int64_t lddx, const float *__restrict__ mask, int64_t ldmask,
                                                           int64_t M, int64_t N, int64_t K, const uint32_t *__restrict__ amax_w,
                                                           const uint32_t *__restrict__ amax_dy, uint32_t *__restrict__ amax_dx) {
  using G = Geo<2, 2>;
  static_assert(G::LDS_DWORDS >= (NN_TM / 2) * NN_CLD, "half the epilogue tile must fit the staging buffer");
  __shared__ __attribute__((aligned(16))) uint32_t lds[G::LDS_DWORDS];
  // all column tiles of a row panel run on one XCD, back to back: dY comes from HBM once, then from that L2
  const int64_t nct = cdiv(K, NN_TN), lidx = xcd_logical(blockIdx.x, gridDim.x);
  if (lidx >= nct * cdiv(M, NN_TM)) return;
  const int64_t c0 = (lidx % nct) * NN_TN, m0 = (lidx / nct) * NN_TM;
  RowKLoader<NN_TM, 256, F16> al(dY, lddy, m0, M);
  PlaneLoader<NN_TN, 256> bl(Wh, Wm, N, c0, K);
  float inv = 1.f, vmax = 0.f;
  if constexpr (F16) {
    al.sy = amax_dy ? w_scale(amax_dy[0], false) : 1.f;
    inv = w_scale(amax_w[0], true) * (amax_dy ? w_scale(amax_dy[0], true) : 1.f);
  }
  f32x16 acc[2][2];
  mainloop<2, 2, RowKLoader<NN_TM, 256, F16>, PlaneLoader<NN_TN, 256>, F16>(al, bl, lds, 0, N, acc);   // ends with a barrier: the operand buffers are free
  float *ct = reinterpret_cast<float *>(lds);
  const int t = threadIdx.x;
  const int wave = t >> 6;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {  // 64 rows per pass (the waves with wr == pass own them)
    // this pass's ReLU mask, fetched up front from clamped rows (loads inside the bounds branch below would be
    // waited for one by one) while the tile goes through LDS
    float4 k4[8];
    if (mask) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int64_t m = m0 + 64 * pass + (t >> 5) + 8 * i, c = c0 + (t & 31) * 4;
        k4[i] = *reinterpret_cast<const float4 *>(mask + (m < M ? m : M - 1) * ldmask + (c < K ? c : K - 4));
      }
    }
    if ((wave >> 1) == pass)
      foreach_acc<2, 2>(acc, [&](int row, int col, float v) { ct[(row - 64 * pass) * NN_CLD + col] = v; });
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = (t >> 5) + 8 * i, c4 = (t & 31) * 4;
      const int64_t m = m0 + 64 * pass + row, c = c0 + c4;
      if (m < M && c < K) {  // K % 4 == 0 (host-checked), so a float4 never straddles the edge
        float4 v = *reinterpret_cast<const float4 *>(ct + row * NN_CLD + c4);
        if constexpr (F16) { v.x *= inv; v.y *= inv; v.z *= inv; v.w *= inv; }
        if (mask) {
          v.x = k4[i].x > 0.f ? v.x : 0.f;
          v.y = k4[i].y > 0.f ? v.y : 0.f;
          v.z = k4[i].z > 0.f ? v.z : 0.f;
          v.w = k4[i].w > 0.f ? v.w : 0.f;
        }
        if constexpr (F16) vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        if constexpr (F16) {   // streaming store: dX rows must not push the dY rows the row panel's other column tiles re-read out
          typedef float nt4 __attribute__((ext_vector_type(4)));   // of L2 (fc_0's 800-column dX, 7 column tiles: 0.795 -> 0.766 ms)
          __builtin_nontemporal_store(nt4{v.x, v.y, v.z, v.w}, reinterpret_cast<nt4 *>(dX + m * lddx + c));
        } else {
          *reinterpret_cast<float4 *>(dX + m * lddx + c) = v;
        }
      }
    }
    __syncthreads();
  }
  if constexpr (F16) {
    if (amax_dx) svr_amax_publish(amax_dx, vmax);   // (uniform)
  }
}

// W[n][k] f32 -> f16 planes [k][n] of W 2^s (amax: max|W| -> s with max|W 2^s| in [2^13, 2^14)): wh = rn16, wl = rn16(W 2^s - wh)
__global__ void pack_planes_f16_kernel(const float *__restrict__ W, int64_t ldw, const uint32_t *__restrict__ amax,
                                       uint16_t *__restrict__ hi, uint16_t *__restrict__ lo, int64_t N, int64_t K) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // over (k, n/2)
  if (idx >= K * (N / 2)) return;
  const int64_t k = idx / (N / 2), n = (idx % (N / 2)) * 2;
  const float sc = w_scale(amax[0], false);
  const float w0 = W[n * ldw + k] * sc, w1 = W[(n + 1) * ldw + k] * sc;
  const uint32_t h = pack_f16(w0, w1);
  const f32x2 hf = unpack_f16(h);
  *reinterpret_cast<uint32_t *>(hi + k * N + n) = h;
  *reinterpret_cast<uint32_t *>(lo + k * N + n) = pack_f16(w0 - hf.x, w1 - hf.y);
}

// slab[z][N,K] = dY[rows of split z]^T X[rows of split z].  128 x 128 tile, 4 waves, 36 KB of LDS: three
// workgroups per CU keep enough loads in flight (the 256 x 128 / 8-wave variant read X only once but ran one
// workgroup per CU and was parked on memory 55 % of the time).
constexpr int TN_TM = 128, TN_TN = 128;
// dY is transposed and split ONCE per call into bf16 planes [n][m] (dy_planes_kernel below): every one of the K/128
// workgroups that share an N-tile then reads its A operand with four 8-byte loads per thread and plain LDS stores
// instead of sixteen dword loads + a transposing store.
__global__ __launch_bounds__(256) void linear_tn_x3_kernel(const uint16_t *__restrict__ dYh,
                                                           const uint16_t *__restrict__ dYm, int64_t Mpad,
                                                           const float *__restrict__ X, int64_t ldx,
                                                           float *__restrict__ slab, int64_t M, int64_t N, int64_t K,
                                                           int64_t rows_per_split, int splits) {
  using G = Geo<2, 2>;
  __shared__ __attribute__((aligned(16))) uint32_t lds[G::LDS_DWORDS];
  // all output tiles of one row range (split) run on one XCD, back to back, and walk the rows together: dY and X
  // rows come from HBM once, the other tiles read them from that XCD's L2
  const int64_t tk = cdiv(K, TN_TN), tiles = tk * cdiv(N, TN_TM), lidx = xcd_logical(blockIdx.x, gridDim.x);
  if (lidx >= tiles * splits) return;
  const int64_t split = lidx / tiles, tile = lidx % tiles;
  // the N-tiles of one K-tile (they read the same X columns) are neighbours in the dispatch order
  const int64_t tn = cdiv(N, TN_TM);
  const int64_t i0 = (tile % tn) * TN_TM, j0 = (tile / tn) * TN_TN;
  const int64_t kbeg = split * rows_per_split;
  const int64_t kend = min(M, kbeg + rows_per_split);
  PlaneLoader<TN_TM, 256> al(dYh, dYm, Mpad, i0, N);   // zero padded to Mpad: no tail handling needed
  TransLoader<TN_TN, 256> bl(X, ldx, j0, K);
  f32x16 acc[2][2];
  mainloop<2, 2>(al, bl, lds, kbeg, kend, acc);
  float *out = slab + split * N * K;
  foreach_acc<2, 2>(acc, [&](int row, int col, float v) {
    int64_t i = i0 + row, j = j0 + col;
    if (i < N && j < K) out[i * K + j] = v;
  });
}

// ---- dW with the X operand staged ROW-MAJOR and read back with gfx950's transposing LDS read.
// The B operand of the MFMA wants, per lane, 8 consecutive rows m of one X column.  linear_tn_x3_kernel gets that by
// transposing on the way IN (16 dword loads per thread and k-step, lane <-> column).  Here X rows go in as they lie
// in memory -- float4 loads, a wave covers two full 512-byte rows per instruction, 4 loads per thread and k-step --
// into a [32 m][128 columns] bf16 image per plane (256-byte rows, 16-byte chunks XOR-swizzled), and
// ds_read_b64_tr_b16 hands every lane its column's 4 consecutive rows (two reads = one 8-k fragment).
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
constexpr int TR_XPLANE = 32 * 64;  // dwords per X plane: 32 rows x 256 B

// byte offset of 16-byte chunk ch (0..15) of row r in the swizzled [32][256 B] image
__device__ __forceinline__ int tr_off(int r, int ch) { return 256 * r + 16 * (ch ^ (((r & 3) << 2) | ((r >> 2) & 3))); }

__device__ __forceinline__ bf16x8 tr_frag(const uint32_t *plane, int byte0, int byte1) {
  typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
  const char *b = reinterpret_cast<const char *>(plane);
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(b + byte0));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(b + byte1));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// Both operands row-major through the transposing read: dY needs no pre-pass either (ATR); the bias gradient is the
// column sum of the dY rows the K-tile-0 workgroups load anyway.
// STAGES = 2: two LDS stages, ONE barrier per k-step -- the split + LDS stores of step k+1 sit in the same barrier
// interval as the MFMAs of step k (the forward kernel's pipeline); STAGES = 1: store after the MFMAs, two barriers.
// Measured at 400 000 x 256 x 2592 (round 2): 2.57 vs 2.59 ms -- and two register sets on top (loads two MFMA phases
// ahead, 246 VGPRs) 2.57 vs 2.51, three sets spill (10.8 ms).  Neither the barrier count nor the load latency binds
// this kernel; per k-step a wave issues ~110 VALU instructions for the bf16 splits and 16 ds_write_b64 against 24 MFMAs
// (768 cycles), and the LDS pipe of a CU is ~80 % busy relative to its matrix pipes.  SVR_TN_STAGES=2 selects it.
// F16 ("f16x3s", svr_linear_bwd_weight_f16x3): the scaled f16 split instead of the bf16 one -- dY is multiplied by 2^s (amax_dy:
// its |max| brought to [2^13, 2^14), exact), both operands split into hi = rn16(v) and lo' = rn16((v - hi) 2^11), products
// hi hi + (hi 2^-11) lo' + lo' (hi 2^-11) with the 2^-11 applied to the hi FRAGMENTS in registers (v_pk_mul_f16), result * 2^-s:
// 22 mantissa bits per operand instead of 16, the same three matrix instructions per block.  X (activations) is not scaled:
// |x| < 65504, and below |x| ~ 0.1 the correction term carries an absolute error floor of 3e-8 |dY| (gemm_f16x3.hip).
// CONV (UNet weight gradients as an implicit GEMM, conv2d_igemm.hip): X is not a matrix in memory -- row m is output pixel
// (b, oy, ox), column j = (tap, c) of the reduction order, and X[m][j] = act(in)[b][oy s - p + ky][ox s - p + kx][c] is
// gathered from the channels-last input (zero outside the image); a thread's columns, hence its tap and channel, are fixed
// for the whole kernel, only the pixel moves.  dY may have any column count (Cout = 1 in the last decoder layer).
struct CwGeom {
  CvSrc S;
  int Wo, HoWo;
  float rWo, rHoWo;
  int k, stride, pad, Cpad;
};
template <int STAGES, bool F16 = false, bool CONV = false, bool VEC4 = true>
__global__ __launch_bounds__(256, 2) void linear_tn_x3_tr_kernel(const float *__restrict__ dY, int64_t lddy,
                                                              const float *__restrict__ X, int64_t ldx,
                                                              float *__restrict__ slab, float *__restrict__ dbpart,
                                                              int64_t M, int64_t N, int64_t K, int64_t rows_per_split,
                                                              int splits, const uint32_t *__restrict__ amax_dy, const CwGeom G) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[STAGES * 4 * TR_XPLANE];  // per stage: dY hi, dY mid, X hi, X mid: [32 m][128 cols]
  uint32_t *la = lds, *lx = lds + 2 * TR_XPLANE;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wr = wave >> 1, wc = wave & 1, lh = lane >> 5;
  const int64_t tk = cdiv(K, TN_TN), tiles = tk * cdiv(N, TN_TM), lidx = xcd_logical(blockIdx.x, gridDim.x);
  if (lidx >= tiles * splits) return;
  const int64_t split = lidx / tiles, tile = lidx % tiles;
  const int64_t tn = cdiv(N, TN_TM);
  const int64_t i0 = (tile % tn) * TN_TM, j0 = (tile / tn) * TN_TN;
  const int64_t kbeg = split * rows_per_split;
  const int64_t kend = min(M, kbeg + rows_per_split);
  // loaders: float4 idx = t + 256 i -> (row = idx >> 5, columns 4 (idx & 31) .. +3); columns past the extent are
  // clamped (they only feed outputs that are never stored), rows past the range are clamped and zeroed on the way in
  float4 av[4], xv[4];
  const int64_t acol = CONV ? i0 + 4 * (int64_t)(t & 31) : min(i0 + 4 * (int64_t)(t & 31), N - 4), xcol = min(j0 + 4 * (int64_t)(t & 31), K - 4);
  int ctap_y = 0, ctap_x = 0, cch = 0;
  if constexpr (CONV) {
    const int tap = (int)(xcol / G.Cpad);
    cch = (int)(xcol % G.Cpad);
    ctap_y = tap / G.k - G.pad;
    ctap_x = tap % G.k - G.pad;
  }
  auto gload = [&](int64_t k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t m = min(k0 + (t >> 5) + 8 * i, kend - 1);
      if constexpr (CONV) {
        if (VEC4 && (N & 3) == 0 && acol + 4 <= N) {
          av[i] = *reinterpret_cast<const float4 *>(dY + m * lddy + acol);
        } else {
          float e[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) e[u] = acol + u < N ? dY[m * lddy + acol + u] : 0.f;
          av[i] = make_float4(e[0], e[1], e[2], e[3]);
        }
        int b, rem, oy, ox;
        cv_divmod((int)m, G.HoWo, G.rHoWo, b, rem);
        cv_divmod(rem, G.Wo, G.rWo, oy, ox);
        xv[i] = cv_load4<VEC4>(G.S, b * G.S.H * G.S.W, oy * G.stride + ctap_y, ox * G.stride + ctap_x, cch);
      } else {
        av[i] = *reinterpret_cast<const float4 *>(dY + m * lddy + acol);
        xv[i] = *reinterpret_cast<const float4 *>(X + m * ldx + xcol);
      }
    }
  };
  const bool want_db = dbpart != nullptr && j0 == 0;
  float4 dbs = make_float4(0.f, 0.f, 0.f, 0.f);
  float sy = 1.f;
  if constexpr (F16) sy = amax_dy ? w_scale(amax_dy[0], false) : 1.f;
  auto lstore = [&](int64_t k0, int stage) {
    uint32_t *la = lds + stage * 4 * TR_XPLANE, *lx = la + 2 * TR_XPLANE;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = (t >> 5) + 8 * i;
      const bool ok = k0 + r < kend;
      const int off = (tr_off(r, (t & 31) >> 1) + 8 * (t & 1)) >> 2;  // dwords
      const float4 a = ok ? av[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      float4 x = ok ? xv[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      if constexpr (CONV) x = cv_act4(x, G.S.act);
      if (want_db) { dbs.x += a.x; dbs.y += a.y; dbs.z += a.z; dbs.w += a.w; }
      uint32_t h0, m0, h1, m1;
      if constexpr (F16) {
        split_x(a.x * sy, a.y * sy, h0, m0);
        split_x(a.z * sy, a.w * sy, h1, m1);
      } else {
        split2(a.x, a.y, h0, m0);
        split2(a.z, a.w, h1, m1);
      }
      *reinterpret_cast<uint2 *>(la + off) = make_uint2(h0, h1);
      *reinterpret_cast<uint2 *>(la + TR_XPLANE + off) = make_uint2(m0, m1);
      if constexpr (F16) {
        split_x(x.x, x.y, h0, m0);
        split_x(x.z, x.w, h1, m1);
      } else {
        split2(x.x, x.y, h0, m0);
        split2(x.z, x.w, h1, m1);
      }
      *reinterpret_cast<uint2 *>(lx + off) = make_uint2(h0, h1);
      *reinterpret_cast<uint2 *>(lx + TR_XPLANE + off) = make_uint2(m0, m1);
    }
  };
  // transposed-read addresses of this lane: group g = 16-lane group inside the half, lane 4q+p supplies row q, chunk p>>1
  const int g = (lane >> 4) & 1, q = (lane & 15) >> 2, p = lane & 3;
  int aoff[2][2][2], boff[2][2][2];  // [ks][tile][read] byte offsets into a plane
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        const int row = ks * 16 + lh * 8 + rr * 4 + q;
        aoff[ks][j][rr] = tr_off(row, wr * 8 + j * 4 + 2 * g + (p >> 1)) + 8 * (p & 1);
        boff[ks][j][rr] = tr_off(row, wc * 8 + j * 4 + 2 * g + (p >> 1)) + 8 * (p & 1);
      }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // CONV: output-channel counts of 32 / 64 fill a quarter / half of the tile's 128 rows -- the waves (and the 32-row blocks) past the
  // last valid row skip their fragment reads and matrix instructions (uniform per wave)
  const int nrows_valid = CONV ? (int)min<int64_t>(N - i0, (int64_t)TN_TM) - wr * 64 : 64;
  auto mma = [&](int stage) {
    const uint32_t *la = lds + stage * 4 * TR_XPLANE, *lx = la + 2 * TR_XPLANE;
    if constexpr (CONV) {
      if (nrows_valid <= 0) return;
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 ah[2], am[2], bh[2], bm[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        ah[j] = tr_frag(la, aoff[ks][j][0], aoff[ks][j][1]);
        am[j] = tr_frag(la + TR_XPLANE, aoff[ks][j][0], aoff[ks][j][1]);
        bh[j] = tr_frag(lx, boff[ks][j][0], boff[ks][j][1]);
        bm[j] = tr_frag(lx + TR_XPLANE, boff[ks][j][0], boff[ks][j][1]);
      }
      if constexpr (F16) {   // the same 16-bit lanes read as halves
        f16x8 fah[2], fam[2], fbh[2], fbm[2], fahs[2], fbhs[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          fah[j] = __builtin_bit_cast(f16x8, ah[j]);
          fam[j] = __builtin_bit_cast(f16x8, am[j]);
          fbh[j] = __builtin_bit_cast(f16x8, bh[j]);
          fbm[j] = __builtin_bit_cast(f16x8, bm[j]);
          fahs[j] = scale_2m11(fah[j]);
          fbhs[j] = scale_2m11(fbh[j]);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          if constexpr (CONV) {
            if (i * 32 >= nrows_valid) continue;
          }
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fam[i], fbhs[j], acc[i][j], 0, 0, 0);   // lo'(dY) hi(x) 2^-11
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fahs[i], fbm[j], acc[i][j], 0, 0, 0);   // hi(dY) 2^-11 lo'(x)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[i], fbh[j], acc[i][j], 0, 0, 0);
          }
        }
      } else {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bm[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
      }
    }
  };
  if (kbeg < kend) {
    gload(kbeg);
    lstore(kbeg, 0);
    __syncthreads();
    if constexpr (STAGES == 2) {
      if (kbeg + XK < kend) gload(kbeg + XK);   // registers hold step k+1 while step k is multiplied
      int st = 0;
      for (int64_t k0 = kbeg; k0 < kend; k0 += XK, st ^= 1) {
        const bool more = k0 + XK < kend;
        if (more) lstore(k0 + XK, st ^ 1);       // stage st^1 was last read in step k-1: free since the barrier
        if (k0 + 2 * XK < kend) gload(k0 + 2 * XK);
        mma(st);
        __syncthreads();
      }
    } else {
      for (int64_t k0 = kbeg; k0 < kend; k0 += XK) {
        const bool more = k0 + XK < kend;
        if (more) gload(k0 + XK);
        __builtin_amdgcn_sched_barrier(0);
        mma(0);
        __syncthreads();  // every wave has read its fragments
        if (more) lstore(k0 + XK, 0);
        __syncthreads();
      }
    }
  }
  float *out = slab + split * N * K;
  float inv = 1.f;
  if constexpr (F16) inv = amax_dy ? w_scale(amax_dy[0], true) : 1.f;
  foreach_acc<2, 2>(acc, [&](int row, int col, float v) {
    int64_t i = i0 + row, j = j0 + col;
    if (i < N && j < K) out[i * K + j] = F16 ? v * inv : v;
  });
  if (want_db) {  // fixed-order sum over the 8 row groups that share a column quad
    float *red = reinterpret_cast<float *>(lds);
    red[t * 4 + 0] = dbs.x; red[t * 4 + 1] = dbs.y; red[t * 4 + 2] = dbs.z; red[t * 4 + 3] = dbs.w;
    __syncthreads();
    if (t < 128) {
      float sum = 0.f;
      for (int rg = 0; rg < 8; ++rg) sum += red[(rg * 32 + (t >> 2)) * 4 + (t & 3)];
      if (i0 + t < N) dbpart[split * N + i0 + t] = sum;
    }
  }
}

// dY[M][N] f32 -> planes hi/mid [N][Mpad] bf16 (m contiguous, zero padded) + per-workgroup column sums for db.
// One workgroup = 64 rows x all columns, 64 columns at a time through an LDS tile.
__global__ __launch_bounds__(256) void dy_planes_kernel(const float *__restrict__ dY, int64_t lddy,
                                                        uint16_t *__restrict__ hi, uint16_t *__restrict__ mid,
                                                        int64_t Mpad, int64_t M, int64_t N, float *__restrict__ dbpart) {
  __shared__ float tile[64][65];
  const int t = threadIdx.x;
  const int64_t m0 = (int64_t)blockIdx.x * 64;
  for (int64_t n0 = 0; n0 < N; n0 += 64) {
    float4 v[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {  // 16 rows x 64 columns per pass, float4 per thread (clamped, masked below)
      const int r = p * 16 + (t >> 4), c = (t & 15) * 4;
      const int64_t m = min(m0 + r, M - 1), n = min(n0 + c, N - 4);
      v[p] = *reinterpret_cast<const float4 *>(dY + m * lddy + n);
    }
    __syncthreads();  // previous column chunk consumed
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int r = p * 16 + (t >> 4), c = (t & 15) * 4;
      const bool ok = m0 + r < M && n0 + c < N;
      tile[r][c] = ok ? v[p].x : 0.f; tile[r][c + 1] = ok ? v[p].y : 0.f;
      tile[r][c + 2] = ok ? v[p].z : 0.f; tile[r][c + 3] = ok ? v[p].w : 0.f;
    }
    __syncthreads();
    const int mq = t & 3, nl = t >> 2;  // 4 threads x 16 rows = 128 contiguous bytes of one plane row
    uint32_t h[8], md[8];
    float csum = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float a = tile[mq * 16 + 2 * i][nl], b = tile[mq * 16 + 2 * i + 1][nl];
      csum += a + b;
      split2(a, b, h[i], md[i]);
    }
    if (n0 + nl < N) {
      uint16_t *ph = hi + (n0 + nl) * Mpad + m0 + mq * 16, *pm = mid + (n0 + nl) * Mpad + m0 + mq * 16;
      *reinterpret_cast<uint4 *>(ph) = make_uint4(h[0], h[1], h[2], h[3]);
      *reinterpret_cast<uint4 *>(ph + 8) = make_uint4(h[4], h[5], h[6], h[7]);
      *reinterpret_cast<uint4 *>(pm) = make_uint4(md[0], md[1], md[2], md[3]);
      *reinterpret_cast<uint4 *>(pm + 8) = make_uint4(md[4], md[5], md[6], md[7]);
      if (dbpart) {
        csum += __shfl_xor(csum, 1);
        csum += __shfl_xor(csum, 2);
        if (mq == 0) dbpart[(int64_t)blockIdx.x * N + n0 + nl] = csum;
      }
    }
  }
}

// db[n] = f64 tree over the row-tile partials [parts][N]
__global__ __launch_bounds__(256) void dy_db_reduce_kernel(const float *__restrict__ dbpart, float *__restrict__ db, int64_t N,
                                                           int64_t parts) {
  __shared__ double red[256];
  const int64_t n = blockIdx.x;
  double s = 0.0;
  for (int64_t p = threadIdx.x; p < parts; p += 256) s += (double)dbpart[p * N + n];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) db[n] = (float)red[0];
}

// out = sum over the split-K slabs, in a FIXED order (bit-reproducible): a workgroup owns 64 outputs, its four 64-lane groups
// each walk every fourth slab with eight loads in flight, and the four partial sums are combined in lane-group order.  (One
// thread per output walking all 110-384 slabs four at a time was a latency chain: 46-110 us per call on the main stream.)
__global__ __launch_bounds__(256) void slab_reduce_x3_kernel(const float *__restrict__ slab, float *__restrict__ out, int64_t rows,
                                                             int64_t cols, int64_t ldo, int splits) {
  const int64_t st = rows * cols;
  const int64_t idx = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
  const int pl = threadIdx.x >> 6;
  const int64_t ic = idx < st ? idx : st - 1;      // (clamped: every lane loads, only valid ones store)
  float s = 0.f;
  int z = pl;
  for (; z + 28 < splits; z += 32) {               // slabs z, z + 4, ..., z + 28
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = slab[(int64_t)(z + 4 * u) * st + ic];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; z < splits; z += 4) s += slab[(int64_t)z * st + ic];
  __shared__ float red[256];
  red[threadIdx.x] = s;
  __syncthreads();
  if (pl == 0 && idx < st) {
    const int t = threadIdx.x;
    out[(idx / cols) * ldo + (idx % cols)] = ((red[t] + red[t + 64]) + red[t + 128]) + red[t + 192];
  }
}

int tn_splits(int64_t M, int64_t N, int64_t K, int64_t *rows_per_split) {
  int64_t tiles = cdiv(N, TN_TM) * cdiv(K, TN_TN);
  int64_t want = cdiv(1536, tiles);  // three 4-wave workgroups per CU, two rounds
  if (want < 1) want = 1;
  int64_t rps = cdiv(cdiv(M, want), XK) * XK;  // multiple of the k-step
  if (rps < XK) rps = XK;
  *rows_per_split = rps;
  return (int)cdiv(M, rps);
}

}  // namespace

extern "C" int64_t svr_linear_bwd_data_bf16x3_workspace(int64_t N, int64_t K) { return 2 * N * K * (int64_t)sizeof(uint16_t) + 256; }

extern "C" int svr_linear_bwd_data_bf16x3(const float *dY, int64_t lddy, const float *W, int64_t ldw, float *dX, int64_t lddx,
                                          int64_t M, int64_t N, int64_t K, int epilogue, const float *mask, int64_t ldmask,
                                          void *workspace, void *stream) {
  // dY == NULL: PREPARE only (W -> split planes in the workspace);  W == NULL: RUN on a workspace prepared earlier
  if (M == 0 && dY) return SVR_OK;
  SVR_CHECK((dY || W) && (!dY || dX) && workspace, SVR_E_BADARG, "linear_bwd_data_bf16x3: null pointer");
  SVR_CHECK(M >= 0 && N > 0 && K > 0 && N % XK == 0 && K % 4 == 0, SVR_E_BADSHAPE,
            "linear_bwd_data_bf16x3: M=%ld N=%ld K=%ld (need N %% 32 == 0, K %% 4 == 0)", (long)M, (long)N, (long)K);
  hipStream_t s = (hipStream_t)stream;
  uint16_t *hi = (uint16_t *)(((uintptr_t)workspace + 15) & ~(uintptr_t)15);
  uint16_t *mid = hi + N * K;
  if (W) hipLaunchKernelGGL(pack_planes_kernel, dim3((unsigned)cdiv(K * (N / 2), 256)), dim3(256), 0, s, W, ldw, hi, mid, N, K);
  if (!dY) return launch_status("linear_bwd_data_bf16x3 (prepare)");
  SVR_CHECK(lddx % 4 == 0 && ldmask % 4 == 0, SVR_E_BADSHAPE, "linear_bwd_data_bf16x3: leading dimensions must be multiples of 4");
  SVR_CHECK(lddy % 4 == 0 && ((uintptr_t)dY & 15) == 0, SVR_E_ALIGN, "linear_bwd_data_bf16x3: dY must be 16-byte aligned");
  SVR_CHECK(epilogue == SVR_EPI_NONE || (epilogue == SVR_EPI_MASK && mask), SVR_E_BADARG, "linear_bwd_data_bf16x3: epilogue %d", epilogue);
  dim3 grid(xcd_grid(cdiv(K, NN_TN) * cdiv(M, NN_TM)));
  hipLaunchKernelGGL(linear_nn_x3_kernel<false>, grid, dim3(256), 0, s, dY, lddy, hi, mid, dX, lddx,
                     epilogue == SVR_EPI_MASK ? mask : nullptr, ldmask, M, N, K, (const uint32_t *)nullptr, (const uint32_t *)nullptr,
                     (uint32_t *)nullptr);
  return launch_status("linear_bwd_data_bf16x3");
}

// dX = dY W on the scaled f16 split through the kernel above (called by svr_linear_bwd_data_f16x3, gemm_f16x3.hip, for N % 32 == 0:
// 0.26 instead of 0.33 ms per 400 000 x 256 x 256 layer and 64-79 instead of 170 us for the projected levels' voxel GEMMs against
// the forward kernel reused with transposed planes -- a k-step of 32 halves the barrier steps).  Workspace: amax word | planes.
namespace svr {
int linear_bwd_data_f16_nn(const float *dY, int64_t lddy, const float *W, int64_t ldw, float *dX, int64_t lddx, int64_t M, int64_t N,
                           int64_t K, int epilogue, const float *mask, int64_t ldmask, const uint32_t *amax_dy, uint32_t *amax_dx,
                           void *workspace, hipStream_t s) {
  uint32_t *amax = (uint32_t *)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  uint16_t *hi = (uint16_t *)(amax + 64), *lo = hi + N * K;
  if (W) {
    (void)hipMemsetAsync(amax, 0, sizeof(uint32_t), s);
    hipLaunchKernelGGL(w_amax_kernel, dim3((unsigned)std::min<int64_t>(cdiv(N * K, 1024), 1024)), dim3(256), 0, s, W, ldw, N, K, amax);
    hipLaunchKernelGGL(pack_planes_f16_kernel, dim3((unsigned)cdiv(K * (N / 2), 256)), dim3(256), 0, s, W, ldw, amax, hi, lo, N, K);
  }
  if (!dY) return launch_status("linear_bwd_data_f16x3 (prepare)");
  dim3 grid(xcd_grid(cdiv(K, NN_TN) * cdiv(M, NN_TM)));
  hipLaunchKernelGGL(linear_nn_x3_kernel<true>, grid, dim3(256), 0, s, dY, lddy, hi, lo, dX, lddx,
                     epilogue == SVR_EPI_MASK ? mask : nullptr, ldmask, M, N, K, (const uint32_t *)amax, amax_dy, amax_dx);
  return launch_status("linear_bwd_data_f16x3");
}
}  // namespace svr

namespace {
int64_t tn_mpad(int64_t M) { return cdiv(M, 64) * 64; }
int64_t align256b(int64_t x) { return (x + 255) / 256 * 256; }
}  // namespace

extern "C" int64_t svr_linear_bwd_weight_bf16x3_workspace(int64_t M, int64_t N, int64_t K) {
  int64_t rps;
  int splits = tn_splits(M, N, K, &rps);
  // slabs | dY planes (hi, mid) | per-row-tile column sums
  const int64_t dbparts = cdiv(M, 64) > splits ? cdiv(M, 64) : splits;  // row tiles (pre-pass) or splits (tr kernel)
  return align256b((int64_t)splits * N * K * 4) + 2 * align256b(N * tn_mpad(M) * 2) + align256b(dbparts * N * 4) + 256;
}

extern "C" int svr_linear_bwd_weight_bf16x3(const float *dY, int64_t lddy, const float *X, int64_t ldx, float *dW,
                                            int64_t lddw, float *db, int64_t M, int64_t N, int64_t K, void *workspace,
                                            void *stream) {
  SVR_CHECK(dY && X && dW && workspace, SVR_E_BADARG, "linear_bwd_weight_bf16x3: null pointer");
  SVR_CHECK(M > 0 && N > 0 && K > 0 && N % 4 == 0, SVR_E_BADSHAPE, "linear_bwd_weight_bf16x3: M=%ld N=%ld K=%ld (N %% 4)", (long)M, (long)N, (long)K);
  SVR_CHECK(lddy % 4 == 0 && ((uintptr_t)dY & 15) == 0, SVR_E_ALIGN, "linear_bwd_weight_bf16x3: dY must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  int64_t rps;
  int splits = tn_splits(M, N, K, &rps);
  const int64_t Mpad = tn_mpad(M), parts = cdiv(M, 64);
  char *w = (char *)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  float *slab = (float *)w;
  w += align256b((int64_t)splits * N * K * 4);
  uint16_t *ph = (uint16_t *)w;
  w += align256b(N * Mpad * 2);
  uint16_t *pm = (uint16_t *)w;
  w += align256b(N * Mpad * 2);
  float *dbpart = (float *)w;
  dim3 grid(xcd_grid(cdiv(K, TN_TN) * cdiv(N, TN_TM) * splits));
  if (K % 4 == 0 && ldx % 4 == 0 && ((uintptr_t)X & 15) == 0 && N >= 4 && K >= 4) {
    // both operands row-major, transposing LDS reads: no pre-pass over dY; db from the K-tile-0 workgroups
    static const int tr_stages = getenv("SVR_TN_STAGES") ? atoi(getenv("SVR_TN_STAGES")) : 1;   // measurement switch
    if (tr_stages == 2)
      hipLaunchKernelGGL(linear_tn_x3_tr_kernel<2>, grid, dim3(256), 0, s, dY, lddy, X, ldx, slab, db ? dbpart : nullptr, M, N, K,
                         rps, splits, (const uint32_t *)nullptr, CwGeom{});
    else
      hipLaunchKernelGGL(linear_tn_x3_tr_kernel<1>, grid, dim3(256), 0, s, dY, lddy, X, ldx, slab, db ? dbpart : nullptr, M, N, K,
                         rps, splits, (const uint32_t *)nullptr, CwGeom{});
    hipLaunchKernelGGL(slab_reduce_x3_kernel, dim3((unsigned)cdiv(N * K, 64)), dim3(256), 0, s, slab, dW, N, K, lddw, splits);
    if (db) hipLaunchKernelGGL(dy_db_reduce_kernel, dim3((unsigned)N), dim3(256), 0, s, dbpart, db, N, (int64_t)splits);
    return launch_status("linear_bwd_weight_bf16x3");
  }
  hipLaunchKernelGGL(dy_planes_kernel, dim3((unsigned)parts), dim3(256), 0, s, dY, lddy, ph, pm, Mpad, M, N, db ? dbpart : nullptr);
  hipLaunchKernelGGL(linear_tn_x3_kernel, grid, dim3(256), 0, s, ph, pm, Mpad, X, ldx, slab, M, N, K, rps, splits);
  hipLaunchKernelGGL(slab_reduce_x3_kernel, dim3((unsigned)cdiv(N * K, 64)), dim3(256), 0, s, slab, dW, N, K, lddw, splits);
  if (db) hipLaunchKernelGGL(dy_db_reduce_kernel, dim3((unsigned)N), dim3(256), 0, s, dbpart, db, N, parts);
  return launch_status("linear_bwd_weight_bf16x3");
}

// ---- dW on the scaled f16 split ("f16x3s"): f32-level at the bf16x3 kernel's cost; same workspace layout ---------------------
extern "C" int64_t svr_linear_bwd_weight_f16x3_workspace(int64_t M, int64_t N, int64_t K) {
  return svr_linear_bwd_weight_bf16x3_workspace(M, N, K);
}

extern "C" int svr_linear_bwd_weight_f16x3(const float *dY, int64_t lddy, const float *X, int64_t ldx, float *dW, int64_t lddw,
                                           float *db, int64_t M, int64_t N, int64_t K, const uint32_t *amax_dy, void *workspace,
                                           void *stream) {
  SVR_CHECK(dY && X && dW && workspace, SVR_E_BADARG, "linear_bwd_weight_f16x3: null pointer");
  SVR_CHECK(M > 0 && N >= 4 && K >= 4 && N % 4 == 0 && K % 4 == 0, SVR_E_BADSHAPE, "linear_bwd_weight_f16x3: M=%ld N=%ld K=%ld (N, K %% 4)", (long)M, (long)N, (long)K);
  SVR_CHECK(lddy % 4 == 0 && ((uintptr_t)dY & 15) == 0 && ldx % 4 == 0 && ((uintptr_t)X & 15) == 0, SVR_E_ALIGN,
            "linear_bwd_weight_f16x3: dY and X must be 16-byte aligned with leading dimensions that are multiples of 4");
  hipStream_t s = (hipStream_t)stream;
  int64_t rps;
  int splits = tn_splits(M, N, K, &rps);
  const int64_t Mpad = tn_mpad(M);
  char *w = (char *)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  float *slab = (float *)w;
  w += align256b((int64_t)splits * N * K * 4) + 2 * align256b(N * Mpad * 2);
  float *dbpart = (float *)w;
  dim3 grid(xcd_grid(cdiv(K, TN_TN) * cdiv(N, TN_TM) * splits));
  hipLaunchKernelGGL((linear_tn_x3_tr_kernel<1, true>), grid, dim3(256), 0, s, dY, lddy, X, ldx, slab, db ? dbpart : nullptr, M, N, K,
                     rps, splits, amax_dy, CwGeom{});
  hipLaunchKernelGGL(slab_reduce_x3_kernel, dim3((unsigned)cdiv(N * K, 64)), dim3(256), 0, s, slab, dW, N, K, lddw, splits);
  if (db) hipLaunchKernelGGL(dy_db_reduce_kernel, dim3((unsigned)N), dim3(256), 0, s, dbpart, db, N, (int64_t)splits);
  return launch_status("linear_bwd_weight_f16x3");
}

// ---- UNet weight gradients as an implicit GEMM (conv2d_igemm.hip has the forward / backward-data kernels) -----------------
namespace {
// dW (Cout, C, k, k) = sum over the row slabs, written in the parameter's own layout (slab columns are (tap, c padded to 16)).
// 64 outputs per workgroup; its four 64-lane groups each walk every fourth slab and are combined in group order (fixed order;
// one thread per output walking up to 190 slabs was 60 us of latency for the first layer's 1 536 weights).
__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(const float *__restrict__ slab, float *__restrict__ dW, int Cout, int C,
                                                                int kk, int Cpad, int splits) {
  __shared__ float red[256];
  const int64_t total = (int64_t)Cout * C * kk, idx = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
  const int g = threadIdx.x >> 6;
  float s = 0.f;
  if (idx < total) {
    const int tap = (int)(idx % kk), c = (int)((idx / kk) % C), co = (int)(idx / ((int64_t)kk * C));
    const int64_t K = (int64_t)kk * Cpad, st = (int64_t)Cout * K, at = (int64_t)co * K + (int64_t)tap * Cpad + c;
    for (int z = g; z < splits; z += 4) s += slab[z * st + at];
  }
  red[threadIdx.x] = s;
  __syncthreads();
  if (g == 0 && idx < total) dW[idx] = ((red[threadIdx.x] + red[threadIdx.x + 64]) + red[threadIdx.x + 128]) + red[threadIdx.x + 192];
}
int conv_wgrad_splits(int64_t M, int64_t N, int64_t K, int64_t *rows_per_split) {
  const int64_t tiles = cdiv(N, TN_TM) * cdiv(K, TN_TN);
  const int64_t want = std::max<int64_t>(1, 384 / tiles);         // the slabs are written and read once each: few, long splits
  int64_t rps = cdiv(cdiv(M, want), XK) * XK;
  rps = std::max<int64_t>(rps, 4 * XK);
  *rows_per_split = rps;
  return (int)cdiv(M, rps);
}
}  // namespace

extern "C" int64_t svr_conv2d_bwd_weight_workspace(const svr_conv2d_desc *d, int32_t Cout) {
  if (!d) return 0;
  const int C = d->C0 + d->C1, Cpad = (C + 15) / 16 * 16;
  const int Ho = (d->H + 2 - d->k) / d->stride + 1, Wo = (d->W + 2 - d->k) / d->stride + 1;
  int64_t rps;
  const int64_t K = (int64_t)d->k * d->k * Cpad;
  const int splits = conv_wgrad_splits((int64_t)d->B * Ho * Wo, Cout, K, &rps);
  return align256b((int64_t)splits * Cout * K * 4) + align256b((int64_t)splits * Cout * 4) + 256;
}

extern "C" int svr_conv2d_bwd_weight(const svr_conv2d_desc *d, const float *dY, const uint32_t *amax_dy, int32_t Cout, float *dW,
                                     float *db, void *workspace, void *stream) {
  SVR_CHECK(d && d->src0 && dY && dW && workspace && Cout > 0, SVR_E_BADARG, "conv2d_bwd_weight: null pointer");
  SVR_CHECK((d->k == 4 && d->stride == 2) || (d->k == 3 && d->stride == 1), SVR_E_UNSUPPORTED, "conv2d_bwd_weight: k=%d stride=%d", d->k, d->stride);
  SVR_CHECK(!d->upsample, SVR_E_UNSUPPORTED, "conv2d_bwd_weight: the x2 upsample is materialised first (svr_conv2d_virtual)");
  SVR_CHECK(d->C1 == 0 || d->src1, SVR_E_BADARG, "conv2d_bwd_weight: C1 = %d without a second source", d->C1);
  const int C = d->C0 + d->C1, Cpad = (C + 15) / 16 * 16;
  const int Ho = (d->H + 2 - d->k) / d->stride + 1, Wo = (d->W + 2 - d->k) / d->stride + 1;
  const int64_t M = (int64_t)d->B * Ho * Wo, K = (int64_t)d->k * d->k * Cpad, N = Cout;
  SVR_CHECK(M > 0 && M < (1 << 24) && (int64_t)d->B * d->H * d->W < (1 << 24), SVR_E_UNSUPPORTED, "conv2d_bwd_weight: %ld output pixels (row decode limit 2^24)", (long)M);
  hipStream_t s = (hipStream_t)stream;
  int64_t rps;
  const int splits = conv_wgrad_splits(M, N, K, &rps);
  char *w = (char *)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  float *slab = (float *)w;
  float *dbpart = (float *)(w + align256b((int64_t)splits * N * K * 4));
  CwGeom G{CvSrc{d->src0, d->src1, d->C0, d->C1, d->H, d->W, d->act}, Wo, Ho * Wo, 1.0f / (float)Wo, 1.0f / (float)(Ho * Wo), d->k, d->stride, 1, Cpad};
  const bool vec4 = d->C0 % 4 == 0 && d->C1 % 4 == 0 && (((uintptr_t)d->src0 | (uintptr_t)d->src1 | (uintptr_t)dY) & 15) == 0;
  dim3 grid(xcd_grid(cdiv(K, TN_TN) * cdiv(N, TN_TM) * splits));
  if (vec4)
    hipLaunchKernelGGL((linear_tn_x3_tr_kernel<1, true, true, true>), grid, dim3(256), 0, s, dY, (int64_t)Cout, (const float *)nullptr, (int64_t)0,
                       slab, db ? dbpart : nullptr, M, N, K, rps, splits, amax_dy, G);
  else
    hipLaunchKernelGGL((linear_tn_x3_tr_kernel<1, true, true, false>), grid, dim3(256), 0, s, dY, (int64_t)Cout, (const float *)nullptr, (int64_t)0,
                       slab, db ? dbpart : nullptr, M, N, K, rps, splits, amax_dy, G);
  const int64_t total = (int64_t)Cout * C * d->k * d->k;
  hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3((unsigned)cdiv(total, 64)), dim3(256), 0, s, (const float *)slab, dW, Cout, C,
                     d->k * d->k, Cpad, splits);
  if (db) hipLaunchKernelGGL(dy_db_reduce_kernel, dim3((unsigned)N), dim3(256), 0, s, dbpart, db, N, (int64_t)splits);
  return launch_status("conv2d_bwd_weight");
}
