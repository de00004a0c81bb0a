// Forward GEMMs of the point MLP at f32 accuracy on the bf16 matrix cores ("bf16x6"), gfx950: the range-safe
// alternative (any f32 magnitude) to the production f16x3 forward (gemm_f16x3.hip), at twice its matrix-core work.
//
//   x = hi + mid + lo exactly to 2^-24 |x|   (three round-to-nearest bf16 terms carry 24 mantissa bits)
//   a*b ~= a_hi b_hi + a_hi b_mid + a_mid b_hi + a_mid b_mid + a_hi b_lo + a_lo b_hi     (f32 accumulate)
//
// The dropped terms (a_mid b_lo, a_lo b_mid, a_lo b_lo) are <= 2^-24 |a b|, i.e. the same size as the rounding
// of an f32 product: results agree with the exact-f32 MFMA path to ~2e-7 relative (tests: 2e-6 vs f64, the same
// gate as the f32 kernel), so logits and ReLU masks keep f32 fidelity, at 6 bf16 MFMAs (6 x 32 cycles per
// 32x32x16 block) instead of 8 f32 MFMAs (8 x 64 cycles).
//
//   Y[M,N] = epi( X[M,K] W[N,K]^T )     X split on the fly, W pre-split once per call into 3 planes.
//
// 128x128 tile, 4 waves x (2x2) v_mfma_f32_32x32x16_bf16 tiles, k-step 16, LDS planes [row][k] with a 40-byte
// row stride (ds_read_b64 fragment reads conflict free), one 30 KB LDS stage (4 workgroups per CU).
#include "common.h"

using namespace svr;

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int YK = 16;   // reduction elements per step
constexpr int YLW = 10;  // dwords per LDS row (16 bf16 + 8 B pad)
constexpr int TM = 128, TN = 128;
constexpr int PLANE = TM * YLW;           // dwords per plane
constexpr int STAGE = 3 * (TM + TN) * YLW;  // dwords per stage

__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
  f32x2 v = {a, b};
  bf16x2 h = __builtin_convertvector(v, bf16x2);
  return __builtin_bit_cast(uint32_t, h);
}

// (x0, x1) -> packed bf16 pairs of the hi / mid / lo terms
__device__ __forceinline__ void split3(float x0, float x1, uint32_t &hi, uint32_t &mid, uint32_t &lo) {
  hi = pack_bf16(x0, x1);
  const float r0 = x0 - __uint_as_float(hi << 16), r1 = x1 - __uint_as_float(hi & 0xffff0000u);
  mid = pack_bf16(r0, r1);
  lo = pack_bf16(r0 - __uint_as_float(mid << 16), r1 - __uint_as_float(mid & 0xffff0000u));
}

__device__ __forceinline__ bf16x8 read_frag(const uint32_t *plane, int row, int lh) {
  const uint2 a = *reinterpret_cast<const uint2 *>(plane + row * YLW + lh * 4);
  const uint2 b = *reinterpret_cast<const uint2 *>(plane + row * YLW + lh * 4 + 2);
  union { uint4 q; bf16x8 v; } f;
  f.q = make_uint4(a.x, a.y, b.x, b.y);
  return f.v;
}

// W[N][K] f32 -> three bf16 planes [N][K]
__global__ void split_planes_kernel(const float *__restrict__ W, int64_t ldw, uint16_t *__restrict__ p0,
                                    uint16_t *__restrict__ p1, uint16_t *__restrict__ p2, int64_t N, int64_t K) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // over (n, k/2)
  if (idx >= N * (K / 2)) return;
  int64_t n = idx / (K / 2), k = (idx % (K / 2)) * 2;
  uint32_t h, m, l;
  split3(W[n * ldw + k], W[n * ldw + k + 1], h, m, l);
  *reinterpret_cast<uint32_t *>(p0 + n * K + k) = h;
  *reinterpret_cast<uint32_t *>(p1 + n * K + k) = m;
  *reinterpret_cast<uint32_t *>(p2 + n * K + k) = l;
}

__global__ __launch_bounds__(256, 3) void linear_nt_x6_kernel(const float *__restrict__ X, int64_t ldx,
                                                           const uint16_t *__restrict__ W0,
                                                           const uint16_t *__restrict__ W1,
                                                           const uint16_t *__restrict__ W2, const float *__restrict__ bias,
                                                           float *__restrict__ Y, int64_t ldy, int64_t M, int64_t N,
                                                           int64_t K, int relu) {
  __shared__ uint32_t lds[STAGE];  // one stage, two barriers per k-step: more workgroups (= loads in flight) per CU
  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int l31 = lane & 31, lh = lane >> 5;
  const int64_t n0 = (int64_t)blockIdx.x * TN, m0 = (int64_t)blockIdx.y * TM;

  // loaders (rows past the extent are clamped: they only feed outputs the guarded epilogue never stores)
  const float *xp[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int64_t r = m0 + (t >> 2) + 64 * i;
    r = r < M ? r : M - 1;
    xp[i] = X + r * ldx + (t & 3) * 4;
  }
  int64_t wrow = n0 + (t >> 1);
  wrow = wrow < N ? wrow : N - 1;
  const int64_t woff = wrow * K + (t & 1) * 8;
  float4 xa[2];
  uint2 wv[3][2];
  auto load = [&](int64_t k0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) xa[i] = *reinterpret_cast<const float4 *>(xp[i] + k0);
    const uint2 *q0 = reinterpret_cast<const uint2 *>(W0 + woff + k0);
    const uint2 *q1 = reinterpret_cast<const uint2 *>(W1 + woff + k0);
    const uint2 *q2 = reinterpret_cast<const uint2 *>(W2 + woff + k0);
    wv[0][0] = q0[0]; wv[0][1] = q0[1];
    wv[1][0] = q1[0]; wv[1][1] = q1[1];
    wv[2][0] = q2[0]; wv[2][1] = q2[1];
  };
  auto store = [&](uint32_t *st) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      uint32_t h0, m0_, l0, h1, m1, l1;
      split3(xa[i].x, xa[i].y, h0, m0_, l0);
      split3(xa[i].z, xa[i].w, h1, m1, l1);
      const int off = ((t >> 2) + 64 * i) * YLW + (t & 3) * 2;
      *reinterpret_cast<uint2 *>(st + off) = make_uint2(h0, h1);
      *reinterpret_cast<uint2 *>(st + PLANE + off) = make_uint2(m0_, m1);
      *reinterpret_cast<uint2 *>(st + 2 * PLANE + off) = make_uint2(l0, l1);
    }
    uint32_t *sb = st + 3 * PLANE;
    const int offb = (t >> 1) * YLW + (t & 1) * 4;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      *reinterpret_cast<uint2 *>(sb + p * PLANE + offb) = wv[p][0];
      *reinterpret_cast<uint2 *>(sb + p * PLANE + offb + 2) = wv[p][1];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  load(0);
  store(lds);
  __syncthreads();
  for (int64_t k0 = 0; k0 < K; k0 += YK) {
    const bool more = k0 + YK < K;
    if (more) load(k0 + YK);
    const uint32_t *pa = lds, *pb = pa + 3 * PLANE;
    bf16x8 a[3][2], b[3][2];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a[p][i] = read_frag(pa + p * PLANE, wr * 64 + i * 32 + l31, lh);
        b[p][i] = read_frag(pb + p * PLANE, wc * 64 + i * 32 + l31, lh);
      }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][i], b[0][j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[2][j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[1][j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[0][j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[1][j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[0][j], acc[i][j], 0, 0, 0);
      }
    __syncthreads();  // every wave has read its fragments
    if (more) store(lds);
    __syncthreads();
  }

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
          Y[m * ldy + n] = v;
        }
      }
    }
}

}  // namespace

extern "C" int64_t svr_linear_fwd_bf16x6_workspace(int64_t N, int64_t K) { return 3 * N * K * (int64_t)sizeof(uint16_t) + 256; }

extern "C" int svr_linear_fwd_bf16x6(const float *X, int64_t ldx, const float *W, int64_t ldw, const float *bias, float *Y,
                                     int64_t ldy, int64_t M, int64_t N, int64_t K, int epilogue, void *workspace,
                                     void *stream) {
  if (M == 0) return SVR_OK;  // empty point set
  SVR_CHECK(X && W && Y && workspace, SVR_E_BADARG, "linear_fwd_bf16x6: null pointer");
  SVR_CHECK(M >= 0 && N > 0 && K > 0 && K % YK == 0, SVR_E_BADSHAPE, "linear_fwd_bf16x6: M=%ld N=%ld K=%ld (K %% 16)", (long)M, (long)N, (long)K);
  SVR_CHECK(ldx % 4 == 0 && ((uintptr_t)X & 15) == 0, SVR_E_ALIGN, "linear_fwd_bf16x6: X must be 16-byte aligned");
  SVR_CHECK(epilogue == SVR_EPI_NONE || ((epilogue == SVR_EPI_BIAS || epilogue == SVR_EPI_BIAS_RELU) && bias), SVR_E_BADARG,
            "linear_fwd_bf16x6: epilogue %d", epilogue);
  if (M == 0) return SVR_OK;
  hipStream_t s = (hipStream_t)stream;
  uint16_t *p0 = (uint16_t *)(((uintptr_t)workspace + 15) & ~(uintptr_t)15);
  uint16_t *p1 = p0 + N * K, *p2 = p1 + N * K;
  hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)cdiv(N * (K / 2), 256)), dim3(256), 0, s, W, ldw, p0, p1, p2, N, K);
  dim3 grid((unsigned)cdiv(N, TN), (unsigned)cdiv(M, TM));
  hipLaunchKernelGGL(linear_nt_x6_kernel, grid, dim3(256), 0, s, X, ldx, p0, p1, p2,
                     epilogue == SVR_EPI_NONE ? nullptr : bias, Y, ldy, M, N, K, epilogue == SVR_EPI_BIAS_RELU ? 1 : 0);
  return launch_status("linear_fwd_bf16x6");
}
