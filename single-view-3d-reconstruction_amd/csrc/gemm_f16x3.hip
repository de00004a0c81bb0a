// Forward GEMMs of the point MLP at f32 accuracy on the f16 matrix cores ("f16x3"), gfx950.
//
// An f32 value splits into two f16 terms that carry 22 mantissa bits:  x = hi + lo,  hi = rn16(x),
// lo = rn16(x - hi),  |x - hi - lo| <= 2^-22 |x|.  With f32 accumulation
//   x*w ~= hi(x) hi(w) + hi(x) lo(w) + lo(x) hi(w)                       (the dropped lo*lo is <= 2^-22 |x w|)
// is as accurate as a plain f32 GEMM (measured: 3e-7 of the f64 result, the same as an f32 FMA chain and within
// the gates the exact-f32 MFMA kernel is held to) at 3 f16 MFMAs (3 x 32 cycles per 32x32x16 block) instead of
// the 6 of the bf16 split (gemm_bf16x6.hip) or 8 x 64 cycles of f32 MFMAs.
//
// f16 has 5 exponent bits, so the lo terms would be subnormal (and lose bits) for |x| < 0.125.  Both are kept
// in the normal range by exact power-of-two scaling, without a second accumulator:
//   * W is normalised once per call: Ws = W * 2^s with amax(Ws) in [2^13, 2^14); planes wh = rn16(Ws),
//     wl = rn16(Ws - wh) and wq = wh * 2^-11 (exact; normal for every weight above 7.6e-6 of the largest);
//   * X is split on the fly into xh = rn16(x) and xl' = rn16((x - xh) * 2^11)   (normal whenever xh is);
//   * acc += xh wh + xh wl + xl' wq   (xl' wq == xl wh exactly), and the epilogue multiplies by 2^-s.
// Domain: |x| < 65504 (f16 range; beyond that the result is inf/nan, not silently wrong).  |x| < 6e-5 keeps an
// absolute error of 1.5e-11.  "bf16x6" (any f32 range) and "f32" stay selectable.
//
//   Y[M,N] = epi( X[M,K] W[N,K]^T )    128x128 tile, 4 waves x (2x2) v_mfma_f32_32x32x16_f16 tiles, k-step 16.
// The kernel is bound by the bytes it pulls through L2 / Infinity Cache (X from HBM, W re-read by every
// workgroup: 16.6 GB at 400 000 x 2592 x 256 = 6.9 TB/s), not by the matrix cores: only wh and wl travel (wq is
// one v_pk_mul_f16 per fragment register) and both streams are prefetched two k-steps ahead.  A 128x256 tile
// (X read once) ran one workgroup per CU and was slower (3.1 ms vs 2.4); `nt` loads of X were slower too (3.5).
#include "common.h"
#include "f16x3.h"
#include <algorithm>

using namespace svr;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int YK = 16;   // reduction elements per step
constexpr int YLW = 8;   // dwords per LDS row: 16 halves = two 16-byte slots, no padding
constexpr int TM = 128;
constexpr int APLANE = TM * YLW;  // dwords per X plane

// A fragment (8 halves of one row) is one 16-byte slot and comes out of ONE ds_read_b128 (256 B/clk; the 8-byte aligned
// padded rows of the first version made it a ds_read2_b64: 128 B/clk).  Slot of (row, half h) = 2 row + (h ^ bit 3 of row):
// the 16-lane groups of ds_read_b128 ({0-3,12-15,20-27}, {4-11,16-19,28-31}, + 32: MI355X_MICROARCH.md "LDS") hold rows that
// pair up mod 8 at distances 8 and 24, so the 16 lanes hit 16 different slots mod 16 -- conflict free.
__device__ __forceinline__ int slot_dw(int row, int h) { return (row * 2 + (h ^ ((row >> 3) & 1))) * 4; }

__device__ __forceinline__ f16x8 read_frag(const uint32_t *plane, int row, int lh) {
  union { uint4 q; f16x8 v; } f;
  f.q = *reinterpret_cast<const uint4 *>(plane + slot_dw(row, lh));
  return f.v;
}

// W[N][K] f32 -> f16 planes hi(Ws), lo(Ws), K-STEP MAJOR: [k-step][hi / lo][n][16 halves].  A workgroup's W tile of one
// k-step is then one contiguous 4 KB block per plane (16 bytes per thread); row-major planes [N][K] gave every thread a
// 16-byte piece of a different 128-byte line, of which only 32 bytes belong to the k-step, and the three workgroups
// of a CU evicted the lines from the 32 KB L1 before the next k-step came back for them: 4x the W bytes through L2.
__device__ __forceinline__ int64_t wplane_index(int64_t k, int plane, int64_t n, int64_t N) {
  return (((k >> 4) * 2 + plane) * N + n) * 16 + (k & 15);
}
// transposed: plane row n / reduction index k are W's COLUMN / ROW (the weight operand of dX = dY W: rows = dX columns)
__global__ void split_w_kernel(const float *__restrict__ W, int64_t ldw, const uint32_t *__restrict__ amax,
                               uint16_t *__restrict__ p0, int64_t N, int64_t K, int transposed) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // over (n, k/2)
  if (idx >= N * (K / 2)) return;
  const int64_t n = idx / (K / 2), k = (idx % (K / 2)) * 2;
  const float sc = w_scale(amax[0], false);
  const float w0 = (transposed ? W[k * ldw + n] : W[n * ldw + k]) * sc, w1 = (transposed ? W[(k + 1) * ldw + n] : W[n * ldw + k + 1]) * sc;
  const uint32_t hi = pack_f16(w0, w1);
  const f32x2 h = unpack_f16(hi);
  *reinterpret_cast<uint32_t *>(p0 + wplane_index(k, 0, n, N)) = hi;
  *reinterpret_cast<uint32_t *>(p0 + wplane_index(k, 1, n, N)) = pack_f16(w0 - h.x, w1 - h.y);
}


// TN = 128: 4 waves, three workgroups per CU.  SCHED selects the instruction-scheduling hint of the k-step.
//
// Every wave runs the same k-step, so waves of different workgroups drift into lockstep and the LDS / VALU phase
// of one does NOT hide behind the MFMA phase of another (measured: MFMA, LDS+VALU and global-load times simply
// add up).  The k-step is therefore software pipelined INSIDE the wave: two LDS stages and two register sets, so
// that the split + LDS stores of step k+1 and the global loads of step k+2 sit in the same barrier-free region
// as the MFMAs of step k and can be issued between them (MFMA co-execution); one barrier per step.
// BWD: the same kernel as the data-gradient GEMM dX = dY W of the scaled f16 split ("f16x3s", svr_linear_bwd_data_f16x3):
// X = dY, the planes hold W TRANSPOSED; dY is multiplied by 2^sx (amax_x: its |max| brought to [2^13, 2^14), exact) on the
// way into the split so that gradients of any magnitude sit in f16's normal range, the epilogue multiplies by 2^-(s + sx),
// applies the ReLU mask of the layer below and (amax_y) leaves |max| of what it stored for the next layer's scale.
template <int TN, int NX, int SCHED, bool BWD = false>
__global__ __launch_bounds__(2 * TN, 1024 / (2 * TN) >= 4 ? 3 : 2) void linear_nt_h3_kernel(
    const float *__restrict__ X, int64_t ldx, const uint16_t *__restrict__ W0,
    const uint32_t *__restrict__ amax, const float *__restrict__ bias, float *__restrict__ Y, int64_t ldy, int64_t M,
    int64_t N, int64_t K, int relu, const uint32_t *__restrict__ amax_x, const float *__restrict__ mask, int64_t ldm,
    uint32_t *__restrict__ amax_y) {
  constexpr int NT = 2 * TN, BPLANE = TN * YLW, XPT = 512 / NT, STAGE = 2 * APLANE + 2 * BPLANE;
  __shared__ __attribute__((aligned(16))) uint32_t lds[2 * STAGE];  // two stages of (hi/lo planes of X and W)
  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wr = wave / (TN / 64), wc = wave % (TN / 64);   // 2 x (TN/64) waves of 64 x 64
  const int l31 = lane & 31, lh = lane >> 5;
  // both column tiles of a row panel run on one XCD, back to back: X comes from HBM once, then from that L2
  const int64_t ntn = cdiv(N, TN), lidx = xcd_logical(blockIdx.x, gridDim.x);
  if (lidx >= ntn * cdiv(M, TM)) return;
  const int64_t n0 = (lidx % ntn) * TN, m0 = (lidx / ntn) * TM;

  // loaders (rows past the extent are clamped: they only feed outputs the guarded epilogue never stores)
  const float *xp[XPT];
#pragma unroll
  for (int i = 0; i < XPT; ++i) {
    int64_t xr = m0 + (t >> 2) + (NT / 4) * i;
    xr = xr < M ? xr : M - 1;
    xp[i] = X + xr * ldx + (t & 3) * 4;             // 128 rows x 16 floats
  }
  int64_t wrow = n0 + (t >> 1);
  wrow = wrow < N ? wrow : N - 1;
  const int64_t woff = wrow * 16 + (t & 1) * 8;     // TN rows x 16 halves per plane and k-step: 16 bytes per thread, contiguous
  struct Regs {
    float4 x[XPT];
    uint4 w[2];
  };
  Regs ra, rb;
  const int64_t klast = K - YK;
  float sx = 1.f;
  if constexpr (BWD) sx = amax_x ? w_scale(amax_x[0], false) : 1.f;
  auto load = [&](Regs &r, int64_t k0) {
    k0 = k0 < klast ? k0 : klast;  // past the end: re-read the last step (never used), keeps the loop branch free
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
      if constexpr (NX) {
        const float *q = xp[i] + k0;
        r.x[i] = make_float4(__builtin_nontemporal_load(q), __builtin_nontemporal_load(q + 1),
                             __builtin_nontemporal_load(q + 2), __builtin_nontemporal_load(q + 3));
      } else {
        r.x[i] = *reinterpret_cast<const float4 *>(xp[i] + k0);
      }
    }
    const uint16_t *wk = W0 + (k0 >> 4) * (2 * N * 16) + woff;  // k-step major planes (split_w_kernel)
    r.w[0] = *reinterpret_cast<const uint4 *>(wk);
    r.w[1] = *reinterpret_cast<const uint4 *>(wk + N * 16);
  };
  auto store = [&](const Regs &r, uint32_t *st) {
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
      uint32_t h0, l0, h1, l1;
      if constexpr (BWD) {
        split_x(r.x[i].x * sx, r.x[i].y * sx, h0, l0);
        split_x(r.x[i].z * sx, r.x[i].w * sx, h1, l1);
      } else {
        split_x(r.x[i].x, r.x[i].y, h0, l0);
        split_x(r.x[i].z, r.x[i].w, h1, l1);
      }
      const int off = slot_dw((t >> 2) + (NT / 4) * i, (t & 3) >> 1) + (t & 1) * 2;   // 4 halves = half a slot
      *reinterpret_cast<uint2 *>(st + off) = make_uint2(h0, h1);
      *reinterpret_cast<uint2 *>(st + APLANE + off) = make_uint2(l0, l1);
    }
    uint32_t *sb = st + 2 * APLANE;
    const int offb = slot_dw(t >> 1, t & 1);
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

  // step k0 on LDS stage `cur`: fragments -> MFMAs; `nxt` (step k0+16, in flight since the previous step) is split
  // and stored to the other stage; `fre` (stored one step ago) is refilled with step k0+32
  auto step = [&](int64_t k0, int cur, Regs &nxt, Regs &fre) {
    load(fre, k0 + 2 * YK);
    const uint32_t *pa = lds + cur * STAGE, *pb = pa + 2 * APLANE;
    f16x8 a[2][2], b[3][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      a[0][i] = read_frag(pa, wr * 64 + i * 32 + l31, lh);
      a[1][i] = read_frag(pa + APLANE, wr * 64 + i * 32 + l31, lh);
      b[0][i] = read_frag(pb, wc * 64 + i * 32 + l31, lh);
      b[1][i] = read_frag(pb + BPLANE, wc * 64 + i * 32 + l31, lh);
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
    if constexpr (SCHED == 1) {
      __builtin_amdgcn_iglp_opt(0);
    } else if constexpr (SCHED == 2) {
      // after the fragment reads: one MFMA, then a slice of the split (VALU) and of the LDS stores, twelve times
#pragma unroll
      for (int g = 0; g < 12; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);  // VALU
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // DS write
      }
    }
    __syncthreads();
  };

  load(ra, 0);
  load(rb, YK);
  store(ra, lds);
  __syncthreads();
  for (int64_t k0 = 0; k0 < K; k0 += 2 * YK) {
    step(k0, 0, rb, ra);
    if (k0 + YK < K) step(k0 + YK, 1, ra, rb);
  }

  float inv = w_scale(amax[0], true);
  if constexpr (BWD) inv *= amax_x ? w_scale(amax_x[0], true) : 1.f;
  float vmax = 0.f;
  // Epilogue.  TN = 128 and float4-able output (round 4): the 128 x 128 tile goes through LDS in two passes of 64 rows (the
  // operand stages are free behind the loop's last barrier: 32 KB = 64 x 128 floats, rows unpadded -- the accumulator
  // stores of a wave are 32 consecutive dwords per row, the float4 reads 512 contiguous bytes per row: conflict free), so
  // that a thread issues 16 float4 stores (and mask loads) on full 512-byte rows instead of 64 dword ones on 128-byte
  // segments; bias / ReLU / mask / scale / |max| are applied on the way out.
  const bool coal = TN == 128 && N % 4 == 0 && ldy % 4 == 0 && (((uintptr_t)Y) & 15) == 0 && (!BWD || !mask || (ldm % 4 == 0 && (((uintptr_t)mask) & 15) == 0));
  if (coal) {
    float *ct = reinterpret_cast<float *>(lds);
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      float4 k4[8];
      if constexpr (BWD) {   // this pass's ReLU mask, all loads up front from clamped coordinates
        if (mask) {
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int64_t m = m0 + 64 * pass + (t >> 5) + 8 * i, c = n0 + (t & 31) * 4;
            k4[i] = *reinterpret_cast<const float4 *>(mask + (m < M ? m : M - 1) * ldm + (c < N ? c : N - 4));
          }
        }
      }
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
      if (bias && c < N) b4 = *reinterpret_cast<const float4 *>(bias + c);   // N % 4 == 0: never straddles the edge
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = (t >> 5) + 8 * i;
        const int64_t m = m0 + 64 * pass + row;
        if (m < M && c < N) {
          float4 v = *reinterpret_cast<const float4 *>(ct + row * 128 + c4);
          v.x = v.x * inv + b4.x; v.y = v.y * inv + b4.y; v.z = v.z * inv + b4.z; v.w = v.w * inv + b4.w;
          if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
          if constexpr (BWD) {
            if (mask) {
              v.x = k4[i].x > 0.f ? v.x : 0.f; v.y = k4[i].y > 0.f ? v.y : 0.f;
              v.z = k4[i].z > 0.f ? v.z : 0.f; v.w = k4[i].w > 0.f ? v.w : 0.f;
            }
            vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
          }
          *reinterpret_cast<float4 *>(Y + m * ldy + c) = v;
        }
      }
      __syncthreads();
    }
  } else {
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int64_t n = n0 + wc * 64 + j * 32 + l31;
      const float bv = (bias && n < N) ? bias[n] : 0.f;
      float mk[16];
      if constexpr (BWD) {
        // the ReLU mask of the tile: all 16 loads issued together, unconditionally and from clamped coordinates (a load inside
        // the bounds branch below is waited for on its own: 64 serial memory round trips per thread, +0.4 ms per layer)
        if (mask) {
          const int64_t nc = n < N ? n : N - 1;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int64_t m = m0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            mk[r] = mask[(m < M ? m : M - 1) * ldm + nc];
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) mk[r] = 1.f;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t m = m0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        float v = acc[i][j][r] * inv + bv;
        if (relu) v = fmaxf(v, 0.f);
        if constexpr (BWD) v = mk[r] > 0.f ? v : 0.f;
        if (m < M && n < N) {
          if constexpr (BWD) vmax = fmaxf(vmax, fabsf(v));
          Y[m * ldy + n] = v;
        }
      }
    }
  }
  if constexpr (BWD) {
    if (amax_y) {   // |max| of this workgroup's part of dX (non-negative floats order like unsigned integers)
svr_amax_publish(amax_y, vmax);
    }
  }
}

// |max| of an (M, N) f32 matrix (leading dimension ld, N % 4 == 0, 16-byte aligned rows) -> bit pattern in amax[0] (zeroed by
// the host first): the scale of a gradient operand of the scaled f16 split
__global__ __launch_bounds__(256) void amax_rows_kernel(const float *__restrict__ X, int64_t ld, int64_t M, int64_t N,
                                                        uint32_t *__restrict__ amax) {
  const int64_t q = N / 4, total = M * q;
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const float4 v = *reinterpret_cast<const float4 *>(X + (i / q) * ld + (i % q) * 4);
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  svr_amax_publish(amax, m);
}

}  // namespace

extern "C" int64_t svr_linear_fwd_f16x3_workspace(int64_t N, int64_t K) { return 2 * N * K * (int64_t)sizeof(uint16_t) + 512; }

extern "C" int svr_linear_fwd_f16x3(const float *X, int64_t ldx, const float *W, int64_t ldw, const float *bias, float *Y,
                                    int64_t ldy, int64_t M, int64_t N, int64_t K, int epilogue, void *workspace,
                                    void *stream) {
  // X == NULL: PREPARE only (W -> scale + split planes in the workspace);  W == NULL: RUN on a workspace prepared earlier
  if (M == 0 && X) return SVR_OK;  // empty point set
  SVR_CHECK((X || W) && (!X || Y) && workspace, SVR_E_BADARG, "linear_fwd_f16x3: null pointer");
  SVR_CHECK(M >= 0 && N > 0 && K > 0 && K % YK == 0, SVR_E_BADSHAPE, "linear_fwd_f16x3: M=%ld N=%ld K=%ld (K %% 16)", (long)M, (long)N, (long)K);
  hipStream_t s = (hipStream_t)stream;
  uint32_t *amax = (uint32_t *)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  uint16_t *p0 = (uint16_t *)(amax + 64);
  if (W) {
    (void)hipMemsetAsync(amax, 0, sizeof(uint32_t), s);
    hipLaunchKernelGGL(w_amax_kernel, dim3((unsigned)std::min<int64_t>(cdiv(N * K, 1024), 1024)), dim3(256), 0, s, W, ldw, N, K, amax);
    hipLaunchKernelGGL(split_w_kernel, dim3((unsigned)cdiv(N * (K / 2), 256)), dim3(256), 0, s, W, ldw, amax, p0, N, K, 0);
  }
  if (!X) return launch_status("linear_fwd_f16x3 (prepare)");
  SVR_CHECK(ldx % 4 == 0 && ((uintptr_t)X & 15) == 0, SVR_E_ALIGN, "linear_fwd_f16x3: X must be 16-byte aligned");
  SVR_CHECK(epilogue == SVR_EPI_NONE || ((epilogue == SVR_EPI_BIAS || epilogue == SVR_EPI_BIAS_RELU) && bias), SVR_E_BADARG,
            "linear_fwd_f16x3: epilogue %d", epilogue);
  const float *eb = epilogue == SVR_EPI_NONE ? nullptr : bias;
  const int relu = epilogue == SVR_EPI_BIAS_RELU ? 1 : 0;
  // SCHED = 2 (one MFMA, a slice of the split, one LDS store, ...): 2.39 ms at 400 000 x 2592 x 256 against 2.49 for
  // the compiler's own order and 2.50 for iglp_opt(0)
  if (N <= 64) {  // narrow outputs (the UNet's 32- and 64-channel layers): 128 x 64 tiles, two waves, half the wasted columns
    dim3 grid(xcd_grid(cdiv(N, 64) * cdiv(M, TM)));
    hipLaunchKernelGGL((linear_nt_h3_kernel<64, 0, 2>), grid, dim3(128), 0, s, X, ldx, p0, amax, eb, Y, ldy, M, N, K, relu,
                       (const uint32_t *)nullptr, (const float *)nullptr, (int64_t)0, (uint32_t *)nullptr);
    return launch_status("linear_fwd_f16x3");
  }
  dim3 grid(xcd_grid(cdiv(N, 128) * cdiv(M, TM)));
  hipLaunchKernelGGL((linear_nt_h3_kernel<128, 0, 2>), grid, dim3(256), 0, s, X, ldx, p0, amax, eb, Y, ldy, M, N, K, relu,
                     (const uint32_t *)nullptr, (const float *)nullptr, (int64_t)0, (uint32_t *)nullptr);
  return launch_status("linear_fwd_f16x3");
}

// ---- scaled f16 split for the BACKWARD products ("f16x3s"): f32-level gradients at the cost of the bf16x3 split -------------
extern "C" int svr_amax_f32(const float *X, int64_t ld, int64_t M, int64_t N, uint32_t *amax, void *stream) {
  SVR_CHECK(amax, SVR_E_BADARG, "amax_f32: null pointer");
  hipStream_t s = (hipStream_t)stream;
  if (M * N == 0) return SVR_OK;
  SVR_CHECK(X && N % 4 == 0 && ld % 4 == 0 && ((uintptr_t)X & 15) == 0, SVR_E_ALIGN, "amax_f32: N, ld multiples of 4, 16-byte aligned");
  hipLaunchKernelGGL(amax_rows_kernel, dim3((unsigned)std::min<int64_t>(cdiv(M * (N / 4), 1024), 2048)), dim3(256), 0, s, X, ld, M, N, amax);
  return launch_status("amax_f32");
}

extern "C" int64_t svr_linear_bwd_data_f16x3_workspace(int64_t N, int64_t K) { return 2 * N * K * (int64_t)sizeof(uint16_t) + 512; }

extern "C" int svr_linear_bwd_data_f16x3(const float *dY, int64_t lddy, const float *W, int64_t ldw, float *dX, int64_t lddx,
                                         int64_t M, int64_t N, int64_t K, int epilogue, const float *mask, int64_t ldmask,
                                         const uint32_t *amax_dy, uint32_t *amax_dx, void *workspace, void *stream) {
  // dY == NULL: PREPARE only (W -> scale + split TRANSPOSED planes in the workspace);  W == NULL: RUN on a prepared workspace
  if (M == 0 && dY) return SVR_OK;
  SVR_CHECK((dY || W) && (!dY || dX) && workspace, SVR_E_BADARG, "linear_bwd_data_f16x3: null pointer");
  SVR_CHECK(M >= 0 && N > 0 && K > 0 && N % YK == 0, SVR_E_BADSHAPE, "linear_bwd_data_f16x3: M=%ld N=%ld K=%ld (N %% 16)", (long)M, (long)N, (long)K);
  hipStream_t s = (hipStream_t)stream;
  // SVR_DX_KERNEL=nt: the forward kernel with transposed planes (k-step 16) instead of the k-step-32 kernel of gemm_bf16x3.hip
  // (process-wide: a workspace prepared under one setting is run under the same one)
  static const bool dx_nn = !(getenv("SVR_DX_KERNEL") && getenv("SVR_DX_KERNEL")[0] == 'n' && getenv("SVR_DX_KERNEL")[1] == 't');
  if (dx_nn && N % 32 == 0 && K % 4 == 0) {   // (a function of the SHAPE only: PREPARE and RUN calls must agree on the plane layout)
    if (dY) {
      SVR_CHECK(lddy % 4 == 0 && ((uintptr_t)dY & 15) == 0, SVR_E_ALIGN, "linear_bwd_data_f16x3: dY must be 16-byte aligned");
      SVR_CHECK(epilogue == SVR_EPI_NONE || (epilogue == SVR_EPI_MASK && mask), SVR_E_BADARG, "linear_bwd_data_f16x3: epilogue %d", epilogue);
      SVR_CHECK(lddx % 4 == 0 && (((uintptr_t)dX) & 15) == 0 && (epilogue != SVR_EPI_MASK || (ldmask % 4 == 0 && (((uintptr_t)mask) & 15) == 0)),
                SVR_E_ALIGN, "linear_bwd_data_f16x3: dX and the mask must be 16-byte aligned with leading dimensions that are multiples of 4");
    }
    return svr::linear_bwd_data_f16_nn(dY, lddy, W, ldw, dX, lddx, M, N, K, epilogue, mask, ldmask, amax_dy, amax_dx, workspace, s);
  }
  uint32_t *amax = (uint32_t *)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  uint16_t *p0 = (uint16_t *)(amax + 64);
  if (W) {   // planes of W^T: K rows (dX columns) x N reduction elements
    (void)hipMemsetAsync(amax, 0, sizeof(uint32_t), s);
    hipLaunchKernelGGL(w_amax_kernel, dim3((unsigned)std::min<int64_t>(cdiv(N * K, 1024), 1024)), dim3(256), 0, s, W, ldw, N, K, amax);
    hipLaunchKernelGGL(split_w_kernel, dim3((unsigned)cdiv(K * (N / 2), 256)), dim3(256), 0, s, W, ldw, amax, p0, K, N, 1);
  }
  if (!dY) return launch_status("linear_bwd_data_f16x3 (prepare)");
  SVR_CHECK(lddy % 4 == 0 && ((uintptr_t)dY & 15) == 0, SVR_E_ALIGN, "linear_bwd_data_f16x3: dY must be 16-byte aligned");
  SVR_CHECK(epilogue == SVR_EPI_NONE || (epilogue == SVR_EPI_MASK && mask), SVR_E_BADARG, "linear_bwd_data_f16x3: epilogue %d", epilogue);
  const float *mk = epilogue == SVR_EPI_MASK ? mask : nullptr;
  dim3 grid(xcd_grid(cdiv(K, 128) * cdiv(M, TM)));
  hipLaunchKernelGGL((linear_nt_h3_kernel<128, 0, 2, true>), grid, dim3(256), 0, s, dY, lddy, p0, amax, (const float *)nullptr, dX, lddx,
                     M, K, N, 0, amax_dy, mk, ldmask, amax_dx);
  return launch_status("linear_bwd_data_f16x3");
}
