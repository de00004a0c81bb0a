// Point-MLP GEMMs on the exact-f32 matrix cores (gfx950).
//
// Replaces nn.Conv1d(.,.,1) fc_0 / fc_1 / fc_2 / fc_out + ReLU (reference model/ifnet.py:19-21,
// 35,55-59) and their autograd: Y = relu(X W^T + b), dX = (dY W) * (X>0), dW = dY^T X, db = sum dY,
// plus the BCE-with-logits loss of trainer/trainer_ifnet.py:46.
#include "common.h"
#include "gemm_core.h"

using namespace svr;

namespace {

typedef TileCfg<2, 2, 2, 2> Cfg128;  // 128 x 128 block tile, 64 x 64 per wave

struct EpiArgs {
  float *C;
  int64_t ldc;
  const float *bias;
  const float *mask;
  int64_t ldmask;
  int mode;
};

template <class Cfg>
__device__ __forceinline__ void epilogue_store(f32x16 (&acc)[Cfg::TM][Cfg::TN], const EpiArgs &e, int64_t m0,
                                               int64_t n0, int64_t M, int64_t N) {
  gemm_foreach<Cfg>(acc, [&](int row, int col, float v) {
    int64_t m = m0 + row, n = n0 + col;
    if (m < M && n < N) {
      if (e.mode == SVR_EPI_BIAS || e.mode == SVR_EPI_BIAS_RELU) v += e.bias[n];
      if (e.mode == SVR_EPI_BIAS_RELU) v = fmaxf(v, 0.f);
      if (e.mode == SVR_EPI_MASK) v = (e.mask[m * e.ldmask + n] > 0.f) ? v : 0.f;
      e.C[m * e.ldc + n] = v;
    }
  });
}

// Y[M,N] = X[M,K] W[N,K]^T      (both operands k-contiguous)
__global__ __launch_bounds__(256) void linear_nt_kernel(const float *__restrict__ X, int64_t ldx,
                                                        const float *__restrict__ W, int64_t ldw, EpiArgs e,
                                                        int64_t M, int64_t N, int64_t K) {
  using Cfg = Cfg128;
  const int64_t n0 = (int64_t)blockIdx.x * Cfg::BN, m0 = (int64_t)blockIdx.y * Cfg::BM;
  auto afn = [=](int row, int kt) -> const float * {
    int64_t m = m0 + row;
    return m < M ? X + m * ldx + (int64_t)kt * BK : nullptr;
  };
  auto bfn = [=](int row, int kt) -> const float * {
    int64_t n = n0 + row;
    return n < N ? W + n * ldw + (int64_t)kt * BK : nullptr;
  };
  typedef RowKLoader<Cfg::BM, decltype(afn)> AL;
  typedef RowKLoader<Cfg::BN, decltype(bfn)> BL;
  __shared__ GemmSmem<Cfg, AL, BL> sm;
  AL al{afn};
  BL bl{bfn};
  f32x16 acc[Cfg::TM][Cfg::TN];
  gemm_mainloop<Cfg>(al, bl, sm, 0, (int)(K / BK), acc);
  epilogue_store<Cfg>(acc, e, m0, n0, M, N);
}

// dX[M,K] = dY[M,N] W[N,K]       (A k-contiguous, B = W is [k=n][col=k])
__global__ __launch_bounds__(256) void linear_nn_kernel(const float *__restrict__ dY, int64_t lddy,
                                                        const float *__restrict__ W, int64_t ldw, EpiArgs e,
                                                        int64_t M, int64_t N, int64_t K) {
  using Cfg = Cfg128;
  const int64_t c0 = (int64_t)blockIdx.x * Cfg::BN, m0 = (int64_t)blockIdx.y * Cfg::BM;
  auto afn = [=](int row, int kt) -> const float * {
    int64_t m = m0 + row;
    return m < M ? dY + m * lddy + (int64_t)kt * BK : nullptr;
  };
  auto bfn = [=](int k, int col4, int kt) -> const float * {
    int64_t c = c0 + col4;
    return c < K ? W + ((int64_t)kt * BK + k) * ldw + c : nullptr;
  };
  typedef RowKLoader<Cfg::BM, decltype(afn)> AL;
  typedef KRowLoader<Cfg::BN, decltype(bfn)> BL;
  __shared__ GemmSmem<Cfg, AL, BL> sm;
  AL al{afn};
  BL bl{bfn};
  f32x16 acc[Cfg::TM][Cfg::TN];
  gemm_mainloop<Cfg>(al, bl, sm, 0, (int)(N / BK), acc);
  epilogue_store<Cfg>(acc, e, m0, c0, M, K);
}

// slab[z][N,K] = dY[rows of split z]^T X[rows of split z]
__global__ __launch_bounds__(256) void linear_tn_kernel(const float *__restrict__ dY, int64_t lddy,
                                                        const float *__restrict__ X, int64_t ldx,
                                                        float *__restrict__ slab, int64_t M, int64_t N, int64_t K,
                                                        int steps_per_split) {
  using Cfg = Cfg128;
  const int64_t j0 = (int64_t)blockIdx.x * Cfg::BN, i0 = (int64_t)blockIdx.y * Cfg::BM;
  const int z = blockIdx.z;
  auto afn = [=](int k, int col4, int kt) -> const float * {
    int64_t m = (int64_t)kt * BK + k, i = i0 + col4;
    return (m < M && i < N) ? dY + m * lddy + i : nullptr;
  };
  auto bfn = [=](int k, int col4, int kt) -> const float * {
    int64_t m = (int64_t)kt * BK + k, j = j0 + col4;
    return (m < M && j < K) ? X + m * ldx + j : nullptr;
  };
  typedef KRowLoader<Cfg::BM, decltype(afn)> AL;
  typedef KRowLoader<Cfg::BN, decltype(bfn)> BL;
  __shared__ GemmSmem<Cfg, AL, BL> sm;
  AL al{afn};
  BL bl{bfn};
  f32x16 acc[Cfg::TM][Cfg::TN];
  const int total_steps = (int)((M + BK - 1) / BK);
  int kt0 = z * steps_per_split, kt1 = min(total_steps, kt0 + steps_per_split);
  gemm_mainloop<Cfg>(al, bl, sm, kt0, kt1, acc);
  float *out = slab + (int64_t)z * N * K;
  gemm_foreach<Cfg>(acc, [&](int row, int col, float v) {
    int64_t i = i0 + row, j = j0 + col;
    if (i < N && j < K) out[i * K + j] = v;
  });
}

// out[i] = sum_z slab[z][i]   (fixed order: bitwise reproducible); out has row stride ldo
__global__ void slab_reduce_kernel(const float *__restrict__ slab, float *__restrict__ out, int64_t rows,
                                   int64_t cols, int64_t ldo, int splits) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * cols) return;
  float s = 0.f;
  for (int z = 0; z < splits; ++z) s += slab[(int64_t)z * rows * cols + idx];
  out[(idx / cols) * ldo + (idx % cols)] = s;
}

// Column sums: part[chunk][n] = sum over the chunk's rows of Y[m][n].
// thread -> (row lane, column quad): float4 loads, a wave reads whole rows; LDS tree at the end.
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float *__restrict__ Y, int64_t ldy,
                                                             float *__restrict__ part, int64_t M, int64_t N,
                                                             int64_t rows_per_chunk) {
  const int Q = (int)(N / 4);          // host guarantees Q | 256
  const int q = threadIdx.x % Q, rl = threadIdx.x / Q, RL = 256 / Q;
  int64_t r0 = (int64_t)blockIdx.x * rows_per_chunk, r1 = min(M, r0 + rows_per_chunk);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t m = r0 + rl; m < r1; m += RL) {
    float4 v = *reinterpret_cast<const float4 *>(Y + m * ldy + q * 4);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  __shared__ float4 red[256];
  red[threadIdx.x] = s;
  __syncthreads();
  if ((int)threadIdx.x < Q) {
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < RL; ++k) {
      float4 v = red[k * Q + q];
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    *reinterpret_cast<float4 *>(part + (int64_t)blockIdx.x * N + q * 4) = a;
  }
}

// generic fallback (any N): one thread per column
__global__ void colsum_partial_slow_kernel(const float *__restrict__ Y, int64_t ldy, float *__restrict__ part, int64_t M,
                                           int64_t N, int64_t rows_per_chunk) {
  int64_t n = (int64_t)blockIdx.y * blockDim.x + threadIdx.x;
  if (n >= N) return;
  int64_t r0 = (int64_t)blockIdx.x * rows_per_chunk, r1 = min(M, r0 + rows_per_chunk);
  float s = 0.f;
  for (int64_t m = r0; m < r1; ++m) s += Y[m * ldy + n];
  part[(int64_t)blockIdx.x * N + n] = s;
}

// out[n] = sum_chunks part[chunk][n]: one workgroup per column, f64 tree (fixed order: reproducible)
__global__ __launch_bounds__(256) void colsum_final_kernel(const float *__restrict__ part, float *__restrict__ out,
                                                           int64_t N, int chunks) {
  const int64_t n = blockIdx.x;
  double s = 0.0;
  for (int c = threadIdx.x; c < chunks; c += 256) s += (double)part[(int64_t)c * N + n];
  __shared__ double red[256];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[n] = (float)red[0];
}

// fc_out forward: 16 lanes per row, float4 loads.
__global__ __launch_bounds__(256) void fc_out_fwd_kernel(const float *__restrict__ H, int64_t ldh,
                                                         const float *__restrict__ w, const float *__restrict__ b,
                                                         float *__restrict__ logits,
                                                         const int32_t *__restrict__ row_map, int64_t M, int64_t K) {
  int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  int sub = threadIdx.x & 15;
  float s = 0.f;
  if (row < M) {
    const float *h = H + row * ldh;
    for (int64_t k = sub * 4; k < K; k += 64) {
      float4 a = *reinterpret_cast<const float4 *>(h + k);
      float4 c = *reinterpret_cast<const float4 *>(w + k);
      s += a.x * c.x + a.y * c.y + a.z * c.z + a.w * c.w;
    }
  }
#pragma unroll
  for (int o = 8; o >= 1; o >>= 1) s += __shfl_xor(s, o);
  if (row < M && sub == 0) logits[row_map ? (int64_t)row_map[row] : row] = s + b[0];
}

// fc_out backward: dH = dlogit*w*(H>0); per-block partial of dw[k] = sum_m dlogit[m]*H[m][k].
__global__ __launch_bounds__(256) void fc_out_bwd_kernel(const float *__restrict__ H, int64_t ldh,
                                                         const float *__restrict__ w,
                                                         const float *__restrict__ dlogits,
                                                         const int32_t *__restrict__ row_map, float *__restrict__ dH,
                                                         int64_t lddh, float *__restrict__ part, int64_t M, int64_t K,
                                                         int64_t rows_per_block, uint32_t *__restrict__ amax_dh) {
  float vmax = 0.f;
  // thread t owns columns t, t+256, ... ; rows looped (coalesced across the wave per row)
  int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
  for (int64_t k = threadIdx.x; k < K; k += blockDim.x) {
    float wk = w[k], s = 0.f;
    for (int64_t m = r0; m < r1; ++m) {
      float h = H[m * ldh + k], g = dlogits[row_map ? (int64_t)row_map[m] : m];
      s += g * h;
      const float d = h > 0.f ? g * wk : 0.f;
      dH[m * lddh + k] = d;
      vmax = fmaxf(vmax, fabsf(d));
    }
    part[(int64_t)blockIdx.x * (K + 1) + k] = s;
  }
  if (amax_dh) {  // (uniform)
svr_amax_publish(amax_dh, vmax);
  }
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int64_t m = r0; m < r1; ++m) s += dlogits[row_map ? (int64_t)row_map[m] : m];
    part[(int64_t)blockIdx.x * (K + 1) + K] = s;
  }
}

// K == 256 form: a workgroup row = 64 threads x float4, four rows per pass, four passes in flight; the rows'
// dlogits are staged in LDS first (one dependent row_map -> dlogits lookup per row instead of per thread and row).
__global__ __launch_bounds__(256) void fc_out_bwd_k256_kernel(const float *__restrict__ H, int64_t ldh,
                                                              const float *__restrict__ w,
                                                              const float *__restrict__ dlogits,
                                                              const int32_t *__restrict__ row_map, float *__restrict__ dH,
                                                              int64_t lddh, float *__restrict__ part, int64_t M,
                                                              int64_t rows_per_block, uint32_t *__restrict__ amax_dh) {
  float vmax = 0.f;
  __shared__ float gs[512];
  __shared__ float red[4][257];
  const int t = threadIdx.x, c4 = (t & 63) * 4, rs = t >> 6;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int nr = (int)min(rows_per_block, M - r0);
  for (int i = t; i < nr; i += 256) gs[i] = dlogits[row_map ? (int64_t)row_map[r0 + i] : r0 + i];
  __syncthreads();
  const float4 wk = *reinterpret_cast<const float4 *>(w + c4);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int i0 = rs; i0 < nr; i0 += 16) {
    float4 h[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = min(i0 + 4 * u, nr - 1);
      h[u] = *reinterpret_cast<const float4 *>(H + (r0 + i) * ldh + c4);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + 4 * u;
      if (i < nr) {
        const float g = gs[i];
        s.x += g * h[u].x; s.y += g * h[u].y; s.z += g * h[u].z; s.w += g * h[u].w;
        const float4 d = make_float4(h[u].x > 0.f ? g * wk.x : 0.f, h[u].y > 0.f ? g * wk.y : 0.f, h[u].z > 0.f ? g * wk.z : 0.f,
                                     h[u].w > 0.f ? g * wk.w : 0.f);
        *reinterpret_cast<float4 *>(dH + (r0 + i) * lddh + c4) = d;
        vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(d.x), fabsf(d.y))), fmaxf(fabsf(d.z), fabsf(d.w)));
      }
    }
  }
  if (amax_dh) {  // (uniform) |max| of the stored dH: the scale of the next layer's "f16x3s" products
svr_amax_publish(amax_dh, vmax);
  }
  red[rs][c4] = s.x; red[rs][c4 + 1] = s.y; red[rs][c4 + 2] = s.z; red[rs][c4 + 3] = s.w;
  __syncthreads();
  part[(int64_t)blockIdx.x * 257 + t] = ((red[0][t] + red[1][t]) + red[2][t]) + red[3][t];
  if (t == 0) {
    float sb = 0.f;
    for (int i = 0; i < nr; ++i) sb += gs[i];
    part[(int64_t)blockIdx.x * 257 + 256] = sb;
  }
}

__global__ __launch_bounds__(256) void fc_out_bwd_final_kernel(const float *__restrict__ part, float *__restrict__ dw,
                                                               float *__restrict__ db, int64_t K, int blocks) {
  const int64_t k = blockIdx.x;  // 0..K (K = the bias slot)
  double s = 0.0;
  for (int b = threadIdx.x; b < blocks; b += 256) s += (double)part[(int64_t)b * (K + 1) + k];
  __shared__ double red[256];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (k < K) dw[k] = (float)red[0];
    else if (db) db[0] = (float)red[0];
  }
}

// BCE with logits: per-sample sums (double) then mean over the batch.
__global__ __launch_bounds__(256) void bce_kernel(const float *__restrict__ z, const float *__restrict__ y,
                                                  float *__restrict__ dz, double *__restrict__ persample, int64_t N,
                                                  float gscale_over_B) {
  int b = blockIdx.y;
  double s = 0.0;
  for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (int64_t)gridDim.x * blockDim.x) {
    float x = z[b * N + n], t = y[b * N + n];
    // max(x,0) - x*t + log1p(exp(-|x|))   (ATen binary_cross_entropy_with_logits)
    float l = fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
    s += (double)l;
    if (dz) dz[b * N + n] = (1.f / (1.f + expf(-x)) - t) * gscale_over_B;
  }
  __shared__ double red[256];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) atomicAdd(persample + b, red[0]);
}

__global__ void bce_final_kernel(const double *__restrict__ persample, float *__restrict__ loss, int64_t B) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double s = 0.0;
    for (int64_t b = 0; b < B; ++b) s += persample[b];
    loss[0] = (float)(s / (double)B);
  }
}

int check_epi(int epilogue, const float *bias, const float *mask) {
  SVR_CHECK(epilogue >= SVR_EPI_NONE && epilogue <= SVR_EPI_MASK, SVR_E_BADARG, "bad epilogue %d", epilogue);
  SVR_CHECK(!(epilogue == SVR_EPI_BIAS || epilogue == SVR_EPI_BIAS_RELU) || bias, SVR_E_BADARG, "epilogue needs bias");
  SVR_CHECK(epilogue != SVR_EPI_MASK || mask, SVR_E_BADARG, "epilogue needs mask");
  return SVR_OK;
}

int tn_splits(int64_t M, int64_t N, int64_t K, int *steps_per_split) {
  int64_t tiles = cdiv(N, 128) * cdiv(K, 128);
  int64_t steps = cdiv(M, BK);
  int64_t want = cdiv(2048, tiles);  // ~2048 workgroups
  int64_t splits = want < 1 ? 1 : want;
  if (splits > steps) splits = steps > 0 ? steps : 1;
  int64_t sps = cdiv(steps, splits);
  splits = steps > 0 ? cdiv(steps, sps) : 1;
  *steps_per_split = (int)sps;
  return (int)splits;
}


}  // namespace

namespace svr {
static int64_t colsum_rows_per_chunk(int64_t M) {
  int64_t r = cdiv(M > 0 ? M : 1, 2048);
  return r < 256 ? 256 : r;
}
int64_t colsum_workspace_floats(int64_t M, int64_t N) { return cdiv(M > 0 ? M : 1, colsum_rows_per_chunk(M)) * N; }
// out[n] = sum_m Y[m][n]; part: colsum_workspace_floats(M, N) floats
void colsum_launch(const float *Y, int64_t ldy, float *out, float *part, int64_t M, int64_t N, hipStream_t s) {
  int64_t rpc = colsum_rows_per_chunk(M);
  int chunks = (int)cdiv(M, rpc);
  bool vec = N % 4 == 0 && N / 4 <= 256 && 256 % (N / 4) == 0 && ldy % 4 == 0 && ((uintptr_t)Y & 15) == 0;
  if (vec)
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((unsigned)chunks), dim3(256), 0, s, Y, ldy, part, M, N, rpc);
  else
    hipLaunchKernelGGL(colsum_partial_slow_kernel, dim3((unsigned)chunks, (unsigned)cdiv(N, 64)), dim3(64), 0, s, Y, ldy, part, M, N, rpc);
  hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)N), dim3(256), 0, s, part, out, N, chunks);
}
}  // namespace svr

extern "C" int svr_linear_fwd(const float *X, int64_t ldx, const float *W, int64_t ldw, const float *bias, float *Y,
                              int64_t ldy, int64_t M, int64_t N, int64_t K, int epilogue, const float *mask,
                              int64_t ldmask, void *stream) {
  if (M == 0) return SVR_OK;  // empty point set
  SVR_CHECK(X && W && Y, SVR_E_BADARG, "linear_fwd: null pointer");
  SVR_CHECK(M >= 0 && N > 0 && K > 0 && K % BK == 0, SVR_E_BADSHAPE, "linear_fwd: M=%ld N=%ld K=%ld (K %% 16)", (long)M, (long)N, (long)K);
  SVR_CHECK(ldx % 4 == 0 && ldw % 4 == 0 && (((uintptr_t)X | (uintptr_t)W) & 15) == 0, SVR_E_ALIGN, "linear_fwd: operands must be 16-byte aligned");
  if (int rc = check_epi(epilogue, bias, mask)) return rc;
  if (M == 0) return SVR_OK;
  EpiArgs e{Y, ldy, bias, mask, ldmask, epilogue};
  dim3 grid((unsigned)cdiv(N, Cfg128::BN), (unsigned)cdiv(M, Cfg128::BM));
  hipLaunchKernelGGL(linear_nt_kernel, grid, dim3(256), 0, (hipStream_t)stream, X, ldx, W, ldw, e, M, N, K);
  return launch_status("linear_fwd");
}

extern "C" int svr_linear_bwd_data(const float *dY, int64_t lddy, const float *W, int64_t ldw, float *dX, int64_t lddx,
                                   int64_t M, int64_t N, int64_t K, int epilogue, const float *mask, int64_t ldmask,
                                   void *stream) {
  if (M == 0) return SVR_OK;
  SVR_CHECK(dY && W && dX, SVR_E_BADARG, "linear_bwd_data: null pointer");
  SVR_CHECK(M >= 0 && N > 0 && K > 0 && N % BK == 0 && K % 4 == 0, SVR_E_BADSHAPE, "linear_bwd_data: M=%ld N=%ld K=%ld", (long)M, (long)N, (long)K);
  SVR_CHECK(lddy % 4 == 0 && ldw % 4 == 0 && (((uintptr_t)dY | (uintptr_t)W) & 15) == 0, SVR_E_ALIGN, "linear_bwd_data: operands must be 16-byte aligned");
  SVR_CHECK(epilogue == SVR_EPI_NONE || epilogue == SVR_EPI_MASK, SVR_E_BADARG, "linear_bwd_data: epilogue %d", epilogue);
  if (int rc = check_epi(epilogue, nullptr, mask)) return rc;
  if (M == 0) return SVR_OK;
  EpiArgs e{dX, lddx, nullptr, mask, ldmask, epilogue};
  dim3 grid((unsigned)cdiv(K, Cfg128::BN), (unsigned)cdiv(M, Cfg128::BM));
  hipLaunchKernelGGL(linear_nn_kernel, grid, dim3(256), 0, (hipStream_t)stream, dY, lddy, W, ldw, e, M, N, K);
  return launch_status("linear_bwd_data");
}

extern "C" int64_t svr_linear_bwd_weight_workspace(int64_t M, int64_t N, int64_t K) {
  int sps;
  int splits = tn_splits(M, N, K, &sps);
  return ((int64_t)splits * N * K + colsum_workspace_floats(M, N)) * (int64_t)sizeof(float);
}

extern "C" int svr_linear_bwd_weight(const float *dY, int64_t lddy, const float *X, int64_t ldx, float *dW, int64_t lddw,
                                     float *db, int64_t M, int64_t N, int64_t K, void *workspace, void *stream) {
  SVR_CHECK(dY && X && dW && workspace, SVR_E_BADARG, "linear_bwd_weight: null pointer");
  SVR_CHECK(M > 0 && N > 0 && K > 0 && N % 4 == 0 && K % 4 == 0, SVR_E_BADSHAPE, "linear_bwd_weight: M=%ld N=%ld K=%ld", (long)M, (long)N, (long)K);
  SVR_CHECK(lddy % 4 == 0 && ldx % 4 == 0 && (((uintptr_t)dY | (uintptr_t)X) & 15) == 0, SVR_E_ALIGN, "linear_bwd_weight: operands must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  int sps;
  int splits = tn_splits(M, N, K, &sps);
  float *slab = (float *)workspace;
  dim3 grid((unsigned)cdiv(K, Cfg128::BN), (unsigned)cdiv(N, Cfg128::BM), (unsigned)splits);
  hipLaunchKernelGGL(linear_tn_kernel, grid, dim3(256), 0, s, dY, lddy, X, ldx, slab, M, N, K, sps);
  hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)cdiv(N * K, 256)), dim3(256), 0, s, slab, dW, N, K, lddw, splits);
  if (db) colsum_launch(dY, lddy, db, slab + (int64_t)splits * N * K, M, N, s);
  return launch_status("linear_bwd_weight");
}

extern "C" int svr_fc_out_fwd(const float *H, int64_t ldh, const float *w, const float *b, float *logits,
                              const int32_t *row_map, int64_t M, int64_t K, void *stream) {
  if (M <= 0) return SVR_OK;
  SVR_CHECK(H && w && b && logits, SVR_E_BADARG, "fc_out_fwd: null pointer");
  SVR_CHECK(K > 0 && K % 4 == 0 && ldh % 4 == 0, SVR_E_BADSHAPE, "fc_out_fwd: K=%ld ldh=%ld", (long)K, (long)ldh);
  if (M <= 0) return SVR_OK;
  hipLaunchKernelGGL(fc_out_fwd_kernel, dim3((unsigned)cdiv(M * 16, 256)), dim3(256), 0, (hipStream_t)stream, H, ldh, w, b, logits, row_map, M, K);
  return launch_status("fc_out_fwd");
}

namespace { constexpr int64_t FCO_ROWS = 512; }

extern "C" int64_t svr_fc_out_bwd_workspace(int64_t M, int64_t K) { return cdiv(M > 0 ? M : 1, FCO_ROWS) * (K + 1) * (int64_t)sizeof(float); }

extern "C" int svr_fc_out_bwd(const float *H, int64_t ldh, const float *w, const float *dlogits, const int32_t *row_map,
                              float *dH, int64_t lddh, float *dw, float *db, int64_t M, int64_t K, uint32_t *amax_dh,
                              void *workspace, void *stream) {
  SVR_CHECK(H && w && dlogits && dH && dw && workspace, SVR_E_BADARG, "fc_out_bwd: null pointer");
  SVR_CHECK(M > 0 && K > 0, SVR_E_BADSHAPE, "fc_out_bwd: M=%ld K=%ld", (long)M, (long)K);
  hipStream_t s = (hipStream_t)stream;
  int blocks = (int)cdiv(M, FCO_ROWS);
  float *part = (float *)workspace;
  if (K == 256 && ldh % 4 == 0 && lddh % 4 == 0 && ((((uintptr_t)H | (uintptr_t)dH | (uintptr_t)w)) & 15) == 0)
    hipLaunchKernelGGL(fc_out_bwd_k256_kernel, dim3((unsigned)blocks), dim3(256), 0, s, H, ldh, w, dlogits, row_map, dH, lddh, part, M, FCO_ROWS, amax_dh);
  else
    hipLaunchKernelGGL(fc_out_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, s, H, ldh, w, dlogits, row_map, dH, lddh, part, M, K, FCO_ROWS, amax_dh);
  hipLaunchKernelGGL(fc_out_bwd_final_kernel, dim3((unsigned)(K + 1)), dim3(256), 0, s, part, dw, db, K, blocks);
  return launch_status("fc_out_bwd");
}

extern "C" int svr_bce_logits_sum_mean(const float *logits, const float *targets, float *loss, float *dlogits, int64_t B,
                                       int64_t N, float gscale, void *workspace, void *stream) {
  SVR_CHECK(logits && targets && loss && workspace, SVR_E_BADARG, "bce: null pointer");
  SVR_CHECK(B > 0 && N > 0, SVR_E_BADSHAPE, "bce: B=%ld N=%ld", (long)B, (long)N);
  hipStream_t s = (hipStream_t)stream;
  hipError_t err = hipMemsetAsync(workspace, 0, (size_t)B * sizeof(double), s);
  SVR_CHECK(err == hipSuccess, (int)err, "bce: memset failed: %s", hipGetErrorString(err));
  int64_t gx64 = cdiv(N, 256); unsigned gx = (unsigned)(gx64 < 64 ? gx64 : 64);
  hipLaunchKernelGGL(bce_kernel, dim3(gx, (unsigned)B), dim3(256), 0, s, logits, targets, dlogits, (double *)workspace, N, gscale / (float)B);
  hipLaunchKernelGGL(bce_final_kernel, dim3(1), dim3(64), 0, s, (const double *)workspace, loss, B);
  return launch_status("bce");
}
