// BatchNorm3d (training / eval) fused with MaxPool3d(2), channels-last, gfx950.
//
// Replaces nn.BatchNorm3d after the ReLU of every encoder stage and the nn.MaxPool3d(2)
// between stages (reference model/ifnet.py:136-142,165-192; 32-variant :76-80,102-114), and
// their autograd.  The BN output is what gets sampled AND pooled, so one pass over the
// activation writes the full-resolution normalised volume, the pooled volume and the pool
// argmax; the backward merges {gather gradient, un-pooled gradient} -> BN backward -> ReLU mask.
//
// Bandwidth bound (HBM): every element is read once / written once per pass, float4 per lane.
// Statistics are accumulated in short f32 runs and carried in f64 (torch's CPU kernel uses a
// double accumulator for float input), so mean / variance are good to ~1e-7 relative.
#include "common.h"

using namespace svr;

namespace {

constexpr int STAT_BLOCKS_MAX = 2048;

// ---------------------------------------------------------------- statistics
// thread -> (row lane r = tid / Q, quad q = tid % Q); rows strided by 256/Q.
__global__ __launch_bounds__(256) void bn_stats_partial_kernel(const float *__restrict__ x, double *__restrict__ part,
                                                               int64_t rows, int C, int64_t rows_per_block) {
  const int Q = C / 4;
  const int q = threadIdx.x % Q, rl = threadIdx.x / Q, RL = 256 / Q;
  int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  double s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
  int64_t r = r0 + rl;
  while (r < r1) {
    float fs[4] = {0, 0, 0, 0}, fss[4] = {0, 0, 0, 0};
    // f32 runs of 16 rows, as two batches of 8 loads issued together (clamped rows, masked afterwards): a load per
    // loop iteration behind the `r < r1` branch would be waited for before the next one is issued
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      float4 v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int64_t rr = r + (int64_t)i * RL;
        v[i] = *reinterpret_cast<const float4 *>(x + (rr < r1 ? rr : r1 - 1) * C + q * 4);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (r + (int64_t)i * RL < r1) {
          fs[0] += v[i].x; fs[1] += v[i].y; fs[2] += v[i].z; fs[3] += v[i].w;
          fss[0] += v[i].x * v[i].x; fss[1] += v[i].y * v[i].y; fss[2] += v[i].z * v[i].z; fss[3] += v[i].w * v[i].w;
        }
      }
      r += 8 * (int64_t)RL;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { s[i] += (double)fs[i]; ss[i] += (double)fss[i]; }
  }
  __shared__ double red[256 * 8];
#pragma unroll
  for (int i = 0; i < 4; ++i) { red[threadIdx.x * 8 + i] = s[i]; red[threadIdx.x * 8 + 4 + i] = ss[i]; }
  __syncthreads();
  if ((int)threadIdx.x < Q) {
    double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int k = 0; k < RL; ++k)
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] += red[(k * Q + q) * 8 + i];
    double *o = part + (int64_t)blockIdx.x * 2 * C;
#pragma unroll
    for (int i = 0; i < 4; ++i) { o[q * 4 + i] = a[i]; o[C + q * 4 + i] = a[4 + i]; }
  }
}

// what svr_bn_finalize computes for one channel from (mean, biased variance): running statistics (training), scale / shift /
// invstd, the f32 mean
struct BnFinalize {
  const float *gamma, *beta;
  float *rmean, *rvar, *ss, *mean_f32;   // ss == nullptr: nothing to finalize
  float eps, momentum;
};
__device__ __forceinline__ void bn_finalize_channel(const BnFinalize &f, int c, int C, double mean, double var, int64_t rows,
                                                    int training) {
  if (training) {
    if (f.rmean) f.rmean[c] = (float)((1.0 - (double)f.momentum) * (double)f.rmean[c] + (double)f.momentum * mean);
    if (f.rvar) {
      double unb = rows > 1 ? var * (double)rows / (double)(rows - 1) : var;
      f.rvar[c] = (float)((1.0 - (double)f.momentum) * (double)f.rvar[c] + (double)f.momentum * unb);
    }
  }
  double invstd = 1.0 / sqrt(var + (double)f.eps);
  float g = f.gamma ? f.gamma[c] : 1.f, b = f.beta ? f.beta[c] : 0.f;
  float scale = (float)invstd * g;
  f.ss[c] = scale;
  f.ss[C + c] = b - (float)mean * scale;
  f.ss[2 * C + c] = (float)invstd;
  f.mean_f32[c] = (float)mean;
}

// one workgroup per channel: f64 tree over the block partials (fixed order); with fin.ss the same thread also does
// svr_bn_finalize's work for its channel (training mode), one launch instead of two
__global__ __launch_bounds__(256) void bn_stats_final_kernel(const double *__restrict__ part, double *__restrict__ stats,
                                                             int64_t rows, int C, int blocks, BnFinalize fin) {
  const int c = blockIdx.x;
  double s = 0, ss = 0;
  for (int b = threadIdx.x; b < blocks; b += 256) { s += part[(int64_t)b * 2 * C + c]; ss += part[(int64_t)b * 2 * C + C + c]; }
  __shared__ double r1[256], r2[256];
  r1[threadIdx.x] = s; r2[threadIdx.x] = ss;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) { r1[threadIdx.x] += r1[threadIdx.x + o]; r2[threadIdx.x] += r2[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    double mean = r1[0] / (double)rows;
    double var = r2[0] / (double)rows - mean * mean;
    var = var > 0 ? var : 0;
    if (stats) { stats[c] = mean; stats[C + c] = var; }
    if (fin.ss) bn_finalize_channel(fin, c, C, mean, var, rows, 1);
  }
}

__global__ void bn_finalize_kernel(const double *__restrict__ stats, BnFinalize fin, int64_t rows, int C, int training) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double mean, var;
  if (training) {
    mean = stats[c];
    var = stats[C + c];
  } else {
    mean = (double)fin.rmean[c];
    var = (double)fin.rvar[c];
  }
  bn_finalize_channel(fin, c, C, mean, var, rows, training);
}

// ---------------------------------------------------------------- apply + pool
struct Vol {
  int B, D, H, W, C;
};

// one thread = one 2x2x2 cell (ceil-div grid, so odd trailing voxels are still normalised) x 4 channels
__global__ __launch_bounds__(256) void bn_apply_pool_kernel(const float *__restrict__ x, const float *__restrict__ ss,
                                                            float *__restrict__ y, float *__restrict__ pooled,
                                                            uint8_t *__restrict__ argmax, Vol v, int64_t total) {
  int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= total) return;
  const int Q = v.C / 4;
  const int Dc = (v.D + 1) / 2, Hc = (v.H + 1) / 2, Wc = (v.W + 1) / 2;
  const int Dp = v.D / 2, Hp = v.H / 2, Wp = v.W / 2;
  // 32-bit index arithmetic (64-bit divisions by run-time values cost ~100 instructions each; the kernel was spending
  // ~700 VALU instructions per thread on 8 loads).  total < 2^32 is checked on the host.
  const uint32_t g32 = (uint32_t)gid;
  int q = (int)(g32 % (uint32_t)Q);
  uint32_t cell = g32 / (uint32_t)Q;
  int cx = (int)(cell % (uint32_t)Wc); cell /= (uint32_t)Wc;
  int cy = (int)(cell % (uint32_t)Hc); cell /= (uint32_t)Hc;
  int cz = (int)(cell % (uint32_t)Dc);
  int64_t b = cell / (uint32_t)Dc;
  float4 sc = *reinterpret_cast<const float4 *>(ss + q * 4);
  float4 sh = *reinterpret_cast<const float4 *>(ss + v.C + q * 4);
  float4 best = make_float4(0.f, 0.f, 0.f, 0.f);
  int am[4] = {0, 0, 0, 0};
  bool first = true;
  // the cell's eight voxels are loaded up front, unconditionally, from clamped coordinates (loads inside the bounds
  // branch would be waited for one at a time); the scheduling barrier keeps the compiler from re-serialising them
  float4 in8[8];
  int64_t off8[8];
  {  // corner k = base + per-axis strides (0 where the cell is cut by an odd extent: the clamped load is discarded)
    const int64_t base = (((b * v.D + cz * 2) * v.H + cy * 2) * v.W + cx * 2) * v.C + q * 4;
    const int64_t sz = (cz * 2 + 1 < v.D) ? (int64_t)v.H * v.W * v.C : 0, sy = (cy * 2 + 1 < v.H) ? (int64_t)v.W * v.C : 0;
    const int64_t sx = (cx * 2 + 1 < v.W) ? v.C : 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      off8[k] = base + (k >> 2) * sz + ((k >> 1) & 1) * sy + (k & 1) * sx;
      in8[k] = *reinterpret_cast<const float4 *>(x + off8[k]);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    int z = cz * 2 + (k >> 2), yy = cy * 2 + ((k >> 1) & 1), xx = cx * 2 + (k & 1);
    if (z < v.D && yy < v.H && xx < v.W) {
      const int64_t off = off8[k];
      const float4 a = in8[k];
      float4 o;
      o.x = a.x * sc.x + sh.x; o.y = a.y * sc.y + sh.y; o.z = a.z * sc.z + sh.z; o.w = a.w * sc.w + sh.w;
      *reinterpret_cast<float4 *>(y + off) = o;
      // first maximum in (z,y,x) scan order wins, NaN propagates (ATen max_pool3d: val > max || isnan(val))
      if (first) { best = o; first = false; }
      else {
        if (o.x > best.x || o.x != o.x) { best.x = o.x; am[0] = k; }
        if (o.y > best.y || o.y != o.y) { best.y = o.y; am[1] = k; }
        if (o.z > best.z || o.z != o.z) { best.z = o.z; am[2] = k; }
        if (o.w > best.w || o.w != o.w) { best.w = o.w; am[3] = k; }
      }
    }
  }
  if (pooled && cz < Dp && cy < Hp && cx < Wp) {
    int64_t po = (((b * Dp + cz) * Hp + cy) * Wp + cx) * v.C + q * 4;
    *reinterpret_cast<float4 *>(pooled + po) = best;
    if (argmax) *reinterpret_cast<uint32_t *>(argmax + po) = (uint32_t)am[0] | ((uint32_t)am[1] << 8) | ((uint32_t)am[2] << 16) | ((uint32_t)am[3] << 24);
  }
}

// ---------------------------------------------------------------- backward
// dy_total at the 8 voxels of a cell = dy (gather gradient, optional) + dpooled routed by argmax.
template <bool APPLY>
__global__ __launch_bounds__(256) void bn_bwd_kernel(const float *__restrict__ x, const float *__restrict__ dy,
                                                     const float *__restrict__ dpooled,
                                                     const uint8_t *__restrict__ argmax,
                                                     const float *__restrict__ mean, const float *__restrict__ ss,
                                                     const double *__restrict__ sums, double *__restrict__ part,
                                                     float *__restrict__ dx, Vol v, int64_t cells, int relu_mask,
                                                     float *__restrict__ dgamma, float *__restrict__ dbeta,
                                                     uint32_t *__restrict__ amax_dx) {
  float vmax = 0.f;   // APPLY: |max| of the dx this thread stores (amax_dx: the scale of the next "f16x3s" product, no extra pass)
  if (APPLY && blockIdx.x == 0 && (int)threadIdx.x < v.C) {   // dgamma = sum dy*xhat, dbeta = sum dy (no extra launch)
    if (dbeta) dbeta[threadIdx.x] = (float)sums[threadIdx.x];
    if (dgamma) dgamma[threadIdx.x] = (float)sums[v.C + threadIdx.x];
  }
  const int Q = v.C / 4;
  const int q = threadIdx.x % Q, cl = threadIdx.x / Q, CL = 256 / Q;
  const int Dc = (v.D + 1) / 2, Hc = (v.H + 1) / 2, Wc = (v.W + 1) / 2;
  const int Dp = v.D / 2, Hp = v.H / 2, Wp = v.W / 2;
  const double n = (double)v.B * v.D * v.H * v.W;
  float4 mu = *reinterpret_cast<const float4 *>(mean + q * 4);
  float4 sc = *reinterpret_cast<const float4 *>(ss + q * 4);          // gamma*invstd
  float4 is = *reinterpret_cast<const float4 *>(ss + 2 * v.C + q * 4);  // invstd
  float m1[4] = {0, 0, 0, 0}, m2[4] = {0, 0, 0, 0};
  if (APPLY && !(relu_mask & 2)) {  // bit 1: frozen (running) statistics -> no batch-mean terms, dx = scale * dy
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      m1[i] = (float)(sums[q * 4 + i] / n);
      m2[i] = (float)(sums[v.C + q * 4 + i] / n);
    }
  }
  double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
  for (int64_t cell = (int64_t)blockIdx.x * CL + cl; cell < cells; cell += (int64_t)gridDim.x * CL) {
    uint32_t t = (uint32_t)cell;  // cells < 2^32 (host-checked): 32-bit divisions
    int cx = (int)(t % (uint32_t)Wc); t /= (uint32_t)Wc;
    int cy = (int)(t % (uint32_t)Hc); t /= (uint32_t)Hc;
    int cz = (int)(t % (uint32_t)Dc);
    int64_t b = t / (uint32_t)Dc;
    float4 dp = make_float4(0.f, 0.f, 0.f, 0.f);
    uint32_t am = 0xffffffffu;
    if (dpooled && cz < Dp && cy < Hp && cx < Wp) {
      int64_t po = (((b * Dp + cz) * Hp + cy) * Wp + cx) * v.C + q * 4;
      dp = *reinterpret_cast<const float4 *>(dpooled + po);
      am = *reinterpret_cast<const uint32_t *>(argmax + po);
    }
    float f1[4] = {0, 0, 0, 0}, f2[4] = {0, 0, 0, 0};
    // all 16 loads of the cell first (see bn_apply_pool_kernel)
    float4 a8[8], g8[8];
    int64_t off8[8];
    {  // corner k = base + per-axis strides (0 where the cell is cut by an odd extent: the clamped load is discarded)
      const int64_t base = (((b * v.D + cz * 2) * v.H + cy * 2) * v.W + cx * 2) * v.C + q * 4;
      const int64_t sz = (cz * 2 + 1 < v.D) ? (int64_t)v.H * v.W * v.C : 0, sy = (cy * 2 + 1 < v.H) ? (int64_t)v.W * v.C : 0;
      const int64_t sx = (cx * 2 + 1 < v.W) ? v.C : 0;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        off8[k] = base + (k >> 2) * sz + ((k >> 1) & 1) * sy + (k & 1) * sx;
        a8[k] = *reinterpret_cast<const float4 *>(x + off8[k]);
      }
    }
    if (dy) {  // wave-uniform
#pragma unroll
      for (int k = 0; k < 8; ++k) g8[k] = *reinterpret_cast<const float4 *>(dy + off8[k]);
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) g8[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      int z = cz * 2 + (k >> 2), yy = cy * 2 + ((k >> 1) & 1), xx = cx * 2 + (k & 1);
      if (z < v.D && yy < v.H && xx < v.W) {
        const int64_t off = off8[k];
        const float4 a = a8[k];
        float4 g = g8[k];
        if ((am & 0xff) == (uint32_t)k) g.x += dp.x;
        if (((am >> 8) & 0xff) == (uint32_t)k) g.y += dp.y;
        if (((am >> 16) & 0xff) == (uint32_t)k) g.z += dp.z;
        if (((am >> 24) & 0xff) == (uint32_t)k) g.w += dp.w;
        float xh[4] = {(a.x - mu.x) * is.x, (a.y - mu.y) * is.y, (a.z - mu.z) * is.z, (a.w - mu.w) * is.w};
        float gg[4] = {g.x, g.y, g.z, g.w};
        if (APPLY) {
          float av[4] = {a.x, a.y, a.z, a.w};
          float scv[4] = {sc.x, sc.y, sc.z, sc.w};
          float o[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            o[i] = scv[i] * (gg[i] - m1[i] - xh[i] * m2[i]);
            if ((relu_mask & 1) && !(av[i] > 0.f)) o[i] = 0.f;
          }
          *reinterpret_cast<float4 *>(dx + off) = make_float4(o[0], o[1], o[2], o[3]);
          vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) { f1[i] += gg[i]; f2[i] += gg[i] * xh[i]; }
        }
      }
    }
    if (!APPLY) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { s1[i] += (double)f1[i]; s2[i] += (double)f2[i]; }
    }
  }
  if (APPLY && amax_dx) {  // (uniform)
svr_amax_publish(amax_dx, vmax);
  }
  if (!APPLY) {
    __shared__ double red[256 * 8];
#pragma unroll
    for (int i = 0; i < 4; ++i) { red[threadIdx.x * 8 + i] = s1[i]; red[threadIdx.x * 8 + 4 + i] = s2[i]; }
    __syncthreads();
    if ((int)threadIdx.x < Q) {
      double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int k = 0; k < CL; ++k)
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] += red[(k * Q + q) * 8 + i];
      double *o = part + (int64_t)blockIdx.x * 2 * v.C;
#pragma unroll
      for (int i = 0; i < 4; ++i) { o[q * 4 + i] = a[i]; o[v.C + q * 4 + i] = a[4 + i]; }
    }
  }
}

__global__ __launch_bounds__(256) void sum_parts_kernel(const double *__restrict__ part, double *__restrict__ out, int cols,
                                                        int blocks) {
  const int c = blockIdx.x;
  double s = 0;
  for (int b = threadIdx.x; b < blocks; b += 256) s += part[(int64_t)b * cols + c];
  __shared__ double red[256];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[c] = red[0];
}

int check_c(int C) {
  SVR_CHECK(C >= 4 && C <= 256 && C % 4 == 0 && 256 % (C / 4) == 0, SVR_E_UNSUPPORTED, "bn: C=%d (need 4 | C, C/4 | 256)", C);
  return SVR_OK;
}

int stats_blocks(int64_t rows, int64_t *rows_per_block) {
  int64_t b = cdiv(rows, 256);
  if (b > STAT_BLOCKS_MAX) b = STAT_BLOCKS_MAX;
  if (b < 1) b = 1;
  *rows_per_block = cdiv(rows, b);
  return (int)cdiv(rows, *rows_per_block);
}

int bwd_blocks(int64_t cells, int C) {
  int CL = 256 / (C / 4);
  int64_t b = cdiv(cells, CL);
  if (b > STAT_BLOCKS_MAX) b = STAT_BLOCKS_MAX;
  return (int)(b < 1 ? 1 : b);
}

}  // namespace

namespace svr {
// mean / biased variance from per-block partial sums [blocks][2C] (f64): shared with conv3d.hip's fused conv_in forward
void bn_stats_final_launch(const double *part, double *stats, int64_t rows, int C, int blocks, hipStream_t s) {
  hipLaunchKernelGGL(bn_stats_final_kernel, dim3(C), dim3(256), 0, s, part, stats, rows, C, blocks, BnFinalize{});
}
// ... the same + svr_bn_finalize(training = 1) in ONE launch (stats may be null): shared with stage1.hip
void bn_stats_finalize_launch(const double *part, double *stats, int64_t rows, int C, int blocks, const float *gamma,
                              const float *beta, float *rmean, float *rvar, float *ss, float *mean_f32, float eps, float momentum,
                              hipStream_t s) {
  hipLaunchKernelGGL(bn_stats_final_kernel, dim3(C), dim3(256), 0, s, part, stats, rows, C, blocks,
                     BnFinalize{gamma, beta, rmean, rvar, ss, mean_f32, eps, momentum});
}
// out[c] = ordered f64 sum over `blocks` partial rows [blocks][cols]: shared with stage1.hip's backward reduction
void bn_sum_parts_launch(const double *part, double *out, int cols, int blocks, hipStream_t s) {
  hipLaunchKernelGGL(sum_parts_kernel, dim3(cols), dim3(256), 0, s, part, out, cols, blocks);
}
}  // namespace svr

extern "C" int64_t svr_bn_stats_workspace(int64_t rows, int32_t C) {
  (void)rows;
  return (int64_t)STAT_BLOCKS_MAX * 2 * C * (int64_t)sizeof(double);
}

extern "C" int svr_bn_stats(const float *x, double *stats, int64_t rows, int32_t C, void *workspace, void *stream) {
  if (int rc = check_c(C)) return rc;
  SVR_CHECK(x && stats && workspace && rows > 0, SVR_E_BADARG, "bn_stats: bad argument");
  hipStream_t s = (hipStream_t)stream;
  int64_t rpb;
  int blocks = stats_blocks(rows, &rpb);
  hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(blocks), dim3(256), 0, s, x, (double *)workspace, rows, C, rpb);
  hipLaunchKernelGGL(bn_stats_final_kernel, dim3(C), dim3(256), 0, s, (const double *)workspace, stats, rows, C, blocks, BnFinalize{});
  return launch_status("bn_stats");
}

// svr_bn_stats + svr_bn_finalize(training = 1) as two launches instead of three (the per-channel tree also finalizes)
extern "C" int svr_bn_stats_finalize(const float *x, double *stats, const float *gamma, const float *beta, float *running_mean,
                                     float *running_var, float *scale_shift, float *mean_f32, int64_t rows, int32_t C, float eps,
                                     float momentum, void *workspace, void *stream) {
  if (int rc = check_c(C)) return rc;
  SVR_CHECK(x && scale_shift && mean_f32 && workspace && rows > 0, SVR_E_BADARG, "bn_stats_finalize: bad argument");
  hipStream_t s = (hipStream_t)stream;
  int64_t rpb;
  int blocks = stats_blocks(rows, &rpb);
  hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(blocks), dim3(256), 0, s, x, (double *)workspace, rows, C, rpb);
  bn_stats_finalize_launch((const double *)workspace, stats, rows, C, blocks, gamma, beta, running_mean, running_var, scale_shift,
                           mean_f32, eps, momentum, s);
  return launch_status("bn_stats_finalize");
}

// Finalize from per-workgroup partial sums part[blocks][2][C] (sum, sum of squares; what svr_conv3d_k3_fwd_f16x3_stats and
// svr_conv3d_c1_fwd_stats' kernels leave): mean / biased variance (stats, may be NULL) + svr_bn_finalize(training = 1), one launch.
extern "C" int svr_bn_finalize_parts(const double *part, int32_t blocks, double *stats, const float *gamma, const float *beta,
                                     float *running_mean, float *running_var, float *scale_shift, float *mean_f32, int64_t rows,
                                     int32_t C, float eps, float momentum, void *stream) {
  SVR_CHECK(part && blocks > 0 && scale_shift && mean_f32 && rows > 0 && C > 0, SVR_E_BADARG, "bn_finalize_parts: bad argument");
  bn_stats_finalize_launch(part, stats, rows, C, blocks, gamma, beta, running_mean, running_var, scale_shift, mean_f32, eps, momentum,
                           (hipStream_t)stream);
  return launch_status("bn_finalize_parts");
}

extern "C" int svr_bn_finalize(const double *stats, const float *gamma, const float *beta, float *running_mean,
                               float *running_var, float *scale_shift, float *mean_f32, int64_t rows, int32_t C, float eps,
                               float momentum, int training, void *stream) {
  SVR_CHECK(scale_shift && mean_f32 && (training ? stats != nullptr : (running_mean && running_var)), SVR_E_BADARG, "bn_finalize: bad argument");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 64)), dim3(64), 0, (hipStream_t)stream, stats,
                     BnFinalize{gamma, beta, running_mean, running_var, scale_shift, mean_f32, eps, momentum}, rows, C, training);
  return launch_status("bn_finalize");
}

extern "C" int svr_bn_apply_pool(const float *x, const float *scale_shift, float *y, float *pooled, uint8_t *argmax,
                                 int32_t B, int32_t D, int32_t H, int32_t W, int32_t C, void *stream) {
  if (int rc = check_c(C)) return rc;
  SVR_CHECK(x && scale_shift && y, SVR_E_BADARG, "bn_apply_pool: null pointer");
  Vol v{B, D, H, W, C};
  int64_t total = (int64_t)B * ((D + 1) / 2) * ((H + 1) / 2) * ((W + 1) / 2) * (C / 4);
  if (total == 0) return SVR_OK;
  SVR_CHECK(total < (1LL << 32), SVR_E_UNSUPPORTED, "bn_apply_pool: %ld work items (32-bit index arithmetic)", (long)total);
  hipLaunchKernelGGL(bn_apply_pool_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x,
                     scale_shift, y, pooled, argmax, v, total);
  return launch_status("bn_apply_pool");
}

extern "C" int svr_bn_bwd_reduce(const float *x, const float *dy, const float *dpooled, const uint8_t *argmax,
                                 const float *mean_f32, const float *scale_shift, double *sums, int32_t B, int32_t D,
                                 int32_t H, int32_t W, int32_t C, void *workspace, void *stream) {
  if (int rc = check_c(C)) return rc;
  SVR_CHECK(x && mean_f32 && scale_shift && sums && workspace, SVR_E_BADARG, "bn_bwd_reduce: null pointer");
  SVR_CHECK(!dpooled || argmax, SVR_E_BADARG, "bn_bwd_reduce: dpooled needs argmax");
  hipStream_t s = (hipStream_t)stream;
  Vol v{B, D, H, W, C};
  int64_t cells = (int64_t)B * ((D + 1) / 2) * ((H + 1) / 2) * ((W + 1) / 2);
  SVR_CHECK(cells < (1LL << 32), SVR_E_UNSUPPORTED, "bn_bwd: %ld cells (32-bit index arithmetic)", (long)cells);
  int blocks = bwd_blocks(cells, C);
  hipLaunchKernelGGL(bn_bwd_kernel<false>, dim3(blocks), dim3(256), 0, s, x, dy, dpooled, argmax, mean_f32, scale_shift,
                     (const double *)nullptr, (double *)workspace, (float *)nullptr, v, cells, 0, (float *)nullptr, (float *)nullptr,
                     (uint32_t *)nullptr);
  hipLaunchKernelGGL(sum_parts_kernel, dim3(2 * C), dim3(256), 0, s, (const double *)workspace, sums, 2 * C, blocks);
  return launch_status("bn_bwd_reduce");
}

extern "C" int svr_bn_bwd_apply(const float *x, const float *dy, const float *dpooled, const uint8_t *argmax,
                                const float *mean_f32, const float *scale_shift, const float *gamma, const double *sums,
                                float *dx, float *dgamma, float *dbeta, int32_t B, int32_t D, int32_t H, int32_t W,
                                int32_t C, int relu_mask, uint32_t *amax_dx, void *stream) {
  (void)gamma;
  if (int rc = check_c(C)) return rc;
  SVR_CHECK(x && mean_f32 && scale_shift && sums && dx, SVR_E_BADARG, "bn_bwd_apply: null pointer");
  SVR_CHECK(!dpooled || argmax, SVR_E_BADARG, "bn_bwd_apply: dpooled needs argmax");
  hipStream_t s = (hipStream_t)stream;
  Vol v{B, D, H, W, C};
  int64_t cells = (int64_t)B * ((D + 1) / 2) * ((H + 1) / 2) * ((W + 1) / 2);
  SVR_CHECK(cells < (1LL << 32), SVR_E_UNSUPPORTED, "bn_bwd: %ld cells (32-bit index arithmetic)", (long)cells);
  int CL = 256 / (C / 4);
  int64_t blocks = cdiv(cells, CL);
  if (blocks > 65535 * 16) blocks = 65535 * 16;
  hipLaunchKernelGGL(bn_bwd_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, x, dy, dpooled, argmax, mean_f32,
                     scale_shift, sums, (double *)nullptr, dx, v, cells, relu_mask, dgamma, dbeta, amax_dx);
  return launch_status("bn_bwd_apply");
}
