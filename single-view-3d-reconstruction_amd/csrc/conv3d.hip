// 3x3x3 / padding-1 convolution on channels-last volumes for gfx950, exact-f32 MFMA (v_mfma_f32_32x32x2_f32).
//
// Replaces nn.Conv3d(ci, co, 3, padding=1) (+ fused bias / ReLU) of the IF-Net encoder
// (reference model/ifnet.py:126-135 and :164-191; 32-variant :69-74,100-113) and its autograd.
// Everything works on 4x4x8-voxel output BRICKS whose input halo tile (6x6x10 voxels) is staged in LDS once
// and then serves all 27 taps (a tap = a constant offset in the tile); the plain implicit GEMM re-read every
// input voxel 27 times from L2 and was bound by that (profiles/r01_step_v1 vs v4):
//   conv3d_brick_kernel            forward and backward-data (flipped, transposed weights): A = voxels x ci from
//                                  the LDS halo tile, B = the tap's [ci][co] weight slice, double buffered in LDS;
//   conv3d_bwd_weight_brick_kernel dWp[tap][ci][co] = sum_voxels in[voxel+tap][ci] dout[voxel][co]: voxels are the
//                                  MFMA k, 27 accumulator tiles stay in registers across a persistent brick loop;
//   conv3d_c1_* / conv3d_to1       Ci == 1 (conv_in): forward as a [voxels x 27] x [27 x Co] MFMA product from a
//                                  scalar halo tile, weight gradient as a 27 x Co outer product, data gradient direct.
#include "common.h"
#include <algorithm>
#include "gemm_core.h"

using namespace svr;

namespace svr {
// defined in gemm.hip
void colsum_launch(const float *Y, int64_t ldy, float *out, float *part, int64_t M, int64_t N, hipStream_t s);
int64_t colsum_workspace_floats(int64_t M, int64_t N);
// defined in bn_pool.hip
void bn_stats_final_launch(const double *part, double *stats, int64_t rows, int C, int blocks, hipStream_t s);
}  // namespace svr

namespace {

__global__ void pack_weight_kernel(const float *__restrict__ W, float *__restrict__ Wf, float *__restrict__ Wb, int Ci,
                                   int Co) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;  // over (co, ci, tap) in the source order
  if (idx >= Co * Ci * 27) return;
  int tap = idx % 27, ci = (idx / 27) % Ci, co = idx / (27 * Ci);
  float v = W[idx];
  if (Wf) Wf[((size_t)tap * Ci + ci) * Co + co] = v;
  if (Wb) Wb[((size_t)(26 - tap) * Co + co) * Ci + ci] = v;
}

__global__ void unpack_wgrad_kernel(const float *__restrict__ dWp, float *__restrict__ dW, int Ci, int Co) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= Co * Ci * 27) return;
  int tap = idx % 27, ci = (idx / 27) % Ci, co = idx / (27 * Ci);
  dW[idx] = dWp[((size_t)tap * Ci + ci) * Co + co];
}

struct ConvShape {
  int B, D, H, W, Ci, Co;
};

// ---- Ci == 1 forward on the matrix core (production): out[voxel][co] = sum_tap in[voxel+tap] W[tap][co] is a
// [voxels x 27] x [27 x CO] product.  One workgroup = one 8x8x8 brick whose scalar halo tile (10^3 floats)
// sits in LDS; A[i = voxel][k = tap] is read from it, B[k = tap][j = co] lives in 14 registers per lane.
// 14 MFMAs (32x32x2, exact f32) per 32 voxels replace 27*CO LDS-fed FMAs per voxel; the kernel is then bound
// by its 64 B/voxel output stream.
constexpr int C1B = 8, C1H = C1B + 2;
template <int CO>
__global__ __launch_bounds__(256) void conv3d_c1_mfma_kernel(const float *__restrict__ in, const float *__restrict__ Wp,
                                                             const float *__restrict__ bias, float *__restrict__ out,
                                                             ConvShape s, int nbz, int nby, int nbx, int mode,
                                                             double *__restrict__ spart) {
  __shared__ float tile[C1H * C1H * C1H];
  __shared__ __attribute__((aligned(16))) float otile[4 * 32 * (CO + 4)];  // per-wave output staging, rows padded by 16 B
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  int64_t q = blockIdx.x;
  const int bx = (int)(q % nbx); q /= nbx;
  const int by = (int)(q % nby); q /= nby;
  const int bz = (int)(q % nbz);
  const int64_t b = q / nbz;
  const int z0 = bz * C1B, y0 = by * C1B, x0 = bx * C1B;
  const float *inb = in + b * (int64_t)s.D * s.H * s.W;
  {  // halo tile: the four loads of a thread are issued together (clamped coordinates), zero padding applied after
    constexpr int NH = C1H * C1H * C1H, HIT = (NH + 255) / 256;
    float hv[HIT];
    uint32_t hok = 0;
#pragma unroll
    for (int i = 0; i < HIT; ++i) {
      const int idx = min(t + 256 * i, NH - 1);
      const int hx = idx % C1H, hy = (idx / C1H) % C1H, hz = idx / (C1H * C1H);
      const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = x0 + hx - 1;
      if (gz >= 0 && gz < s.D && gy >= 0 && gy < s.H && gx >= 0 && gx < s.W) hok |= 1u << i;
      hv[i] = inb[((int64_t)min(max(gz, 0), s.D - 1) * s.H + min(max(gy, 0), s.H - 1)) * s.W + min(max(gx, 0), s.W - 1)];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < HIT; ++i)
      if (t + 256 * i < NH) tile[t + 256 * i] = ((hok >> i) & 1u) ? hv[i] : 0.f;
  }
  float bw[14];
  int toff[14];
#pragma unroll
  for (int kp = 0; kp < 14; ++kp) {
    const int tap = 2 * kp + lh;
    bw[kp] = (tap < 27 && l31 < CO) ? Wp[tap * CO + l31] : 0.f;
    const int tc = tap < 27 ? tap : 26;
    toff[kp] = ((tc / 9 - 1) * C1H + ((tc / 3) % 3 - 1)) * C1H + (tc % 3 - 1);
  }
  const float bv = (mode != SVR_EPI_NONE && l31 < CO) ? bias[l31] : 0.f;
  float ls[4] = {0.f, 0.f, 0.f, 0.f}, lq[4] = {0.f, 0.f, 0.f, 0.f};  // this lane's channel quad: sum, sum of squares
  __syncthreads();
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) {  // wave w: z-slices 2w, 2w+1, two 32-voxel row tiles each
    const int vz = 2 * wave + (rt >> 1), yh = rt & 1;
    const int hb = ((vz + 1) * C1H + (yh * 4 + l31 / C1B) + 1) * C1H + (l31 % C1B) + 1;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int kp = 0; kp < 14; ++kp) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(tile[hb + toff[kp]], bw[kp], acc, 0, 0, 0);
    // epilogue through LDS: the MFMA layout would store 64-byte pieces (16 channels of one voxel per half wave, 16
    // instructions per tile); restaged as [voxel][co], every lane writes one float4 and a wave-instruction covers
    // two 512-byte x-rows of the brick
    float *ot = otile + wave * (32 * (CO + 4));
    if (l31 < CO) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * lh;
        float v = acc[r] + bv;
        if (mode == SVR_EPI_BIAS_RELU) v = fmaxf(v, 0.f);
        ot[i * (CO + 4) + l31] = v;
      }
    }
    __syncthreads();
    {
      constexpr int G4 = CO / 4, VPI = 64 / G4;  // float4 groups per voxel, voxels per wave-instruction
      const int gz = z0 + vz;
#pragma unroll
      for (int p = 0; p < 32 / VPI; ++p) {
        const int i = p * VPI + lane / G4, c4 = (lane % G4) * 4;
        const int gy = y0 + yh * 4 + i / C1B, gx = x0 + i % C1B;
        if (gz < s.D && gy < s.H && gx < s.W) {
          const float4 o4 = *reinterpret_cast<const float4 *>(ot + i * (CO + 4) + c4);
          *reinterpret_cast<float4 *>(out + ((((int64_t)b * s.D + gz) * s.H + gy) * s.W + gx) * CO + c4) = o4;
          if (spart) {  // BatchNorm statistics of what was just written (the following BN would re-read all of it)
            ls[0] += o4.x; ls[1] += o4.y; ls[2] += o4.z; ls[3] += o4.w;
            lq[0] += o4.x * o4.x; lq[1] += o4.y * o4.y; lq[2] += o4.z * o4.z; lq[3] += o4.w * o4.w;
          }
        }
      }
    }
    __syncthreads();
  }
  if (spart) {  // per-workgroup sums in f64 (f32 partials of <= 8 values per lane), fixed order
    constexpr int G4 = CO / 4;
    float *red = otile;  // 256 x 8 floats <= 4 * 32 * (CO + 4)
    static_assert(256 * 8 <= 4 * 32 * (CO + 4), "staging buffer too small for the statistics reduction");
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[t * 8 + e] = ls[e]; red[t * 8 + 4 + e] = lq[e]; }
    __syncthreads();
    if (t < 2 * CO) {
      const int ch = t % CO, which = t / CO;  // 0: sum, 1: sum of squares
      double acc = 0.0;
      for (int k = 0; k < 256 / G4; ++k) acc += (double)red[(k * G4 + ch / 4) * 8 + which * 4 + (ch & 3)];
      spart[(int64_t)blockIdx.x * 2 * CO + which * CO + ch] = acc;
    }
  }
}

// ---- Co == 1 "forward" = backward-data of a Ci==1 conv: out[m] = sum_{tap,ci} in[m+tap][ci] Wp[tap][ci]
template <int CI>
__global__ __launch_bounds__(256) void conv3d_to1_kernel(const float *__restrict__ in, const float *__restrict__ Wp,
                                                         float *__restrict__ out, ConvShape s) {
  __shared__ float w[27 * CI];
  for (int i = threadIdx.x; i < 27 * CI; i += blockDim.x) w[i] = Wp[i];
  __syncthreads();
  const int64_t M = (int64_t)s.B * s.D * s.H * s.W;
  int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  int x = (int)(m % s.W);
  int64_t r = m / s.W;
  int y = (int)(r % s.H);
  r /= s.H;
  int z = (int)(r % s.D);
  int64_t b = r / s.D;
  const float *ib = in + b * s.D * s.H * s.W * CI;
  float acc = 0.f;
  for (int tap = 0; tap < 27; ++tap) {
    int zz = z + tap / 9 - 1, yy = y + (tap / 3) % 3 - 1, xx = x + tap % 3 - 1;
    if (zz >= 0 && zz < s.D && yy >= 0 && yy < s.H && xx >= 0 && xx < s.W) {
      const float *p = ib + (((int64_t)zz * s.H + yy) * s.W + xx) * CI;
#pragma unroll
      for (int c = 0; c < CI; c += 4) {
        float4 v = *reinterpret_cast<const float4 *>(p + c);
        acc += v.x * w[tap * CI + c] + v.y * w[tap * CI + c + 1] + v.z * w[tap * CI + c + 2] + v.w * w[tap * CI + c + 3];
      }
    }
  }
  out[m] = acc;
}

// The same sum, round 4 (config 5 needs the occupancy grid's gradient: 0.87 ms at 4 x 128^3 x 16 with the kernel above, which
// issues 27 x CI/4 sixteen-byte loads per output at 64-byte lane strides -- bound by the address path, 32 lines per wave load).
// Here a wave owns 64 / (CI/4) consecutive x and a column of ZT outputs along z; a lane owns ONE channel quad of one x, so a wave
// load is 1 KB contiguous (8 lines); for every (dy, dx) the lane walks the ZT + 2 input voxels of its column once and feeds each
// into the (up to) three outputs it belongs to, with that (dy, dx)'s 3 x 4 weights in registers: (ZT + 2) / ZT x 9 loads per
// output and quad instead of 27.  The quads of an output are summed with two (three) xor-shuffles.
template <int CI, int ZT>
__global__ __launch_bounds__(256) void conv3d_to1_col_kernel(const float *__restrict__ in, const float *__restrict__ Wp,
                                                             float *__restrict__ out, ConvShape s, int nxb, int nzb) {
  constexpr int QL = CI / 4, XL = 64 / QL;
  __shared__ __attribute__((aligned(16))) float w[27 * CI];
  for (int i = threadIdx.x; i < 27 * CI; i += blockDim.x) w[i] = Wp[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, q = lane % QL, xi = lane / QL;
  const int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);      // one wave = one (b, z block, y, x block)
  const int64_t total = (int64_t)s.B * nzb * s.H * nxb;
  if (wid >= total) return;
  const int xb = (int)(wid % nxb);
  int64_t r = wid / nxb;
  const int y = (int)(r % s.H);
  r /= s.H;
  const int z0 = (int)(r % nzb) * ZT;
  const int64_t b = r / nzb;
  const int x = xb * XL + xi;
  const float *ib = in + b * s.D * s.H * s.W * CI + q * 4;
  float acc[ZT];
#pragma unroll
  for (int o = 0; o < ZT; ++o) acc[o] = 0.f;
  // (runtime loops: fully unrolled, the compiler put all nine columns' loads in flight -- 500 registers, one wave per SIMD)
#pragma unroll 1
  for (int dy = 0; dy < 3; ++dy) {
    const int yy = y + dy - 1;
#pragma unroll 1
    for (int dx = 0; dx < 3; ++dx) {
      const int xx = x + dx - 1;
      const bool okxy = yy >= 0 && yy < s.H && xx >= 0 && xx < s.W;
      float4 k[3];
#pragma unroll
      for (int dz = 0; dz < 3; ++dz) k[dz] = *reinterpret_cast<const float4 *>(w + ((dz * 3 + dy) * 3 + dx) * CI + q * 4);
      const float *col = ib + ((int64_t)(okxy ? yy : 0) * s.W + (okxy ? xx : 0)) * CI;
      float4 v[ZT + 2];
#pragma unroll
      for (int j = 0; j < ZT + 2; ++j) {     // input voxel z0 - 1 + j
        const int zz = z0 - 1 + j;
        v[j] = (okxy && zz >= 0 && zz < s.D) ? *reinterpret_cast<const float4 *>(col + (int64_t)zz * s.H * s.W * CI) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int o = 0; o < ZT; ++o)
#pragma unroll
        for (int dz = 0; dz < 3; ++dz) {      // output z0 + o reads input z0 + o + dz - 1 = v[o + dz]
          const float4 a = v[o + dz];
          acc[o] += a.x * k[dz].x + a.y * k[dz].y + a.z * k[dz].z + a.w * k[dz].w;
        }
    }
  }
#pragma unroll
  for (int o = 0; o < ZT; ++o) {
#pragma unroll
    for (int d = 1; d < QL; d <<= 1) acc[o] += __shfl_xor(acc[o], d);
    if (q == 0 && x < s.W && z0 + o < s.D) out[((b * s.D + z0 + o) * s.H + y) * s.W + x] = acc[o];
  }
}

constexpr int BW_ROWS = 64;   // (b,z,y) rows per workgroup of the Ci == 1 weight-gradient kernel: 2048 workgroups at
                              // 128^3 x 8 (with 256 rows the grid was 2 workgroups per CU and the kernel latency bound)

// ---- backward-weight, brick version (production): a workgroup walks 4x4x8-voxel bricks of the output.
// Per brick it stages the 32-channel slice of the input WITH its one-voxel halo (6x6x10 voxels) and the
// 32-channel slice of dout in LDS once, and all 27 taps are then served from LDS: the shifted input voxel
// of a tap is a constant offset in the halo tile.  The voxels are the MFMA k (32x32x2, exact f32):
//   A[i = ci][k = voxel] = in[voxel + tap][ci]   (32 consecutive channels of one voxel: conflict free)
//   B[k = voxel][j = co] = dout[voxel][co]       (loaded once per k pair, reused by the wave's 7 taps)
// Wave w owns taps w, w+4, ...; the 27 accumulator tiles stay in registers across ALL bricks of the
// workgroup (persistent loop), so the partial sums leave the chip once per workgroup (slab + ordered sum).
constexpr int BRZ = 4, BRY = 4, BRX = 8, BRV = BRZ * BRY * BRX;          // brick: 128 voxels
constexpr int HLZ = BRZ + 2, HLY = BRY + 2, HLX = BRX + 2, HLV = HLZ * HLY * HLX;  // halo tile: 360 voxels
constexpr int BCP = 33;                                                  // padded channel stride (floats)

__global__ __launch_bounds__(256) void conv3d_bwd_weight_brick_kernel(const float *__restrict__ in,
                                                                      const float *__restrict__ dout,
                                                                      float *__restrict__ slab, ConvShape s, int nbz,
                                                                      int nby, int nbx, int co_tiles) {
  __shared__ float sin[HLV * BCP];
  __shared__ float sdo[BRV * BCP];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int pair = blockIdx.y;
  const int ci0 = (pair / co_tiles) * 32, co0 = (pair % co_tiles) * 32;
  const int64_t bricks = (int64_t)s.B * nbz * nby * nbx;
  int tapoff[7];
#pragma unroll
  for (int ti = 0; ti < 7; ++ti) {
    const int tap = wave + 4 * ti;
    tapoff[ti] = ((tap / 9 - 1) * HLY + ((tap / 3) % 3 - 1)) * HLX + (tap % 3 - 1);
  }
  f32x16 acc[7];
#pragma unroll
  for (int ti = 0; ti < 7; ++ti)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[ti][r] = 0.f;

  for (int64_t brick = blockIdx.x; brick < bricks; brick += gridDim.x) {
    int64_t q = brick;
    const int bx = (int)(q % nbx); q /= nbx;
    const int by = (int)(q % nby); q /= nby;
    const int bz = (int)(q % nbz);
    const int64_t b = q / nbz;
    const int z0 = bz * BRZ, y0 = by * BRY, x0 = bx * BRX;
    const float *inb = in + b * (int64_t)s.D * s.H * s.W * s.Ci;
    const float *dob = dout + b * (int64_t)s.D * s.H * s.W * s.Co;
    // stage the input halo tile: (voxel, float4 of 4 channels)
    for (int idx = t; idx < HLV * 8; idx += 256) {
      const int hv = idx >> 3, c4 = (idx & 7) * 4;
      const int hx = hv % HLX, hy = (hv / HLX) % HLY, hz = hv / (HLX * HLY);
      const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = x0 + hx - 1;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gz >= 0 && gz < s.D && gy >= 0 && gy < s.H && gx >= 0 && gx < s.W && ci0 + c4 < s.Ci)
        v = *reinterpret_cast<const float4 *>(inb + (((int64_t)gz * s.H + gy) * s.W + gx) * s.Ci + ci0 + c4);
      float *d = sin + hv * BCP + c4;
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    for (int idx = t; idx < BRV * 8; idx += 256) {
      const int v_ = idx >> 3, c4 = (idx & 7) * 4;
      const int vx = v_ % BRX, vy = (v_ / BRX) % BRY, vz = v_ / (BRX * BRY);
      const int gz = z0 + vz, gy = y0 + vy, gx = x0 + vx;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gz < s.D && gy < s.H && gx < s.W && co0 + c4 < s.Co)
        v = *reinterpret_cast<const float4 *>(dob + (((int64_t)gz * s.H + gy) * s.W + gx) * s.Co + co0 + c4);
      float *d = sdo + v_ * BCP + c4;
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    __syncthreads();
#pragma unroll 4
    for (int kp = 0; kp < BRV / 2; ++kp) {
      const int v_ = 2 * kp + lh;
      const int vx = v_ % BRX, vy = (v_ / BRX) % BRY, vz = v_ / (BRX * BRY);
      const int hbase = (((vz + 1) * HLY + vy + 1) * HLX + vx + 1) * BCP + l31;
      const float bv = sdo[v_ * BCP + l31];
#pragma unroll
      for (int ti = 0; ti < 7; ++ti) {
        if (wave + 4 * ti < 27) {  // wave-uniform
          const float av = sin[hbase + tapoff[ti] * BCP];
          acc[ti] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[ti], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }
  // slab layout shared with the reduce kernel: [part][tap][pair][32 ci][32 co]
  const int pairs = gridDim.y;
#pragma unroll
  for (int ti = 0; ti < 7; ++ti) {
    const int tap = wave + 4 * ti;
    if (tap < 27) {
      float *o = slab + ((((int64_t)blockIdx.x) * 27 + tap) * pairs + pair) * 1024;
#pragma unroll
      for (int r = 0; r < 16; ++r) o[((r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + l31] = acc[ti][r];
    }
  }
}

// ---- forward / backward-data, brick version (production): one workgroup = one 4x4x8 output brick, wave w
// = z-slice w (32 voxels = one MFMA row tile), all Co columns.  Per CK-channel chunk of the input the halo
// tile (6x6x10 voxels) is staged in LDS once and serves all 27 taps (the plain implicit GEMM re-reads every
// input voxel 27 times from L2, which is what bounds the Co = 32 layers).  Exact f32 (32x32x2 MFMA):
//   A[i = voxel][k = ci] from LDS (voxel stride CK+1 floats: conflict free), B[k = ci][j = co] = Wp rows
//   straight from global memory (128-B rows, shared by every wave on the chip -> L1/L2 hits).
template <int CK, int TNB>
__global__ __launch_bounds__(256) void conv3d_brick_kernel(const float *__restrict__ in, const float *__restrict__ Wp,
                                                           const float *__restrict__ bias, float *__restrict__ out,
                                                           const float *__restrict__ mask, ConvShape s, int nbz, int nby,
                                                           int nbx, int mode) {
  constexpr int CP = CK + 1;
  constexpr int NC = TNB * 32;                 // output columns of this workgroup
  constexpr int TG = TNB == 1 ? 3 : 1;         // taps per barrier (narrow tiles: 16 MFMAs per tap are too few)
  constexpr int WQ = TG * CK * NC / 4;         // float4 per weight group (TG taps, CK input channels)
  constexpr int WR_ = (WQ + 255) / 256;        // float4 per thread
  __shared__ float sin[HLV * CP];
  __shared__ float sw[2][TG * CK * NC];        // weight slices of the current / next tap group: [tap][ci][co]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  int64_t q = blockIdx.x;
  const int bx = (int)(q % nbx); q /= nbx;
  const int by = (int)(q % nby); q /= nby;
  const int bz = (int)(q % nbz);
  const int64_t b = q / nbz;
  const int z0 = bz * BRZ, y0 = by * BRY, x0 = bx * BRX;
  const int co0 = blockIdx.y * NC;
  const float *inb = in + b * (int64_t)s.D * s.H * s.W * s.Ci;
  // this lane's A voxel inside the wave's z-slice
  const int avy = l31 / BRX, avx = l31 % BRX;
  const int hbase = (((wave + 1) * HLY + avy + 1) * HLX + avx + 1) * CP;
  f32x16 acc[TNB];
#pragma unroll
  for (int j = 0; j < TNB; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  float4 wreg[WR_];
  auto wload = [&](int ci0, int tap0) {   // taps tap0 .. tap0+TG-1
#pragma unroll
    for (int i = 0; i < WR_; ++i) {
      const int idx = t + 256 * i;
      const int tg = idx / (CK * NC / 4), rem = idx % (CK * NC / 4);
      const int k = rem / (NC / 4), c4 = (rem % (NC / 4)) * 4;
      wreg[i] = (idx < WQ && co0 + c4 < s.Co)
                    ? *reinterpret_cast<const float4 *>(Wp + ((int64_t)(tap0 + tg) * s.Ci + ci0 + k) * s.Co + co0 + c4)
                    : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto wstore = [&](float *dst) {
#pragma unroll
    for (int i = 0; i < WR_; ++i) {
      const int idx = t + 256 * i;
      if (idx < WQ) *reinterpret_cast<float4 *>(dst + idx * 4) = wreg[i];
    }
  };

  for (int ci0 = 0; ci0 < s.Ci; ci0 += CK) {
    wload(ci0, 0);
    if (ci0 > 0) __syncthreads();  // every wave is done with the previous chunk's halo tile and weight slices
    for (int idx = t; idx < HLV * (CK / 4); idx += 256) {
      const int hv = idx / (CK / 4), c4 = (idx % (CK / 4)) * 4;
      const int hx = hv % HLX, hy = (hv / HLX) % HLY, hz = hv / (HLX * HLY);
      const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = x0 + hx - 1;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gz >= 0 && gz < s.D && gy >= 0 && gy < s.H && gx >= 0 && gx < s.W)
        v = *reinterpret_cast<const float4 *>(inb + (((int64_t)gz * s.H + gy) * s.W + gx) * s.Ci + ci0 + c4);
      float *d = sin + hv * CP + c4;
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    wstore(sw[0]);
    __syncthreads();
    for (int tap0 = 0; tap0 < 27; tap0 += TG) {
      if (tap0 + TG < 27) wload(ci0, tap0 + TG);
      const float *wg = sw[(tap0 / TG) & 1] + lh * NC + l31;
#pragma unroll
      for (int tg = 0; tg < TG; ++tg) {
        const int tap = tap0 + tg;
        const int toff = (((tap / 9 - 1) * HLY + ((tap / 3) % 3 - 1)) * HLX + (tap % 3 - 1)) * CP;
        const float *wb = wg + tg * CK * NC;
#pragma unroll
        for (int kp = 0; kp < CK / 2; ++kp) {
          const float av = sin[hbase + toff + 2 * kp + lh];
#pragma unroll
          for (int j = 0; j < TNB; ++j)
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, wb[2 * kp * NC + j * 32], acc[j], 0, 0, 0);
        }
      }
      if (tap0 + TG < 27) wstore(sw[((tap0 / TG) + 1) & 1]);
      __syncthreads();
    }
  }
#pragma unroll
  for (int j = 0; j < TNB; ++j) {
    const int co = co0 + j * 32 + l31;
    if (co >= s.Co) continue;
    const float bv = (mode == SVR_EPI_BIAS || mode == SVR_EPI_BIAS_RELU) ? bias[co] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = (r & 3) + 8 * (r >> 2) + 4 * lh;
      const int gz = z0 + wave, gy = y0 + i / BRX, gx = x0 + i % BRX;
      if (gz < s.D && gy < s.H && gx < s.W) {
        const int64_t o = ((((int64_t)b * s.D + gz) * s.H + gy) * s.W + gx) * s.Co + co;
        float v = acc[j][r] + bv;
        if (mode == SVR_EPI_BIAS_RELU) v = fmaxf(v, 0.f);
        if (mode == SVR_EPI_MASK) v = mask[o] > 0.f ? v : 0.f;
        out[o] = v;
      }
    }
  }
}

// dWp = ordered f64 sum of the workgroup slabs: 64 outputs x 4 part lanes per workgroup.  param_layout: the sums go
// straight into the parameter's layout (Co,Ci,3,3,3) instead of [tap][ci][co] (no separate unpack launch).  Workgroups
// past the weight's (blockIdx.x >= wblocks) reduce the bias-gradient partials dbpart [dbparts][Co] -> db instead: the
// whole post-processing of a weight gradient is ONE launch (it was three launch-bound kernels per layer on the main stream).
__global__ __launch_bounds__(256) void conv3d_bwd_weight_reduce_kernel(const float *__restrict__ slab,
                                                                       float *__restrict__ dWp, int Ci, int Co,
                                                                       int ci_tiles, int co_tiles, int parts,
                                                                       int param_layout, int wblocks,
                                                                       const float *__restrict__ dbpart,
                                                                       float *__restrict__ db, int dbparts) {
  __shared__ double red[256];
  if ((int)blockIdx.x >= wblocks) {   // bias gradient of output channel co: ordered f64 tree over the partials
    const int co = blockIdx.x - wblocks;
    double sum = 0.0;
    for (int p = threadIdx.x; p < dbparts; p += 256) sum += (double)dbpart[(int64_t)p * Co + co];
    red[threadIdx.x] = sum;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) db[co] = (float)red[0];
    return;
  }
  const int per = 27 * ci_tiles * co_tiles * 1024;  // [tap][tile][32][32]
  const int idx = blockIdx.x * 64 + (threadIdx.x & 63);
  const int pl = threadIdx.x >> 6;
  double sum = 0.0;
  {  // eight slab loads in flight per lane (one at a time was a latency chain of parts / 4 round trips), fixed order
    const int ic = idx < per ? idx : per - 1;
    int p = pl;
    for (; p + 28 < parts; p += 32) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = slab[(int64_t)(p + 4 * u) * per + ic];
#pragma unroll
      for (int u = 0; u < 8; ++u) sum += (double)v[u];
    }
    for (; p < parts; p += 4) sum += (double)slab[(int64_t)p * per + ic];
  }
  red[threadIdx.x] = sum;
  __syncthreads();
  if (pl == 0 && idx < per) {
    const double tot = red[threadIdx.x] + red[threadIdx.x + 64] + red[threadIdx.x + 128] + red[threadIdx.x + 192];
    const int j = idx & 31, i = (idx >> 5) & 31;
    const int tile = (idx >> 10) % (ci_tiles * co_tiles), tap = idx / (1024 * ci_tiles * co_tiles);
    const int ci = (tile / co_tiles) * 32 + i, co = (tile % co_tiles) * 32 + j;
    if (ci < Ci && co < Co) {
      if (param_layout) dWp[((size_t)co * Ci + ci) * 27 + tap] = (float)tot;
      else dWp[((size_t)tap * Ci + ci) * Co + co] = (float)tot;
    }
  }
}

// ---- Ci == 1 backward-weight: A[i = tap][k = voxel] = in[voxel + tap], B[k = voxel][j = co] = dout
__global__ __launch_bounds__(256) void conv3d_c1_bwd_weight_kernel(const float *__restrict__ in,
                                                                   const float *__restrict__ dout,
                                                                   float *__restrict__ slab, ConvShape s) {
  const int chunk = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int tap = l31;  // as the A row
  const int dz = tap / 9 - 1, dy = (tap / 3) % 3 - 1, dx = tap % 3 - 1;
  const bool tap_ok = tap < 27, co_ok = l31 < s.Co;
  const int64_t nrows = (int64_t)s.B * s.D * s.H;
  const int64_t r0 = (int64_t)chunk * BW_ROWS;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int64_t row = r0 + wave; row < r0 + BW_ROWS && row < nrows; row += 4) {
    int y = (int)(row % s.H);
    int64_t t = row / s.H;
    int z = (int)(t % s.D);
    int64_t b = t / s.D;
    int zz = z + dz, yy = y + dy;
    bool rok = tap_ok && zz >= 0 && zz < s.D && yy >= 0 && yy < s.H;
    const float *arow = in + ((b * s.D + (rok ? zz : 0)) * s.H + (rok ? yy : 0)) * (int64_t)s.W;
    const float *brow = dout + row * (int64_t)s.W * s.Co + (co_ok ? l31 : 0);
    for (int x0 = 0; x0 < s.W; x0 += 16) {  // 16 loads in flight per lane, then 8 MFMAs
      float av[8], bv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int x = x0 + 2 * u + lh, xs = x + dx;
        const bool aok = rok && x < s.W && xs >= 0 && xs < s.W, bok = co_ok && x < s.W;
        const float a = arow[aok ? xs : 0], bq = brow[bok ? (int64_t)x * s.Co : 0];  // unconditional, clamped
        av[u] = aok ? a : 0.f;
        bv[u] = bok ? bq : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
    }
  }
  // the four waves' tiles are summed in a fixed order through LDS: one [tap 32][co 32] slab per workgroup
  __shared__ float red[4 * 1024];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    int i = (r & 3) + 8 * (r >> 2) + 4 * lh;
    red[wave * 1024 + i * 32 + l31] = acc[r];
  }
  __syncthreads();
  float *o = slab + (int64_t)chunk * 1024;
  for (int e = threadIdx.x; e < 1024; e += 256) o[e] = ((red[e] + red[1024 + e]) + red[2048 + e]) + red[3072 + e];
}

// ---- Ci == 1, Co == 16 (conv_in): v_mfma_f32_16x16x4f32 with k = 4 consecutive voxels.  The B operand
// (lane: co = lane & 15, voxel = lane >> 4) is then one fully coalesced 256-byte load of dout per instruction;
// the 27 taps are two 16-row A tiles.  Same slab layout as the generic kernel.
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void conv3d_c1_bwd_weight_co16_kernel(const float *__restrict__ in,
                                                                        const float *__restrict__ dout,
                                                                        float *__restrict__ slab, ConvShape s,
                                                                        float *__restrict__ dbpart) {
  __shared__ float red[4 * 1024];
  const int chunk = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, kq = lane >> 4;
  float dbs = 0.f;  // bias gradient of channel l15 over this lane's voxels (dout is read exactly once, here)
  for (int e = threadIdx.x; e < 4096; e += 256) red[e] = 0.f;
  int dz[2], dy[2], dx[2];
  bool tok[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int tap = 16 * h + l15;
    dz[h] = tap / 9 - 1; dy[h] = (tap / 3) % 3 - 1; dx[h] = tap % 3 - 1;
    tok[h] = tap < 27;
  }
  const int64_t nrows = (int64_t)s.B * s.D * s.H;
  const int64_t r0 = (int64_t)chunk * BW_ROWS;
  f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  for (int64_t row = r0 + wave; row < r0 + BW_ROWS && row < nrows; row += 4) {
    const int y = (int)(row % s.H);
    const int64_t t = row / s.H;
    const int z = (int)(t % s.D);
    const int64_t b = t / s.D;
    // wave-uniform bases + 32-bit lane offsets (scalar-base addressing, no 64-bit lane arithmetic in the loop)
    const float *inb = in + b * (int64_t)s.D * s.H * s.W;
    const float *dob = dout + row * (int64_t)s.W * 16;
    int aoff[2];
    bool rok[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int zz = z + dz[h], yy = y + dy[h];
      rok[h] = tok[h] && zz >= 0 && zz < s.D && yy >= 0 && yy < s.H;
      aoff[h] = rok[h] ? (zz * s.H + yy) * s.W : 0;
    }
    for (int x0 = 0; x0 < s.W; x0 += 32) {  // 8 groups of 4 voxels: 24 loads in flight, then 16 MFMAs
      const bool full = x0 + 32 <= s.W;  // wave-uniform
      // phase 1: all 24 loads, unconditional, from clamped coordinates, plus the validity masks; phase 2 (behind a
      // scheduling barrier, otherwise the compiler interleaves load / wait / MFMA one at a time): mask and multiply.
      // The loaded value is always consumed (bit mask, not a select) so the load cannot be sunk into a branch.
      float ar[2][8], br[8];
      uint32_t am[2][8], bm[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int x = x0 + 4 * u + kq;
        const bool bok = full || x < s.W;
        br[u] = dob[(bok ? x : 0) * 16 + l15];
        bm[u] = bok ? 0xffffffffu : 0u;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int xs = x + dx[h], xc = min(max(xs, 0), s.W - 1);
          ar[h][u] = inb[aoff[h] + xc];
          am[h][u] = (rok[h] && xs == xc && bok) ? 0xffffffffu : 0u;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float bv = __uint_as_float(__float_as_uint(br[u]) & bm[u]);
        dbs += bv;
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(__float_as_uint(ar[0][u]) & am[0][u]), bv, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(__float_as_uint(ar[1][u]) & am[1][u]), bv, acc[1], 0, 0, 0);
      }
    }
  }
  __syncthreads();  // red is zeroed
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave * 1024 + (16 * h + 4 * kq + r) * 32 + l15] = acc[h][r];
  __syncthreads();
  float *o = slab + (int64_t)chunk * 1024;
  for (int e = threadIdx.x; e < 1024; e += 256) o[e] = ((red[e] + red[1024 + e]) + red[2048 + e]) + red[3072 + e];
  if (dbpart) {  // fixed-order sum of the 16 (wave, voxel-slot) partials per channel
    __syncthreads();
    red[threadIdx.x] = dbs;
    __syncthreads();
    if (threadIdx.x < 16) {
      float sum = 0.f;
      for (int i = 0; i < 16; ++i) sum += red[i * 16 + threadIdx.x];
      dbpart[(int64_t)chunk * 16 + threadIdx.x] = sum;
    }
  }
}

// ---- Ci == 1, Co == 16, W <= 128: the same product with the A operand served from LDS.  A workgroup owns 16 x-rows
// (one y block of one z plane); the 3 x 18 input rows it can touch sit in LDS with their zero padding, so
// A[tap][voxel] = in[voxel + tap] is one ds_read_b32 at a lane-constant offset (no bounds checks, no scattered
// global loads: the texture addresser was the limit of the kernel above: 16 scattered dword loads per 32 voxels).
constexpr int C1L_ROWS = 16, C1L_WP = 135, C1L_PS = 2453;  // row stride = 7, plane stride = 21 (mod 64): the 9 (dz,dy)
                                                           // rows of a tap tile x 6 columns land in distinct banks
__global__ __launch_bounds__(256) void conv3d_c1_bwd_weight_co16_lds_kernel(const float *__restrict__ in,
                                                                            const float *__restrict__ dout,
                                                                            float *__restrict__ slab, ConvShape s, int nyb,
                                                                            float *__restrict__ dbpart) {
  __shared__ float tile[3 * C1L_PS];
  __shared__ float red[4 * 1024];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int l15 = lane & 15, kq = lane >> 4;
  const int chunk = blockIdx.x;
  const int yb = chunk % nyb, z = (chunk / nyb) % s.D;
  const int64_t b = chunk / ((int64_t)nyb * s.D);
  const int y0 = yb * C1L_ROWS;
  const float *inb = in + b * (int64_t)s.D * s.H * s.W;
  for (int e = t; e < 4096; e += 256) red[e] = 0.f;
  // ---- stage the 3 x 18 x (W + 2) input window, zero padded; 8 loads in flight per thread
  const int WW = s.W + 2, total = 3 * 18 * WW;
  for (int base = 0; base < total; base += 256 * 8) {
    float v[8];
    uint32_t okm = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int idx = min(base + t + 256 * i, total - 1);
      const int r = idx / WW, xx = idx % WW;
      const int gz = z + r / 18 - 1, gy = y0 + r % 18 - 1, gx = xx - 1;
      if (gz >= 0 && gz < s.D && gy >= 0 && gy < s.H && gx >= 0 && gx < s.W) okm |= 1u << i;
      v[i] = inb[((int64_t)min(max(gz, 0), s.D - 1) * s.H + min(max(gy, 0), s.H - 1)) * s.W + min(max(gx, 0), s.W - 1)];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int idx = base + t + 256 * i;
      if (idx < total) {
        const int r = idx / WW, xx = idx % WW;
        tile[(r / 18) * C1L_PS + (r % 18) * C1L_WP + xx] = ((okm >> i) & 1u) ? v[i] : 0.f;
      }
    }
  }
  __syncthreads();
  // ---- lane constants: tap of each of the two 16-row A tiles -> LDS offset of (dz, dy, dx)
  int aoff[2];
  uint32_t amask[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int tap = 16 * h + l15, tc = tap < 27 ? tap : 26;
    aoff[h] = (tc / 9) * C1L_PS + ((tc / 3) % 3) * C1L_WP + (tc % 3);
    amask[h] = tap < 27 ? 0xffffffffu : 0u;
  }
  f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  float dbs = 0.f;
  for (int yl = wave; yl < C1L_ROWS; yl += 4) {
    const int gy = y0 + yl;
    if (gy >= s.H) break;
    const float *dob = dout + ((((int64_t)b * s.D + z) * s.H + gy) * s.W) * 16;
    const int rowoff = yl * C1L_WP;  // (yl + dy + 1 - 1 ... ) : dy index 0..2 is already in aoff, window row 0 = gy - 1
    for (int x0 = 0; x0 < s.W; x0 += 32) {
      float br[8];
      uint32_t bm[8];
      const bool full = x0 + 32 <= s.W;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int x = x0 + 4 * u + kq;
        const bool bok = full || x < s.W;
        br[u] = dob[(bok ? x : 0) * 16 + l15];
        bm[u] = bok ? 0xffffffffu : 0u;
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int x = min(x0 + 4 * u + kq, s.W - 1);
        const float bv = __uint_as_float(__float_as_uint(br[u]) & bm[u]);
        dbs += bv;
        const float a0 = __uint_as_float(__float_as_uint(tile[aoff[0] + rowoff + x]) & amask[0] & bm[u]);
        const float a1 = __uint_as_float(__float_as_uint(tile[aoff[1] + rowoff + x]) & amask[1] & bm[u]);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bv, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bv, acc[1], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave * 1024 + (16 * h + 4 * kq + r) * 32 + l15] = acc[h][r];
  __syncthreads();
  float *o = slab + (int64_t)chunk * 1024;
  for (int e = threadIdx.x; e < 1024; e += 256) o[e] = ((red[e] + red[1024 + e]) + red[2048 + e]) + red[3072 + e];
  if (dbpart) {
    __syncthreads();
    red[threadIdx.x] = dbs;
    __syncthreads();
    if (threadIdx.x < 16) {
      float sum = 0.f;
      for (int i = 0; i < 16; ++i) sum += red[i * 16 + threadIdx.x];
      dbpart[(int64_t)chunk * 16 + threadIdx.x] = sum;
    }
  }
}

// db[co] = ordered f64 sum of the per-workgroup partial bias gradients [parts][Co]
__global__ void conv3d_db_reduce_kernel(const float *__restrict__ dbpart, float *__restrict__ db, int Co, int parts) {
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

// one workgroup per (tap, co): f64 tree over the per-wave partial slabs (fixed order)
// (param_layout: dW(Co,1,3,3,3) instead of [tap][co]; workgroups >= 27 Co reduce the bias-gradient partials dbpart [parts][Co])
__global__ __launch_bounds__(256) void conv3d_c1_bwd_weight_reduce_kernel(const float *__restrict__ slab,
                                                                          float *__restrict__ dWp, int Co, int parts,
                                                                          int param_layout, const float *__restrict__ dbpart,
                                                                          float *__restrict__ db) {
  const bool bias = (int)blockIdx.x >= 27 * Co;
  const int tap = blockIdx.x / Co, co = blockIdx.x % Co;
  double sum = 0.0;
  if (bias) {
    for (int p = threadIdx.x; p < parts; p += 256) sum += (double)dbpart[(int64_t)p * Co + co];
  } else {
    for (int p = threadIdx.x; p < parts; p += 256) sum += (double)slab[(int64_t)p * 1024 + tap * 32 + co];
  }
  __shared__ double red[256];
  red[threadIdx.x] = sum;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (bias) db[co] = (float)red[0];
    else dWp[param_layout ? co * 27 + tap : tap * Co + co] = (float)red[0];
  }
}

// workgroups (= partial slabs) per channel-tile pair of the brick backward-weight kernel: ~2 per CU in total
int bw_brick_parts(int B, int D, int H, int W, int Ci, int Co) {
  int64_t bricks = (int64_t)B * cdiv(D, BRZ) * cdiv(H, BRY) * cdiv(W, BRX);
  int64_t pairs = cdiv(Ci, 32) * cdiv(Co, 32);
  int64_t parts = cdiv(512, pairs);
  if (parts > bricks) parts = bricks;
  return (int)(parts < 1 ? 1 : parts);
}

int check_shape(int B, int D, int H, int W, int Ci, int Co) {
  SVR_CHECK(B > 0 && D > 0 && H > 0 && W > 0, SVR_E_BADSHAPE, "conv3d: empty volume %dx%dx%dx%d", B, D, H, W);
  SVR_CHECK(Ci >= 1 && Co >= 1, SVR_E_BADSHAPE, "conv3d: Ci=%d Co=%d", Ci, Co);
  return SVR_OK;
}

}  // namespace

namespace svr {
// shared with conv3d_bwdw_bf16.hip: dWp = ordered f64 sum of `parts` slabs [part][tap][tile][32][32]
void conv3d_bwd_weight_reduce_launch(const float *slab, float *dWp, int Ci, int Co, int cit, int cot, int parts,
                                     hipStream_t s, int param_layout, const float *dbpart, float *db, int dbparts) {
  const int per = 27 * cit * cot * 1024;
  const int wblocks = (int)cdiv(per, 64);
  hipLaunchKernelGGL(conv3d_bwd_weight_reduce_kernel, dim3(wblocks + (db ? Co : 0)), dim3(256), 0, s, slab, dWp, Ci, Co, cit, cot,
                     parts, param_layout, wblocks, dbpart, db, dbparts);
}
// shared with stage1.hip: conv_in's dWp[tap][co] / db[co] = ordered f64 sums of per-workgroup slabs [part][32 taps][32] / [part][Co]
void conv3d_c1_wgrad_reduce_launch(const float *slab, float *dWp, int Co, int parts, int param_layout, const float *dbpart,
                                   float *db, hipStream_t s) {
  hipLaunchKernelGGL(conv3d_c1_bwd_weight_reduce_kernel, dim3(27 * Co + (db ? Co : 0)), dim3(256), 0, s, slab, dWp, Co, parts,
                     param_layout, dbpart, db);
}
}  // namespace svr

extern "C" int svr_conv3d_pack_weight(const float *W, float *Wp_fwd, float *Wp_bwd, int32_t Ci, int32_t Co, void *stream) {
  SVR_CHECK(W && (Wp_fwd || Wp_bwd), SVR_E_BADARG, "pack_weight: null pointer");
  int n = Ci * Co * 27;
  hipLaunchKernelGGL(pack_weight_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, W, Wp_fwd, Wp_bwd, Ci, Co);
  return launch_status("pack_weight");
}

extern "C" int svr_conv3d_unpack_wgrad(const float *dWp, float *dW, int32_t Ci, int32_t Co, void *stream) {
  SVR_CHECK(dWp && dW, SVR_E_BADARG, "unpack_wgrad: null pointer");
  int n = Ci * Co * 27;
  hipLaunchKernelGGL(unpack_wgrad_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, dWp, dW, Ci, Co);
  return launch_status("unpack_wgrad");
}

extern "C" int svr_conv3d_k3(const float *in, const float *Wp, const float *bias, float *out, int32_t B, int32_t D,
                             int32_t H, int32_t W, int32_t Ci, int32_t Co, int epilogue, const float *mask, void *stream) {
  if (int rc = check_shape(B, D, H, W, Ci, Co)) return rc;
  SVR_CHECK(in && Wp && out, SVR_E_BADARG, "conv3d: null pointer");
  SVR_CHECK(epilogue >= SVR_EPI_NONE && epilogue <= SVR_EPI_MASK, SVR_E_BADARG, "conv3d: epilogue %d", epilogue);
  SVR_CHECK(!(epilogue == SVR_EPI_BIAS || epilogue == SVR_EPI_BIAS_RELU) || bias, SVR_E_BADARG, "conv3d: epilogue needs bias");
  SVR_CHECK(epilogue != SVR_EPI_MASK || mask, SVR_E_BADARG, "conv3d: epilogue needs mask");
  hipStream_t s = (hipStream_t)stream;
  ConvShape sh{B, D, H, W, Ci, Co};
  const int64_t M = (int64_t)B * D * H * W;
  if (Ci == 1) {
    SVR_CHECK(epilogue != SVR_EPI_MASK, SVR_E_UNSUPPORTED, "conv3d: Ci=1 with mask epilogue");
    const int nbz = (int)cdiv(D, C1B), nby = (int)cdiv(H, C1B), nbx = (int)cdiv(W, C1B);
    const unsigned grid = (unsigned)((int64_t)B * nbz * nby * nbx);
    if (Co == 16) hipLaunchKernelGGL(conv3d_c1_mfma_kernel<16>, dim3(grid), dim3(256), 0, s, in, Wp, bias, out, sh, nbz, nby, nbx, epilogue, (double *)nullptr);
    else if (Co == 32) hipLaunchKernelGGL(conv3d_c1_mfma_kernel<32>, dim3(grid), dim3(256), 0, s, in, Wp, bias, out, sh, nbz, nby, nbx, epilogue, (double *)nullptr);
    else SVR_CHECK(false, SVR_E_UNSUPPORTED, "conv3d: Ci=1 supports Co in {16,32}, got %d", Co);
    return launch_status("conv3d_c1_fwd");
  }
  if (Co == 1) {
    SVR_CHECK(epilogue == SVR_EPI_NONE, SVR_E_UNSUPPORTED, "conv3d: Co=1 supports no epilogue");
    unsigned grid = (unsigned)cdiv(M, 256);
    static const int to1_col = getenv("SVR_TO1_COL") ? atoi(getenv("SVR_TO1_COL")) : 1;   // measurement switch: 0 = one output per thread
    constexpr int ZT = 8;
    const int nzb = (int)cdiv(D, ZT), nxb16 = (int)cdiv(W, 16), nxb32 = (int)cdiv(W, 8);
    if (to1_col && Ci == 16 && (((uintptr_t)in) & 15) == 0)
      hipLaunchKernelGGL((conv3d_to1_col_kernel<16, ZT>), dim3((unsigned)cdiv((int64_t)B * nzb * H * nxb16, 4)), dim3(256), 0, s, in, Wp, out, sh, nxb16, nzb);
    else if (to1_col && Ci == 32 && (((uintptr_t)in) & 15) == 0)
      hipLaunchKernelGGL((conv3d_to1_col_kernel<32, ZT>), dim3((unsigned)cdiv((int64_t)B * nzb * H * nxb32, 4)), dim3(256), 0, s, in, Wp, out, sh, nxb32, nzb);
    else if (Ci == 16) hipLaunchKernelGGL(conv3d_to1_kernel<16>, dim3(grid), dim3(256), 0, s, in, Wp, out, sh);
    else if (Ci == 32) hipLaunchKernelGGL(conv3d_to1_kernel<32>, dim3(grid), dim3(256), 0, s, in, Wp, out, sh);
    else SVR_CHECK(false, SVR_E_UNSUPPORTED, "conv3d: Co=1 supports Ci in {16,32}, got %d", Ci);
    return launch_status("conv3d_to1");
  }
  SVR_CHECK(Ci % 16 == 0 && Co % 4 == 0, SVR_E_UNSUPPORTED, "conv3d: need Ci %% 16 == 0 and Co %% 4 == 0 (Ci=%d Co=%d)", Ci, Co);
  SVR_CHECK((((uintptr_t)in | (uintptr_t)Wp) & 15) == 0, SVR_E_ALIGN, "conv3d: operands must be 16-byte aligned");
  {
    const int nbz = (int)cdiv(D, BRZ), nby = (int)cdiv(H, BRY), nbx = (int)cdiv(W, BRX);
    const unsigned bricks = (unsigned)((int64_t)B * nbz * nby * nbx);
#define LAUNCH_BRICK(CKV, TNV)                                                                                      \
  hipLaunchKernelGGL((conv3d_brick_kernel<CKV, TNV>), dim3(bricks, (unsigned)cdiv(Co, TNV * 32)), dim3(256), 0, s, in, \
                     Wp, bias, out, mask, sh, nbz, nby, nbx, epilogue)
    const int tn = Co <= 32 ? 1 : (Co <= 64 ? 2 : 4);
    if (Ci % 32 == 0) {
      if (tn == 1) LAUNCH_BRICK(32, 1); else if (tn == 2) LAUNCH_BRICK(32, 2); else LAUNCH_BRICK(32, 4);
    } else {
      if (tn == 1) LAUNCH_BRICK(16, 1); else if (tn == 2) LAUNCH_BRICK(16, 2); else LAUNCH_BRICK(16, 4);
    }
#undef LAUNCH_BRICK
  }
  return launch_status("conv3d_brick");
}

extern "C" int64_t svr_conv3d_k3_bwd_weight_workspace(int32_t B, int32_t D, int32_t H, int32_t W, int32_t Ci, int32_t Co) {
  int64_t nrows = (int64_t)B * D * H;
  int64_t chunks = cdiv(nrows, BW_ROWS);
  int64_t M = nrows * W;
  int64_t cs = colsum_workspace_floats(M, Co);
  if (Ci == 1) {  // slabs of the widest variant: one per 16-row y block (conv3d_c1_bwd_weight_co16_lds_kernel)
    const int64_t parts = std::max<int64_t>(chunks, (int64_t)B * D * cdiv(H, C1L_ROWS));
    return (parts * 1024 + (cs > parts * 16 ? cs : parts * 16)) * (int64_t)sizeof(float);
  }
  int64_t tiles = cdiv(Ci, 32) * cdiv(Co, 32);
  return ((int64_t)bw_brick_parts(B, D, H, W, Ci, Co) * 27 * tiles * 1024 + cs) * (int64_t)sizeof(float);
}

extern "C" int svr_conv3d_k3_bwd_weight(const float *in, const float *dout, float *dWp, float *db, int32_t B, int32_t D,
                                        int32_t H, int32_t W, int32_t Ci, int32_t Co, void *workspace, void *stream) {
  if (int rc = check_shape(B, D, H, W, Ci, Co)) return rc;
  SVR_CHECK(in && dout && dWp && workspace, SVR_E_BADARG, "conv3d_bwd_weight: null pointer");
  hipStream_t s = (hipStream_t)stream;
  ConvShape sh{B, D, H, W, Ci, Co};
  const int64_t nrows = (int64_t)B * D * H;
  const int chunks = (int)cdiv(nrows, BW_ROWS);
  float *slab = (float *)workspace;
  int64_t slab_floats;
  if (Ci == 1) {
    SVR_CHECK(Co <= 32, SVR_E_UNSUPPORTED, "conv3d_bwd_weight: Ci=1 needs Co<=32 (got %d)", Co);
    int parts = chunks;
    if (Co == 16 && W <= 128) {
      const int nyb = (int)cdiv(H, C1L_ROWS);
      parts = B * D * nyb;
      float *dbpart = db ? slab + (int64_t)parts * 1024 : nullptr;
      hipLaunchKernelGGL(conv3d_c1_bwd_weight_co16_lds_kernel, dim3(parts), dim3(256), 0, s, in, dout, slab, sh, nyb, dbpart);
      if (db) {
        hipLaunchKernelGGL(conv3d_db_reduce_kernel, dim3(16), dim3(256), 0, s, dbpart, db, 16, parts);
        db = nullptr;  // done
      }
    } else if (Co == 16) {
      float *dbpart = db ? slab + (int64_t)chunks * 1024 : nullptr;  // the colsum scratch is at least chunks * 16 floats
      hipLaunchKernelGGL(conv3d_c1_bwd_weight_co16_kernel, dim3(chunks), dim3(256), 0, s, in, dout, slab, sh, dbpart);
      if (db) {
        hipLaunchKernelGGL(conv3d_db_reduce_kernel, dim3(16), dim3(256), 0, s, dbpart, db, 16, chunks);
        db = nullptr;  // done
      }
    } else
      hipLaunchKernelGGL(conv3d_c1_bwd_weight_kernel, dim3(chunks), dim3(256), 0, s, in, dout, slab, sh);
    hipLaunchKernelGGL(conv3d_c1_bwd_weight_reduce_kernel, dim3(27 * Co), dim3(256), 0, s, slab, dWp, Co, parts, 0,
                       (const float *)nullptr, (float *)nullptr);
    slab_floats = (int64_t)parts * 1024;
  } else {
    SVR_CHECK(Ci % 4 == 0 && Co % 4 == 0, SVR_E_UNSUPPORTED, "conv3d_bwd_weight: need Ci, Co %% 4 == 0 (Ci=%d Co=%d)", Ci, Co);
    int cit = (int)cdiv(Ci, 32), cot = (int)cdiv(Co, 32);
    int nbz = (int)cdiv(D, BRZ), nby = (int)cdiv(H, BRY), nbx = (int)cdiv(W, BRX);
    int parts = bw_brick_parts(B, D, H, W, Ci, Co);
    dim3 grid((unsigned)parts, (unsigned)(cit * cot));
    hipLaunchKernelGGL(conv3d_bwd_weight_brick_kernel, grid, dim3(256), 0, s, in, dout, slab, sh, nbz, nby, nbx, cot);
    int per = 27 * cit * cot * 1024;
    hipLaunchKernelGGL(conv3d_bwd_weight_reduce_kernel, dim3(cdiv(per, 64)), dim3(256), 0, s, slab, dWp, Ci, Co, cit, cot, parts,
                       0, (int)cdiv(per, 64), (const float *)nullptr, (float *)nullptr, 0);
    slab_floats = (int64_t)parts * per;
  }
  if (db) colsum_launch(dout, Co, db, slab + slab_floats, nrows * W, Co, s);
  return launch_status("conv3d_bwd_weight");
}

extern "C" int64_t svr_conv3d_c1_fwd_stats_workspace(int32_t B, int32_t D, int32_t H, int32_t W, int32_t Co) {
  return (int64_t)B * cdiv(D, C1B) * cdiv(H, C1B) * cdiv(W, C1B) * 2 * Co * (int64_t)sizeof(double) + 256;
}

// conv_in (Ci == 1) forward with the BatchNorm statistics of its output fused in: out = epi(conv(in, Wp) + bias) and
// stats[0:Co] = mean, stats[Co:2Co] = biased variance of `out` over (B,D,H,W), in f64 -- what svr_bn_stats(out) returns,
// without re-reading the 1 GB output.  Wp: packed forward weights [27][1][Co]; Co in {16, 32}.
extern "C" int svr_conv3d_c1_fwd_stats(const float *in, const float *Wp, const float *bias, float *out, double *stats,
                                       int32_t B, int32_t D, int32_t H, int32_t W, int32_t Co, int epilogue, void *workspace,
                                       void *stream) {
  SVR_CHECK(in && Wp && out && stats && workspace, SVR_E_BADARG, "conv3d_c1_fwd_stats: null pointer");
  SVR_CHECK(B > 0 && D > 0 && H > 0 && W > 0, SVR_E_BADSHAPE, "conv3d_c1_fwd_stats: empty volume");
  SVR_CHECK(epilogue == SVR_EPI_NONE || ((epilogue == SVR_EPI_BIAS || epilogue == SVR_EPI_BIAS_RELU) && bias), SVR_E_BADARG,
            "conv3d_c1_fwd_stats: epilogue %d", epilogue);
  hipStream_t s = (hipStream_t)stream;
  ConvShape sh{B, D, H, W, 1, Co};
  const int nbz = (int)cdiv(D, C1B), nby = (int)cdiv(H, C1B), nbx = (int)cdiv(W, C1B);
  const unsigned grid = (unsigned)((int64_t)B * nbz * nby * nbx);
  double *part = (double *)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  if (Co == 16) hipLaunchKernelGGL(conv3d_c1_mfma_kernel<16>, dim3(grid), dim3(256), 0, s, in, Wp, bias, out, sh, nbz, nby, nbx, epilogue, part);
  else if (Co == 32) hipLaunchKernelGGL(conv3d_c1_mfma_kernel<32>, dim3(grid), dim3(256), 0, s, in, Wp, bias, out, sh, nbz, nby, nbx, epilogue, part);
  else SVR_CHECK(false, SVR_E_UNSUPPORTED, "conv3d_c1_fwd_stats: Co in {16,32}, got %d", Co);
  bn_stats_final_launch(part, stats, (int64_t)B * D * H * W, Co, (int)grid, s);
  return launch_status("conv3d_c1_fwd_stats");
}
