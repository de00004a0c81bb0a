// LDS-tiled exact-f32 MFMA GEMM core for gfx950 (v_mfma_f32_32x32x2_f32, 64-lane waves).
//
// C[BM x BN] += A[BM x K] * B[K x BN], 256 threads = 4 waves arranged WR x WC, every wave
// owning TM x TN tiles of 32x32.  K is consumed in steps of BK = 16 through a double
// buffered LDS image  As[buf][k][m], Bs[buf][k][n]  (k-major, so the MFMA fragment read
// "lane l -> A[i = l&31][k = l>>5]" is 32 consecutive floats per half wave: conflict free).
// Global loads for step t+1 are issued before the MFMAs of step t and written to the other
// LDS buffer afterwards: one barrier per step.
//
// Loaders are small structs: load(kt) pulls this thread's slice of k-step kt into
// registers, store(tile) writes it into one LDS tile.  Two shapes cover every operand on the
// path:  RowK (a row-major [rows][K] operand, K contiguous: activations X, weights W[N][K],
// the conv input gathered per output voxel) and KRow ([K][cols], cols contiguous: packed
// conv weights, the two operands of a weight-gradient product).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace svr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int BK = 16;

template <int WR_, int WC_, int TM_, int TN_>
struct TileCfg {
  static constexpr int WR = WR_, WC = WC_, TM = TM_, TN = TN_;
  static constexpr int BM = WR * TM * 32, BN = WC * TN * 32;
  static_assert(WR * WC == 4, "4 waves per workgroup");
};

// ---- RowK: tile rows x 16 k, source rows are k-contiguous ------------------------------------
// LD must satisfy LD % 8 == 2 so the 4 scalar LDS writes of a float4 are conflict free
// (lane -> (row = lane/4, quad = lane%4): bank = (quad*4+e)*LD + row).
template <int ROWS, class PtrFn>
struct RowKLoader {
  static constexpr int R = (ROWS + 63) / 64;  // float4 per thread
  static constexpr int LD = ROWS + 2;
  PtrFn fn;  // const float* fn(int row_in_tile, int kt)  -> pointer to 16 contiguous floats or nullptr
  float4 v[R];
  __device__ __forceinline__ void load(int kt) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < R; ++i) {
      int row = (t >> 2) + 64 * i;
      const float *p = (ROWS % 64 == 0 || row < ROWS) ? fn(row, kt) : nullptr;
      v[i] = p ? *reinterpret_cast<const float4 *>(p + (t & 3) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  __device__ __forceinline__ void store(float *tile) const {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < R; ++i) {
      int row = (t >> 2) + 64 * i;
      if (ROWS % 64 == 0 || row < ROWS) {
        float *d = tile + ((t & 3) * 4) * LD + row;
        d[0] = v[i].x;
        d[LD] = v[i].y;
        d[2 * LD] = v[i].z;
        d[3 * LD] = v[i].w;
      }
    }
  }
};

// ---- KRow: 16 k x COLS, source is [k][col] with col contiguous ------------------------------
template <int COLS, class PtrFn>
struct KRowLoader {
  static constexpr int Q = COLS / 4;                 // float4 per k-row
  static constexpr int R = (16 * Q + 255) / 256;     // float4 per thread
  static constexpr int LD = COLS + 4;
  PtrFn fn;  // const float* fn(int k_in_step, int col4, int kt) -> pointer to 4 floats or nullptr
  float4 v[R];
  __device__ __forceinline__ void load(int kt) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < R; ++i) {
      int idx = t + 256 * i;
      const float *p = (idx < 16 * Q) ? fn(idx / Q, (idx % Q) * 4, kt) : nullptr;
      v[i] = p ? *reinterpret_cast<const float4 *>(p) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  __device__ __forceinline__ void store(float *tile) const {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < R; ++i) {
      int idx = t + 256 * i;
      if (idx < 16 * Q) *reinterpret_cast<float4 *>(tile + (idx / Q) * LD + (idx % Q) * 4) = v[i];
    }
  }
};

template <class Cfg, class AL, class BL>
struct GemmSmem {
  float a[2][BK * AL::LD];
  float b[2][BK * BL::LD];
};

// Runs the k loop [kt0, kt1) and leaves the block tile in acc[TM][TN] (MFMA C layout:
// reg r of lane l = C[32*tile_m + (r&3) + 8*(r>>2) + 4*(l>>5)][32*tile_n + (l&31)]).
template <class Cfg, class AL, class BL>
__device__ __forceinline__ void gemm_mainloop(AL &al, BL &bl, GemmSmem<Cfg, AL, BL> &sm, int kt0, int kt1,
                                              f32x16 (&acc)[Cfg::TM][Cfg::TN]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave / Cfg::WC, wc = wave % Cfg::WC;
  const int l31 = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  if (kt0 >= kt1) return;
  al.load(kt0);
  bl.load(kt0);
  al.store(sm.a[0]);
  bl.store(sm.b[0]);
  __syncthreads();
  int buf = 0;
  for (int kt = kt0; kt < kt1; ++kt) {
    const bool more = kt + 1 < kt1;
    if (more) {
      al.load(kt + 1);
      bl.load(kt + 1);
    }
    const float *as = sm.a[buf] + wr * (Cfg::TM * 32) + l31;
    const float *bs = sm.b[buf] + wc * (Cfg::TN * 32) + l31;
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      float af[Cfg::TM], bf[Cfg::TN];
#pragma unroll
      for (int i = 0; i < Cfg::TM; ++i) af[i] = as[(kk * 2 + lh) * AL::LD + i * 32];
#pragma unroll
      for (int j = 0; j < Cfg::TN; ++j) bf[j] = bs[(kk * 2 + lh) * BL::LD + j * 32];
#pragma unroll
      for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
        for (int j = 0; j < Cfg::TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    if (more) {
      al.store(sm.a[buf ^ 1]);
      bl.store(sm.b[buf ^ 1]);
    }
    __syncthreads();
    buf ^= 1;
  }
}

// Visit every accumulator element of this thread: f(row_in_block, col_in_block, value).
template <class Cfg, class F>
__device__ __forceinline__ void gemm_foreach(f32x16 (&acc)[Cfg::TM][Cfg::TN], F f) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave / Cfg::WC, wc = wave % Cfg::WC;
#pragma unroll
  for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j) {
      const int col = wc * (Cfg::TN * 32) + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wr * (Cfg::TM * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        f(row, col, acc[i][j][r]);
      }
    }
}

}  // namespace svr
