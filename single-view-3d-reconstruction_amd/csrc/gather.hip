// Trilinear multi-level feature gather / scatter for gfx950.
//
// Replaces, for the IF-Net extractor (reference model/ifnet.py:156-197, :93-118):
//   coordinate prep (swap xyz->zyx, x2, 7 axis displacements), the per-level
//   F.grid_sample(bilinear, zeros, align_corners False/True), torch.cat and the reshape.
//
// Build this file with -ffp-contract=off: the source-index arithmetic must round after
// every operation exactly like ATen's grid_sampler_unnormalize
// (torch/include/ATen/native/GridSampler.h:27-36), or floor() flips at cell boundaries and
// the bit-exact corner-index gate fails.  Accumulation order of the 8 corners and the
// (x*y)*z weight products follow grid_sampler_3d's CPU kernel so features match bit for bit.
//
// Layout: volumes channels-last (B,D,H,W,C); one corner = one contiguous C-vector, read as
// float4 per lane; a point's 7*C outputs of one level are contiguous in its feature row.
#include "common.h"
#include "gather_common.h"

namespace {

struct LevelArgs {
  const float *vol;
  float *gvol;
  int C, D, H, W, col;
  const int32_t *order;  // backward: visiting order of this level (or null = natural order)
  const int32_t *items;  // backward: (7*B*N) item ids pn*7+j sorted by base cell (svr_gather_item_order), or null
};

// All levels run in ONE launch: the grid is the concatenation of the per-level block ranges.
struct FusedArgs {
  LevelArgs L[SVR_MAX_LEVELS];
  unsigned block_start[SVR_MAX_LEVELS + 1];
  int n;
  int pad_start;  // forward: columns [pad_start, row_stride) of every row are zero-filled by the last level
  int shared;     // forward: levels with C >= 16 use gather_fwd_shared_body (sizes fit 32-bit element offsets)
  int xcd;        // forward: XCD-aware block order inside every level (block ranges are multiples of 8)
};

template <int C, bool JMAJOR = false>
__device__ __forceinline__ void gather_fwd_body(const LevelArgs L, const float *__restrict__ points,
                                                float *__restrict__ feat, const int32_t *__restrict__ order,
                                                int64_t gid, int64_t total, int N, int row_stride, float disp,
                                                int ac) {
  constexpr int V = (C >= 4) ? C / 4 : 1;
  if (gid >= total) return;
  int q = (int)(gid % V);
  int j;
  int64_t pn;
  if (JMAJOR) {  // wide levels: a wave walks consecutive (Morton-sorted) points for ONE displacement -> shared cache lines
    const int64_t bn = total / (7 * V);
    pn = (gid / V) % bn;
    j = (int)(gid / (V * bn));
  } else {
    j = (int)((gid / V) % 7);
    pn = gid / (7 * V);
  }
  if (order) pn = order[pn];
  int b = (int)(pn / N);
  Corner c = sample_corner(points + pn * 3, j, disp, L.D, L.H, L.W, ac);
  Weights w = corner_weights(c);
  const float *vb = L.vol + (size_t)b * L.D * L.H * L.W * C;
  if constexpr (C >= 4) {
    // All eight corner loads are issued first, unconditionally and from clamped coordinates: a load inside the
    // bounds branch is waited for at the end of that branch, i.e. eight memory latencies in a row.  The sum keeps
    // ATen's order and still SKIPS out-of-range corners (bit-exact with grid_sample's zero padding).
    float4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int z = min(max(w.z0 + (k >> 2), 0), L.D - 1), y = min(max(w.y0 + ((k >> 1) & 1), 0), L.H - 1),
                x = min(max(w.x0 + (k & 1), 0), L.W - 1);
      v[k] = *reinterpret_cast<const float4 *>(vb + (((size_t)z * L.H + y) * L.W + x) * C + q * 4);
    }
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int dz = 0; dz < 2; ++dz)
#pragma unroll
      for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
          int z = w.z0 + dz, y = w.y0 + dy, x = w.x0 + dx;
          if (z >= 0 && z < L.D && y >= 0 && y < L.H && x >= 0 && x < L.W) {
            float wt = (w.wx[dx] * w.wy[dy]) * w.wz[dz];
            const float4 u = v[dz * 4 + dy * 2 + dx];
            acc.x = acc.x + u.x * wt;
            acc.y = acc.y + u.y * wt;
            acc.z = acc.z + u.z * wt;
            acc.w = acc.w + u.w * wt;
          }
        }
    *reinterpret_cast<float4 *>(feat + pn * row_stride + L.col + j * C + q * 4) = acc;
  } else {
    float acc = 0.f;
#pragma unroll
    for (int dz = 0; dz < 2; ++dz)
#pragma unroll
      for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
          int z = w.z0 + dz, y = w.y0 + dy, x = w.x0 + dx;
          if (z >= 0 && z < L.D && y >= 0 && y < L.H && x >= 0 && x < L.W) {
            float wt = (w.wx[dx] * w.wy[dy]) * w.wz[dz];
            acc = acc + vb[(((size_t)z * L.H + y) * L.W + x) * C] * wt;
          }
        }
    feat[pn * row_stride + L.col + j] = acc;
  }
}

// Forward body for C >= 16 with the sample geometry computed ONCE per (point, displacement) item.  The plain body
// above recomputes the unnormalise / floor / clamp / 8-corner bounds and 64-bit address arithmetic in every one of the
// C/4 lanes that share an item (~350 VALU instructions per lane against 8 loads: the kernel was VALU bound, 2.7 ms).
// Here a wave owns 64 consecutive items: phase 1, lane l evaluates item l (corner (0,0,0) element offset, validity
// bits of the 8 corners, the six 1-D weights, the output row offset); phase 2, C/4 iterations of 64/(C/4) items each,
// the owning lane's values are broadcast with ds_bpermute, the 8 float4 corner loads go out together (wave-uniform
// base + 32-bit offsets) and are summed in ATen's order (same arithmetic as above -> still bit-exact).
template <int C, bool JMAJOR>
__device__ __forceinline__ void gather_fwd_shared_body(const LevelArgs L, const float *__restrict__ points,
                                                       float *__restrict__ feat, const int32_t *__restrict__ order,
                                                       int64_t wave_id, int64_t BN, int N, int row_stride, float disp,
                                                       int ac, int pad_start) {
  constexpr int V = C / 4, IPI = 64 / V;  // lanes per item, items per iteration
  const int lane = threadIdx.x & 63;
  const int64_t items = BN * 7, first = wave_id * 64;
  if (first >= items) return;
  // ---- phase 1: one item per lane
  const int64_t item = min(first + lane, items - 1);
  int j;
  int64_t pidx;
  if (JMAJOR) { pidx = item % BN; j = (int)(item / BN); } else { j = (int)(item % 7); pidx = item / 7; }
  const int64_t pn = order ? (int64_t)order[pidx] : pidx;
  const int b = (int)(pn / N);
  const Corner c = sample_corner(points + pn * 3, j, disp, L.D, L.H, L.W, ac);
  const Weights w = corner_weights(c);
  int vmask = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int z = w.z0 + (k >> 2), y = w.y0 + ((k >> 1) & 1), x = w.x0 + (k & 1);
    if (z >= 0 && z < L.D && y >= 0 && y < L.H && x >= 0 && x < L.W) vmask |= 1 << k;
  }
  if (first + lane < items) vmask |= 0x100;  // live
  if (pad_start >= 0 && j == 0) vmask |= 0x200;  // this item also writes its row's padding columns
  // element offset of corner (0,0,0); garbage when that corner is out of range (never used then: every valid corner's
  // own offset is in range, and int32 wrap-around cannot happen for in-range corners -- host-checked sizes)
  const int ebase = (int)((((int64_t)b * L.D + w.z0) * L.H + w.y0) * L.W * C + (int64_t)w.x0 * C);  // wraps only for invalid corners
  const int rowoff = (int)(pn * row_stride) + L.col + j * C;
  // ---- phase 2
  const int q4 = (lane % V) * 4;
  const int cz = L.H * L.W * C, cy = L.W * C;
  struct Iter {
    float4 v[8];
    int m, ro;
    float wx0, wx1, wy0, wy1, wz0, wz1;
  };
  auto fetch = [&](Iter &I, int it) {  // broadcast the owning lane's geometry, issue the 8 corner loads
    const int src = (it * IPI + lane / V) << 2;  // byte index for ds_bpermute
    const int e = __builtin_amdgcn_ds_bpermute(src, ebase);
    I.m = __builtin_amdgcn_ds_bpermute(src, vmask);
    I.ro = __builtin_amdgcn_ds_bpermute(src, rowoff);
    I.wx0 = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(w.wx[0])));
    I.wx1 = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(w.wx[1])));
    I.wy0 = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(w.wy[0])));
    I.wy1 = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(w.wy[1])));
    I.wz0 = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(w.wz[0])));
    I.wz1 = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(w.wz[1])));
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int off = e + (k >> 2) * cz + ((k >> 1) & 1) * cy + (k & 1) * C + q4;
      I.v[k] = *reinterpret_cast<const float4 *>(L.vol + (((I.m >> k) & 1) ? off : q4));
    }
  };
  auto finish = [&](const Iter &I) {  // sum in ATen's corner order (out-of-range corners skipped), store
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if ((I.m >> k) & 1) {
        const float wt = (((k & 1) ? I.wx1 : I.wx0) * (((k >> 1) & 1) ? I.wy1 : I.wy0)) * ((k >> 2) ? I.wz1 : I.wz0);
        acc.x = acc.x + I.v[k].x * wt;
        acc.y = acc.y + I.v[k].y * wt;
        acc.z = acc.z + I.v[k].z * wt;
        acc.w = acc.w + I.v[k].w * wt;
      }
    }
    if (I.m & 0x100) {
#ifndef SVR_GATHER_NO_NT
      // streaming stores: the rows (4.2 GB at config 3) are never read back by this kernel, and as ordinary stores they pushed the
      // volumes' lines out of L2 -- 2.16 -> 1.90 ms stand-alone (tools/exp/bench_fc0.py)
      float *o = feat + I.ro + q4;
      __builtin_nontemporal_store(acc.x, o); __builtin_nontemporal_store(acc.y, o + 1);
      __builtin_nontemporal_store(acc.z, o + 2); __builtin_nontemporal_store(acc.w, o + 3);
#else
      *reinterpret_cast<float4 *>(feat + I.ro + q4) = acc;
#endif
      if ((I.m & 0x200) && q4 == 0) {
        float *row = feat + (I.ro - L.col);  // j == 0: ro = row start + L.col
        for (int cc = pad_start; cc < row_stride; ++cc) row[cc] = 0.f;
      }
    }
  };
  // two iterations in flight: the loads of it+1 are issued before the sums of it (V is even)
  Iter A, B;
  fetch(A, 0);
#pragma unroll 1
  for (int it = 0; it < V; it += 2) {
    fetch(B, it + 1);
    __builtin_amdgcn_sched_barrier(0);
    finish(A);
    if (it + 2 < V) fetch(A, it + 2);
    __builtin_amdgcn_sched_barrier(0);
    finish(B);
  }
}

__global__ __launch_bounds__(256) void gather_fwd_fused_kernel(FusedArgs A, const float *__restrict__ points,
                                                               float *__restrict__ feat,
                                                               const int32_t *__restrict__ order, int64_t BN, int N,
                                                               int row_stride, float disp, int ac) {
  int l = 0;
  while (l + 1 < A.n && blockIdx.x >= A.block_start[l + 1]) ++l;
  const LevelArgs L = A.L[l];  // by value: read from the kernel arguments once, then lives in SGPRs
  // XCD-aware order inside a level: workgroups are dealt round-robin over the 8 XCDs (each with its own L2), so
  // consecutive workgroups -- consecutive Morton-sorted points, i.e. the same corner voxels -- would pull the same lines
  // into all 8 L2s.  Every level's block range starts at a multiple of 8 and is padded to one (host), and block i of the
  // level works on logical block (i % 8) * (n / 8) + i / 8: one XCD walks one contiguous eighth of the level's items.
  const unsigned nlev = A.block_start[l + 1] - A.block_start[l], li = blockIdx.x - A.block_start[l];
  const unsigned lblock = A.xcd ? (li % svr::kXcds) * (nlev / svr::kXcds) + li / svr::kXcds : li;
  if (A.shared && L.C >= 16) {  // one wave = 64 items
    const int64_t wave_id = (int64_t)lblock * 4 + (threadIdx.x >> 6);
    const int pad = (l == A.n - 1) ? A.pad_start : -1;
    switch (L.C) {
      case 16: gather_fwd_shared_body<16, false>(L, points, feat, order, wave_id, BN, N, row_stride, disp, ac, pad); break;
      case 32: gather_fwd_shared_body<32, false>(L, points, feat, order, wave_id, BN, N, row_stride, disp, ac, pad); break;
      case 64: gather_fwd_shared_body<64, true>(L, points, feat, order, wave_id, BN, N, row_stride, disp, ac, pad); break;
      case 128: gather_fwd_shared_body<128, true>(L, points, feat, order, wave_id, BN, N, row_stride, disp, ac, pad); break;
    }
    return;
  }
  const int64_t gid = (int64_t)lblock * 256 + threadIdx.x;
  const int V = L.C >= 4 ? L.C / 4 : 1;
  const int64_t total = BN * 7 * V;
  if (l == A.n - 1 && gid < total && gid % (7 * V) == 0) {  // padding columns: written once per row
    float *row = feat + (gid / (7 * V)) * row_stride;
    for (int c = A.pad_start; c < row_stride; ++c) row[c] = 0.f;
  }
  switch (L.C) {
    case 1: gather_fwd_body<1>(L, points, feat, order, gid, total, N, row_stride, disp, ac); break;
    case 16: gather_fwd_body<16>(L, points, feat, order, gid, total, N, row_stride, disp, ac); break;
    case 32: gather_fwd_body<32>(L, points, feat, order, gid, total, N, row_stride, disp, ac); break;
    case 64: gather_fwd_body<64, true>(L, points, feat, order, gid, total, N, row_stride, disp, ac); break;
    case 128: gather_fwd_body<128, true>(L, points, feat, order, gid, total, N, row_stride, disp, ac); break;
  }
}

// Backward: scatter-add into gvol (f32 atomics) and, optionally, the gradient wrt the points.
template <int C, bool GVOL, bool GPTS>
__global__ __launch_bounds__(256) void gather_bwd_kernel(LevelArgs L, const float *__restrict__ points,
                                                         const float *__restrict__ gfeat,
                                                         float *__restrict__ gpoints, int64_t total,
                                                         int N, int row_stride, float disp, int ac) {
  constexpr int V = (C >= 4) ? C / 4 : 1;
  int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool live = gid < total;
  if (!live) gid = total - 1;  // keep the wave converged for the shuffles below
  int q = (int)(gid % V);
  int j = (int)((gid / V) % 7);
  int64_t pn = gid / (7 * V);
  int b = (int)(pn / N);
  Corner c = sample_corner(points + pn * 3, j, disp, L.D, L.H, L.W, ac);
  Weights w = corner_weights(c);
  size_t vbase = (size_t)b * L.D * L.H * L.W * C;
  float g[4] = {0.f, 0.f, 0.f, 0.f};
  if constexpr (C >= 4) {
    float4 t = *reinterpret_cast<const float4 *>(gfeat + pn * row_stride + L.col + j * C + q * 4);
    g[0] = t.x; g[1] = t.y; g[2] = t.z; g[3] = t.w;
  } else {
    g[0] = gfeat[pn * row_stride + L.col + j];
  }
  float gix = 0.f, giy = 0.f, giz = 0.f;
#pragma unroll
  for (int dz = 0; dz < 2; ++dz)
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        int z = w.z0 + dz, y = w.y0 + dy, x = w.x0 + dx;
        if (z >= 0 && z < L.D && y >= 0 && y < L.H && x >= 0 && x < L.W) {
          size_t off = vbase + (((size_t)z * L.H + y) * L.W + x) * C + (C >= 4 ? q * 4 : 0);
          if constexpr (GVOL) {
            if (live) {
              float wt = (w.wx[dx] * w.wy[dy]) * w.wz[dz];
#pragma unroll
              for (int i = 0; i < (C >= 4 ? 4 : 1); ++i) atomicAdd(L.gvol + off + i, g[i] * wt);
            }
          }
          if constexpr (GPTS) {
            float dot = 0.f;
#pragma unroll
            for (int i = 0; i < (C >= 4 ? 4 : 1); ++i) dot += L.vol[off + i] * g[i];
            float sx = dx ? 1.f : -1.f, sy = dy ? 1.f : -1.f, sz = dz ? 1.f : -1.f;
            gix += sx * w.wy[dy] * w.wz[dz] * dot;
            giy += sy * w.wx[dx] * w.wz[dz] * dot;
            giz += sz * w.wx[dx] * w.wy[dy] * dot;
          }
        }
      }
  if constexpr (GPTS) {
#pragma unroll
    for (int s = 1; s < V; s <<= 1) {
      gix += __shfl_xor(gix, s);
      giy += __shfl_xor(giy, s);
      giz += __shfl_xor(giz, s);
    }
    if (live && q == 0) {
      float mx = ac ? (float)(L.W - 1) / 2.f : (float)L.W / 2.f;
      float my = ac ? (float)(L.H - 1) / 2.f : (float)L.H / 2.f;
      float mz = ac ? (float)(L.D - 1) / 2.f : (float)L.D / 2.f;
      // grid x <- 2*pt[2], y <- 2*pt[1], z <- 2*pt[0]   (model/ifnet.py:157)
      atomicAdd(gpoints + pn * 3 + 2, 2.f * mx * gix);
      atomicAdd(gpoints + pn * 3 + 1, 2.f * my * giy);
      atomicAdd(gpoints + pn * 3 + 0, 2.f * mz * giz);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Backward scatter, production path: run-combining + contiguous global atomics.
//
// Global f32 atomics run at ~1.3 TB/s chip-wide only when a wave-instruction adds to CONTIGUOUS
// dwords (MI355X_MICROARCH.md "Global float atomics"); the coarse levels would need 11.5 GB of
// them per level and step.  (LDS-privatised tiles were tried first and rejected: ds_add_f32
// retires ~0.4 lanes/clk/CU on gfx950, 12 ms per coarse level -- profiles/r01_notes.md.)
// Instead the points are visited in Morton order (sort.hip), so consecutive samples of one
// displacement j mostly share their 8 corners at the coarse levels: a lane group keeps the 8
// corner sums of its current run in registers and only flushes them (one 64..256-B contiguous
// atomic per corner) when the base voxel changes.  lane = (run group, channel); the sample's
// source index is uniform within a group.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float corner_w(const Weights &w, int k) {
  return (w.wx[k & 1] * w.wy[(k >> 1) & 1]) * w.wz[k >> 2];
}

// Work item (one wave) = (sample b, displacement j, chunk of G*PG consecutive points, 64-channel group).
// The PG = 2*CW samples of a lane group are evaluated ONCE, two per lane (source index, base voxel,
// fractional weights), and broadcast to the group with wave shuffles while it walks them in order;
// the only per-iteration memory access is the lane's own gradient element (prefetched 4 ahead).
template <int C>
__device__ __forceinline__ void gather_bwd_runs_body(const LevelArgs L, const float *__restrict__ points,
                                                     const float *__restrict__ gfeat,
                                                     const int32_t *__restrict__ order, int B, int N,
                                                     int row_stride, float disp, int ac, int64_t item,
                                                     int64_t waves) {
  constexpr int CW = C < 64 ? C : 64;  // channels per lane group
  constexpr int G = 64 / CW;           // independent runs per wave
  constexpr int CG = C / CW;           // channel groups per sample
  constexpr int PG = 2 * CW;           // consecutive points per run group
  if (item >= waves) return;
  const int lane = threadIdx.x & 63;
  const int ch = lane % CW, grp = lane / CW;
  // item -> (j, b, chunk, cg): j outermost, chunks through a multiplicative permutation so that waves running
  // at the same time flush to distant voxel rows (same-address float atomics serialise at the memory side)
  const int64_t cps = (N + G * PG - 1) / (G * PG);  // chunks per sample
  const int64_t nchunks = cps * B;
  const int cg = (int)(item % CG);
  const int64_t cidx = (item / CG) % nchunks;
  const int j = (int)(item / (CG * nchunks));
  const int64_t cperm = (cidx * 1000003LL) % nchunks;
  const int b = (int)(cperm / cps);
  const int n0 = (int)(cperm % cps) * (G * PG) + grp * PG;
  const int cnt = min(PG, N - n0);  // samples of this group (<= 0: none)
  const int coff = L.col + j * C + cg * CW + ch;
  const int64_t vol = (int64_t)L.D * L.H * L.W;
  float *gb = L.gvol + ((size_t)b * vol) * C + cg * CW + ch;

  // ---- stage 1: two samples per lane
  int key[2];
  float fx[2], fy[2], fz[2];
  int64_t pnv[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int n = n0 + s * CW + ch;
    key[s] = -1;
    fx[s] = fy[s] = fz[s] = 0.f;
    pnv[s] = 0;
    if (s * CW + ch < cnt) {
      int64_t pn = (int64_t)b * N + n;
      if (order) pn = order[pn];
      pnv[s] = pn;
      const float p3[3] = {points[pn * 3], points[pn * 3 + 1], points[pn * 3 + 2]};
      Corner c = sample_corner(p3, j, disp, L.D, L.H, L.W, ac);
      const int x0 = clamp_int(c.x0f), y0 = clamp_int(c.y0f), z0 = clamp_int(c.z0f);
      if (z0 >= -1 && z0 < L.D && y0 >= -1 && y0 < L.H && x0 >= -1 && x0 < L.W) {
        key[s] = (x0 + 1) | ((y0 + 1) << 10) | ((z0 + 1) << 20);
        fx[s] = c.ix - c.x0f;
        fy[s] = c.iy - c.y0f;
        fz[s] = c.iz - c.z0f;
      }
    }
  }

  // ---- stage 2: walk the samples with TWO open runs (slot 0 = most recent base voxel, slot 1 = the one
  // before).  A displaced sample's base alternates between two neighbouring voxels while the points stay
  // inside one voxel of the visiting order, so a single open run would be cut every other sample.
  float acc0[8], acc1[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc0[k] = acc1[k] = 0.f;
  int cur0 = -1, cur1 = -1;
  // flush the corners of a run whose bit is NOT set in `skip`
  auto flush = [&](int cur, const float (&acc)[8], int skip) {
    if (cur >= 0) {
      const int x0 = (cur & 1023) - 1, y0 = ((cur >> 10) & 1023) - 1, z0 = (cur >> 20) - 1;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int z = z0 + (k >> 2), y = y0 + ((k >> 1) & 1), x = x0 + (k & 1);
        if (!((skip >> k) & 1) && z >= 0 && z < L.D && y >= 0 && y < L.H && x >= 0 && x < L.W)
          atomicAdd(gb + (((size_t)z * L.H + y) * L.W + x) * C, acc[k]);
      }
    }
  };
  // Retire run `old`.  If the run that stays open (`surv`) is its face neighbour (base voxels one apart in x, y or
  // z -- consecutive cells of the voxel order, or the two bases a displaced sample alternates between), the four
  // corner voxels of the shared face are handed over in registers and only the other four go out as atomics.
  auto retire = [&]() {  // retires slot 1 (cur1 / acc1); slot 0 stays open
    int skip = 0;
    if (cur1 >= 0 && cur0 >= 0) {
      const int d = cur0 - cur1;
      if (d == 1) { acc0[0] += acc1[1]; acc0[2] += acc1[3]; acc0[4] += acc1[5]; acc0[6] += acc1[7]; skip = 0xAA; }
      else if (d == -1) { acc0[1] += acc1[0]; acc0[3] += acc1[2]; acc0[5] += acc1[4]; acc0[7] += acc1[6]; skip = 0x55; }
      else if (d == 1024) { acc0[0] += acc1[2]; acc0[1] += acc1[3]; acc0[4] += acc1[6]; acc0[5] += acc1[7]; skip = 0xCC; }
      else if (d == -1024) { acc0[2] += acc1[0]; acc0[3] += acc1[1]; acc0[6] += acc1[4]; acc0[7] += acc1[5]; skip = 0x33; }
      else if (d == (1 << 20)) { acc0[0] += acc1[4]; acc0[1] += acc1[5]; acc0[2] += acc1[6]; acc0[3] += acc1[7]; skip = 0xF0; }
      else if (d == -(1 << 20)) { acc0[4] += acc1[0]; acc0[5] += acc1[1]; acc0[6] += acc1[2]; acc0[7] += acc1[3]; skip = 0x0F; }
    }
    flush(cur1, acc1, skip);
  };
  constexpr int UNR = 4;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    for (int tb = 0; tb < CW; tb += UNR) {
      if (s * CW + tb >= cnt) break;  // uniform within the group; other groups keep going
      float gq[UNR];
      int kq[UNR];
      float xq[UNR], yq[UNR], zq[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int src = grp * CW + tb + u;
        kq[u] = __shfl(key[s], src);
        xq[u] = __shfl(fx[s], src);
        yq[u] = __shfl(fy[s], src);
        zq[u] = __shfl(fz[s], src);
        int64_t pn;
        if (order) {  // wave-uniform
          pn = __shfl((int)pnv[s], src);
        } else {
          pn = (int64_t)b * N + n0 + s * CW + tb + u;
        }
        const bool live = s * CW + tb + u < cnt;
        if (!live) pn = (int64_t)b * N;  // any valid row: the value is discarded (unconditional load, no branch)
        const float gload = gfeat[pn * row_stride + coff];
        gq[u] = live ? gload : 0.f;
        if (!live) kq[u] = -1;
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int kk = kq[u];
        if (kk < 0) continue;  // sample touches no voxel (or padding slot)
        if (kk != cur0) {
          if (kk == cur1) {  // hit on the older run: make it the most recent
#pragma unroll
            for (int k = 0; k < 8; ++k) { const float t = acc0[k]; acc0[k] = acc1[k]; acc1[k] = t; }
            cur1 = cur0;
          } else {           // miss: retire the older run, age the recent one, open a new one
            retire();
#pragma unroll
            for (int k = 0; k < 8; ++k) { acc1[k] = acc0[k]; acc0[k] = 0.f; }
            cur1 = cur0;
          }
          cur0 = kk;
        }
        const float wx1 = xq[u], wx0 = 1.f - wx1, wy1 = yq[u], wy0 = 1.f - wy1, wz1 = zq[u], wz0 = 1.f - wz1;
        const float g = gq[u];
        const float a00 = wy0 * wz0 * g, a10 = wy1 * wz0 * g, a01 = wy0 * wz1 * g, a11 = wy1 * wz1 * g;
        acc0[0] += wx0 * a00; acc0[1] += wx1 * a00;
        acc0[2] += wx0 * a10; acc0[3] += wx1 * a10;
        acc0[4] += wx0 * a01; acc0[5] += wx1 * a01;
        acc0[6] += wx0 * a11; acc0[7] += wx1 * a11;
      }
    }
  }
  retire();
  flush(cur0, acc0, 0);
}

// Same run-combining scatter, walking ITEMS (point, displacement j) in the order of svr_gather_item_order: all 7*N items
// of a sample sorted jointly by base cell, so a coarse level's cell holds 7x longer runs than under a per-displacement
// point order (level 5: ~680 consecutive items per cell), and a lane group walks REPS * PG consecutive items with its
// two runs kept open across the repetitions: the atomics issued for the 128-channel levels drop from ~0.8 GB to ~0.15 GB
// and the kernel is bound by reading the gradient rows (1.43 GB per 128-channel level).
constexpr int kItemReps = 2;  // (4: a third of the workgroups ran in a mostly empty second round; 2: 18.25 -> 18.1 ms/step)
// the projected scatter (one wave per 128 items and repetition): shorter walks -- 2.7 rounds of workgroups instead of 1.3
// with a two-thirds empty second one
constexpr int kProjReps = 2;

// CPL = channels per lane: 2 for C >= 32 (lane l owns channels l and l + CW: half the lanes per item, so the geometry
// broadcast, the run logic and the shuffles are paid once per two channels: levels 3/4/5 0.45/0.53/0.51 -> 0.36/0.43/0.38 ms
// at config 3)
// PROJ (backward-only projection of a wide level, see gather_bwd_proj_kernel): the source row is the item's dh0 row (C = 256
// values at pn * row_stride, the same for all 7 displacements), the destination is dP[b][voxel][j][C] and a run is one
// (sample, displacement, cell): the displacement rides in the low 3 bits of the run's sample id.
template <int C, int CPL, bool PROJ = false, int REPS = kItemReps, bool STORE = false>
__device__ __forceinline__ void gather_bwd_items_body(const LevelArgs L, const float *__restrict__ points,
                                                      const float *__restrict__ gfeat, int64_t T, int N, int row_stride,
                                                      float disp, int ac, int64_t witem, int64_t waves) {
  constexpr int CW = (C / CPL) < 64 ? (C / CPL) : 64;  // lanes per group; lane l owns channels l, l + CW, ...
  constexpr int G = 64 / CW;           // independent runs per wave
  constexpr int CG = C / (CW * CPL);   // channel groups per sample
  constexpr int PG = 2 * CW;           // items per repetition and run group
  if (witem >= waves) return;
  const int lane = threadIdx.x & 63;
  const int ch = lane % CW, grp = lane / CW;
  const int cg = (int)(witem % CG);
  const int64_t nchunks = waves / CG;
  const int64_t cperm = ((witem / CG) * 1000003LL) % nchunks;  // spread concurrently running waves over distant cells
  const int64_t base_i = (cperm * G + grp) * (int64_t)(PG * REPS);
  const int64_t vol = (int64_t)L.D * L.H * L.W;
  float *gl = L.gvol + cg * CW * CPL + ch;

  float acc0[8 * CPL], acc1[8 * CPL];   // [corner][channel of the lane]
#pragma unroll
  for (int k = 0; k < 8 * CPL; ++k) acc0[k] = acc1[k] = 0.f;
  int cur0 = -1, cur1 = -1, b0 = 0, b1 = 0;  // open runs: base-voxel key and sample
  int slot0 = 0;                             // STORE: slot of the open run in the partial-sum buffer (L.gvol)
  auto flush = [&](int cur, int bb, const float (&acc)[8 * CPL], int skip) {
    if constexpr (STORE) {
      // two-pass projected scatter: the run's 8 corner sums go to partials[slot][corner][C] with plain coalesced stores
      // (all 8: corners outside the volume are never read back); proj_combine_kernel sums the <= 8 cells of a voxel
      if (cur >= 0) {
        float *pb = L.gvol + (size_t)slot0 * 8 * C + cg * CW * CPL + ch;
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
          for (int p = 0; p < CPL; ++p) pb[k * C + p * CW] = acc[k * CPL + p];
      }
    } else if (cur >= 0) {
      const int x0 = (cur & 1023) - 1, y0 = ((cur >> 10) & 1023) - 1, z0 = (cur >> 20) - 1;
      constexpr size_t VS = PROJ ? (size_t)7 * C : (size_t)C;   // floats per voxel of the destination
      float *gb = PROJ ? gl + (size_t)(bb >> 3) * vol * VS + (size_t)(bb & 7) * C : gl + (size_t)bb * vol * C;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int z = z0 + (k >> 2), y = y0 + ((k >> 1) & 1), x = x0 + (k & 1);
        if (!((skip >> k) & 1) && z >= 0 && z < L.D && y >= 0 && y < L.H && x >= 0 && x < L.W) {
#pragma unroll
          for (int p = 0; p < CPL; ++p) atomicAdd(gb + (((size_t)z * L.H + y) * L.W + x) * VS + p * CW, acc[k * CPL + p]);
        }
      }
    }
  };
  auto retire = [&]() {  // retires slot 1; hands the shared face over to slot 0 when the two are face neighbours
    int skip = 0;
    if (cur1 >= 0 && cur0 >= 0 && b0 == b1) {
      const int d = cur0 - cur1;
      // acc0[to] += acc1[from] for the four corners of the shared face (compile-time indices: a lambda taking them as
      // arguments made the accumulators dynamically indexed -> 144 B/lane of scratch)
#define SVR_HAND(T0, F0, T1, F1, T2, F2, T3, F3)                                                     \
  _Pragma("unroll") for (int p = 0; p < CPL; ++p) {                                                  \
    acc0[T0 * CPL + p] += acc1[F0 * CPL + p]; acc0[T1 * CPL + p] += acc1[F1 * CPL + p];              \
    acc0[T2 * CPL + p] += acc1[F2 * CPL + p]; acc0[T3 * CPL + p] += acc1[F3 * CPL + p];              \
  }
      if (d == 1) { SVR_HAND(0, 1, 2, 3, 4, 5, 6, 7) skip = 0xAA; }
      else if (d == -1) { SVR_HAND(1, 0, 3, 2, 5, 4, 7, 6) skip = 0x55; }
      else if (d == 1024) { SVR_HAND(0, 2, 1, 3, 4, 6, 5, 7) skip = 0xCC; }
      else if (d == -1024) { SVR_HAND(2, 0, 3, 1, 6, 4, 7, 5) skip = 0x33; }
      else if (d == (1 << 20)) { SVR_HAND(0, 4, 1, 5, 2, 6, 3, 7) skip = 0xF0; }
      else if (d == -(1 << 20)) { SVR_HAND(4, 0, 5, 1, 6, 2, 7, 3) skip = 0x0F; }
#undef SVR_HAND
    }
    flush(cur1, b1, acc1, skip);
  };
  constexpr int UNR = 4;   // items per block of loads; two blocks in flight (the dh rows come from the Infinity Cache / HBM: latency bound)
  for (int rep = 0; rep < REPS; ++rep) {
    const int64_t i0 = base_i + (int64_t)rep * PG;
    const int cnt = (int)min((int64_t)PG, T - i0);  // items of this repetition (<= 0: none)
    if (cnt <= 0) break;                            // uniform within the group
    // ---- stage 1: two items per lane
    int key[2], bb[2], goff[2], sid[2];
    float fx[2], fy[2], fz[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      key[s] = -1;
      bb[s] = 0;
      goff[s] = 0;
      sid[s] = (STORE && s * CW + ch < cnt) ? L.order[i0 + s * CW + ch] : 0;   // STORE: run-start count in front of the item
      fx[s] = fy[s] = fz[s] = 0.f;
      if (s * CW + ch < cnt) {
        const int id = L.items[i0 + s * CW + ch];
        const int pn = id / 7, j = id - pn * 7;
        const float p3[3] = {points[(int64_t)pn * 3], points[(int64_t)pn * 3 + 1], points[(int64_t)pn * 3 + 2]};
        Corner c = sample_corner(p3, j, disp, L.D, L.H, L.W, ac);
        const int x0 = clamp_int(c.x0f), y0 = clamp_int(c.y0f), z0 = clamp_int(c.z0f);
        goff[s] = PROJ ? pn * row_stride : pn * row_stride + L.col + j * C;
        bb[s] = PROJ ? (pn / N) * 8 + j : pn / N;
        if (z0 >= -1 && z0 < L.D && y0 >= -1 && y0 < L.H && x0 >= -1 && x0 < L.W) {
          key[s] = (x0 + 1) | ((y0 + 1) << 10) | ((z0 + 1) << 20);
          fx[s] = c.ix - c.x0f;
          fy[s] = c.iy - c.y0f;
          fz[s] = c.iz - c.z0f;
        }
      }
    }
    // ---- stage 2: walk them with the two open runs.  The gradient loads run one block of UNR items AHEAD of the
    // arithmetic (two register sets): with the loads of a block issued and waited for in the same block every 4 items
    // paid a full memory latency and the 128-channel levels ran at 0.5 ms each instead of the 0.29 ms of their reads.
    constexpr int NB = 2 * CW / UNR;
    auto issue = [&](float (&gq)[UNR * CPL], int blk) {
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int t = blk * UNR + u, s = t >= CW ? 1 : 0;
        const int go = __shfl(s ? goff[1] : goff[0], grp * CW + t - s * CW);
#pragma unroll
        for (int p = 0; p < CPL; ++p) gq[u * CPL + p] = gfeat[(t < cnt ? go : 0) + cg * CW * CPL + ch + p * CW];  // unconditional loads
      }
    };
    auto process = [&](const float (&gq)[UNR * CPL], int blk) {
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int t = blk * UNR + u, s = t >= CW ? 1 : 0, src = grp * CW + t - s * CW;
        const int kraw = __shfl(s ? key[1] : key[0], src);
        const int bqu = __shfl(s ? bb[1] : bb[0], src);
        const float xq = __shfl(s ? fx[1] : fx[0], src), yq = __shfl(s ? fy[1] : fy[0], src),
                    zq = __shfl(s ? fz[1] : fz[0], src);
        const int kk = t < cnt ? kraw : -1;
        if (kk < 0) continue;  // item touches no voxel (or padding slot)
        if constexpr (PROJ) {
          // items arrive sorted by (sample, cell, displacement): a run never comes back, one open run is enough (and the
          // next run is another displacement of the same cell -- another slice of dP -- so there is nothing to hand over)
          if (kk != cur0 || bqu != b0) {
            flush(cur0, b0, acc0, 0);
#pragma unroll
            for (int k = 0; k < 8 * CPL; ++k) acc0[k] = 0.f;
            cur0 = kk;
            b0 = bqu;
            if constexpr (STORE) slot0 = __shfl(s ? sid[1] : sid[0], src);
          }
        } else if (kk != cur0 || bqu != b0) {
          if (kk == cur1 && bqu == b1) {  // hit on the older run: make it the most recent
#pragma unroll
            for (int k = 0; k < 8 * CPL; ++k) { const float tt = acc0[k]; acc0[k] = acc1[k]; acc1[k] = tt; }
            cur1 = cur0;
            b1 = b0;
          } else {                        // miss: retire the older run, age the recent one, open a new one
            retire();
#pragma unroll
            for (int k = 0; k < 8 * CPL; ++k) { acc1[k] = acc0[k]; acc0[k] = 0.f; }
            cur1 = cur0;
            b1 = b0;
          }
          cur0 = kk;
          b0 = bqu;
        }
        const float wx1 = xq, wx0 = 1.f - wx1, wy1 = yq, wy0 = 1.f - wy1, wz1 = zq, wz0 = 1.f - wz1;
        const float w00 = wy0 * wz0, w10 = wy1 * wz0, w01 = wy0 * wz1, w11 = wy1 * wz1;
        // the walk is bound by the VALU instructions it issues (one item = 8 corners x CPL channels): explicit FMAs
        // (the file is built with contraction off for the index arithmetic; a scatter has no summation order to keep)
        // and, for an even CPL, channel PAIRS so they become v_pk_mul_f32 / v_pk_fma_f32
        if constexpr (CPL % 2 == 0) {
          typedef float f2 __attribute__((ext_vector_type(2)));
#pragma unroll
          for (int p = 0; p < CPL; p += 2) {
            const f2 g = {gq[u * CPL + p], gq[u * CPL + p + 1]};
            const f2 a00 = g * w00, a10 = g * w10, a01 = g * w01, a11 = g * w11;
#define SVR_ACC2(K, WX, A)                                                                     \
  {                                                                                            \
    const f2 r = __builtin_elementwise_fma(f2{WX, WX}, A, f2{acc0[K * CPL + p], acc0[K * CPL + p + 1]}); \
    acc0[K * CPL + p] = r.x;                                                                   \
    acc0[K * CPL + p + 1] = r.y;                                                               \
  }
            SVR_ACC2(0, wx0, a00) SVR_ACC2(1, wx1, a00) SVR_ACC2(2, wx0, a10) SVR_ACC2(3, wx1, a10)
            SVR_ACC2(4, wx0, a01) SVR_ACC2(5, wx1, a01) SVR_ACC2(6, wx0, a11) SVR_ACC2(7, wx1, a11)
#undef SVR_ACC2
          }
        } else {
#pragma unroll
          for (int p = 0; p < CPL; ++p) {
            const float g = gq[u * CPL + p];
            const float a00 = w00 * g, a10 = w10 * g, a01 = w01 * g, a11 = w11 * g;
            acc0[0 * CPL + p] = __builtin_fmaf(wx0, a00, acc0[0 * CPL + p]); acc0[1 * CPL + p] = __builtin_fmaf(wx1, a00, acc0[1 * CPL + p]);
            acc0[2 * CPL + p] = __builtin_fmaf(wx0, a10, acc0[2 * CPL + p]); acc0[3 * CPL + p] = __builtin_fmaf(wx1, a10, acc0[3 * CPL + p]);
            acc0[4 * CPL + p] = __builtin_fmaf(wx0, a01, acc0[4 * CPL + p]); acc0[5 * CPL + p] = __builtin_fmaf(wx1, a01, acc0[5 * CPL + p]);
            acc0[6 * CPL + p] = __builtin_fmaf(wx0, a11, acc0[6 * CPL + p]); acc0[7 * CPL + p] = __builtin_fmaf(wx1, a11, acc0[7 * CPL + p]);
          }
        }
      }
    };
    float gA[UNR * CPL], gB[UNR * CPL];
    issue(gA, 0);
    for (int blk = 0; blk < NB; blk += 2) {  // all conditions are uniform within the lane group
      if ((blk + 1) * UNR < cnt) issue(gB, blk + 1);
      process(gA, blk);
      if ((blk + 1) * UNR >= cnt) break;
      if ((blk + 2) * UNR < cnt) issue(gA, blk + 2);
      process(gB, blk + 1);
      if ((blk + 2) * UNR >= cnt) break;
    }
  }
  retire();
  flush(cur0, b0, acc0, 0);
}

// channels per lane of the item scatter.  (4 for C >= 64: 186 VGPRs for the whole fused kernel, 2 waves/SIMD, all levels
// slower: 2.81 vs 2.65 ms uniform, 2.19 vs 1.84 surface)
__host__ __device__ inline int items_cpl(int C) { return C >= 32 ? 2 : 1; }

// Backward-only PROJECTION of a wide level l (C_l = 128).  Its feature columns are 35 % of fc_0's K each, and what the
// scatter adds into the level's gradient volume is  dvol_l[v][c] = sum_items w * dfeat[p][(l,j,c)]  with
// dfeat[p][(l,j,c)] = sum_n dh0[p][n] W0[n][(l,j,c)].  The sum over items commutes with the product by W0:
//     dP_l[b][v][j][n] = sum_{items (p,j) -> v} w * dh0[p][n]        (this kernel: a scatter of 256-wide dh0 rows)
//     dvol_l[v][c]     = sum_{j,n} dP_l[v][j][n] W0[n][(l,j,c)]      (a GEMM over voxels: 32 768 x 1792 x 128 at level 4)
//     dW0[n][(l,j,c)]  = sum_v dP_l[v][j][n] vol_l[v][c]             (ditto)
// so neither dX0 nor dW0 of the point MLP has to touch the level's 896 columns over 400 000 points (K 2592 -> 800 with
// both 128-channel levels projected).  The forward pass is unchanged.  Items arrive sorted by (sample, cell, j).
__global__ __launch_bounds__(256) void gather_bwd_proj_kernel(LevelArgs L, const float *__restrict__ points,
                                                              const float *__restrict__ dh, int64_t T, int N, int lddh,
                                                              float disp, int ac, int64_t waves) {
  const int64_t witem = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  gather_bwd_items_body<256, 4, true, kProjReps>(L, points, dh, T, N, lddh, disp, ac, witem, waves);
}

// Two-pass form of the projected scatter (level 4 at config 3: 94 000 runs x 8 KB of run-end float atomics = 0.77 GB at
// ~1.3 TB/s were its floor).  Pass 1 is the same walk with the run sums STORED to partials[slot][corner][256]; pass 2
// (proj_combine_kernel) gives every (voxel, displacement) row of dP the sum of the <= 8 cells that touch it, in a fixed
// order, with plain stores: no atomics, no memset of dP, bit-reproducible.  Slots: a run = maximal stretch of equal
// (sample, cell, displacement) keys inside one wave's 256-item chunk; slot = number of run starts in front of it
// (proj_flag_kernel + exclusive scan), first_slot[key] = slot of the key's first run (proj_table_kernel).
__global__ __launch_bounds__(256) void gather_bwd_proj_store_kernel(LevelArgs L, const float *__restrict__ points,
                                                                    const float *__restrict__ dh, int64_t T, int N, int lddh,
                                                                    float disp, int ac, int64_t waves) {
  const int64_t witem = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  gather_bwd_items_body<256, 4, true, kProjReps, true>(L, points, dh, T, N, lddh, disp, ac, witem, waves);
}

__global__ __launch_bounds__(256) void proj_flag_kernel(const uint32_t *__restrict__ keys, int32_t *__restrict__ flags, int64_t T,
                                                        int chunk, uint32_t sentinel) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i > T) return;
  flags[i] = (i < T && keys[i] != sentinel && (i % chunk == 0 || keys[i] != keys[i - 1])) ? 1 : 0;   // flags[T] = 0
}

__global__ __launch_bounds__(256) void proj_table_kernel(const uint32_t *__restrict__ keys, const int32_t *__restrict__ sidx,
                                                         int32_t *__restrict__ first_slot, int64_t nkeys, int64_t T) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (q > nkeys) return;
  int64_t lo = 0, hi = T;  // first item with a key >= q (the items that touch no voxel carry the key nkeys)
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (keys[mid] < (uint32_t)q) lo = mid + 1; else hi = mid;
  }
  first_slot[q] = sidx[lo];
}

// one wave per row (sample, voxel, displacement) of dP: 64 lanes x float4 = 256 channels
__global__ __launch_bounds__(256) void proj_combine_kernel(const float *__restrict__ partials, const int32_t *__restrict__ first_slot,
                                                           float *__restrict__ dP, int B, int D, int H, int W) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t rows = (int64_t)B * D * H * W * 7;
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const int j = (int)(row % 7);
  int64_t v = row / 7;
  const int x = (int)(v % W);
  v /= W;
  const int y = (int)(v % H);
  v /= H;
  const int z = (int)(v % D);
  const int b = (int)(v / D);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    // the cell whose corner k is this voxel: base voxel (z - dz, y - dy, x - dx), lattice coordinate = base + 1
    const int cz = z - (k >> 2) + 1, cy = y - ((k >> 1) & 1) + 1, cx = x - (k & 1) + 1;
    const int64_t q = ((((int64_t)b * (D + 1) + cz) * (H + 1) + cy) * (W + 1) + cx) * 8 + j;
    const int s0 = first_slot[q], s1 = first_slot[q + 1];
    for (int sl = s0; sl < s1; ++sl) {
      const float4 p = *reinterpret_cast<const float4 *>(partials + ((size_t)sl * 8 + k) * 256 + lane * 4);
      acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w;
    }
  }
  *reinterpret_cast<float4 *>(dP + row * 256 + lane * 4) = acc;
}

__host__ __device__ inline int64_t bwd_items_waves_cpl(int C, int cpl, int64_t T, int reps = kItemReps) {
  const int cw = (C / cpl) < 64 ? (C / cpl) : 64;
  const int64_t per = (int64_t)(64 / cw) * 2 * cw * reps;  // items per wave
  return svr::cdiv(T, per) * (C / (cw * cpl));
}
__host__ __device__ inline int64_t bwd_items_waves(int C, int64_t T) { return bwd_items_waves_cpl(C, items_cpl(C), T); }

__host__ __device__ inline int64_t bwd_runs_waves(int C, int B, int N) {
  int cw = C < 64 ? C : 64;
  int64_t per = (int64_t)(64 / cw) * 2 * cw;  // points per wave
  int64_t cps = (N + per - 1) / per;
  return cps * B * 7 * (C / cw);
}

__global__ __launch_bounds__(256) void gather_bwd_fused_kernel(FusedArgs A, const float *__restrict__ points,
                                                               const float *__restrict__ gfeat,
                                                               const int32_t *__restrict__ default_order, int B, int N,
                                                               int row_stride, float disp, int ac) {
  int l = 0;
  while (l + 1 < A.n && blockIdx.x >= A.block_start[l + 1]) ++l;
  const LevelArgs L = A.L[l];  // by value: read from the kernel arguments once, then lives in SGPRs
  const int64_t item = ((int64_t)(blockIdx.x - A.block_start[l]) * 256 + threadIdx.x) >> 6;
  if (L.items) {  // joint item order of this level
    const int64_t T = (int64_t)7 * B * N, iw = bwd_items_waves(L.C, T);
    switch (L.C) {
      case 16: gather_bwd_items_body<16, 1>(L, points, gfeat, T, N, row_stride, disp, ac, item, iw); break;
      case 32: gather_bwd_items_body<32, 2>(L, points, gfeat, T, N, row_stride, disp, ac, item, iw); break;
      case 64: gather_bwd_items_body<64, 2>(L, points, gfeat, T, N, row_stride, disp, ac, item, iw); break;
      case 128: gather_bwd_items_body<128, 2>(L, points, gfeat, T, N, row_stride, disp, ac, item, iw); break;
    }
    return;
  }
  const int64_t waves = bwd_runs_waves(L.C, B, N);
  const int32_t *order = L.order ? L.order : default_order;
  switch (L.C) {
    case 16: gather_bwd_runs_body<16>(L, points, gfeat, order, B, N, row_stride, disp, ac, item, waves); break;
    case 32: gather_bwd_runs_body<32>(L, points, gfeat, order, B, N, row_stride, disp, ac, item, waves); break;
    case 64: gather_bwd_runs_body<64>(L, points, gfeat, order, B, N, row_stride, disp, ac, item, waves); break;
    case 128: gather_bwd_runs_body<128>(L, points, gfeat, order, B, N, row_stride, disp, ac, item, waves); break;
  }
}

__global__ void corner_index_kernel(const float *__restrict__ points, int32_t *__restrict__ out,
                                    int64_t total, int N, int D, int H, int W, float disp, int ac) {
  int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // over (b, j, n)
  if (gid >= total) return;
  int n = (int)(gid % N);
  int j = (int)((gid / N) % 7);
  int64_t b = gid / ((int64_t)7 * N);
  Corner c = sample_corner(points + (b * N + n) * 3, j, disp, D, H, W, ac);
  out[gid * 3 + 0] = clamp_int(c.z0f);
  out[gid * 3 + 1] = clamp_int(c.y0f);
  out[gid * 3 + 2] = clamp_int(c.x0f);
}

// Deterministic scatter (SVR_GATHER_DETERMINISTIC; tests only): no atomics, one thread per (sample, channel) walks
// j = 0..6, n = 0..N-1 and the 8 corners in ATen's order with plain read-modify-writes -- the summation order of
// grid_sampler_3d_backward's CPU kernel (torch/include/ATen/native/GridSampler.h safe_add_3d call sequence), so the
// gradient volumes equal the reference's CPU autograd bit for bit on identical inputs and are reproducible run to run.
__global__ __launch_bounds__(128) void gather_bwd_serial_kernel(LevelArgs L, const float *__restrict__ points,
                                                                const float *__restrict__ gfeat, int N,
                                                                int row_stride, float disp, int ac) {
  const int b = blockIdx.x, c = threadIdx.x;
  if (c >= L.C) return;
  float *gb = L.gvol + (size_t)b * L.D * L.H * L.W * L.C + c;
  for (int j = 0; j < 7; ++j)
    for (int n = 0; n < N; ++n) {
      const int64_t pn = (int64_t)b * N + n;
      const Corner cr = sample_corner(points + pn * 3, j, disp, L.D, L.H, L.W, ac);
      const Weights w = corner_weights(cr);
      const float g = gfeat[pn * row_stride + L.col + j * L.C + c];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int z = w.z0 + (k >> 2), y = w.y0 + ((k >> 1) & 1), x = w.x0 + (k & 1);
        if (z >= 0 && z < L.D && y >= 0 && y < L.H && x >= 0 && x < L.W) {
          volatile float *q = gb + (((size_t)z * L.H + y) * L.W + x) * L.C;
          *q = *q + corner_w(w, k) * g;
        }
      }
    }
}

// ---------------------------------------------------------------------------------------------
// Backward scatter, PULL form (levels with C <= 64, i.e. the sparse ones: 0.02 .. 1.5 samples per base cell and
// displacement, where run-combining cannot help and the atomic kernel above sits at the ~1.3 TB/s float-atomic rate).
//
// Plan (svr_gather_pull_plan, once per step and level, on a side stream): the 7*B*N items (point, displacement j) are
// keyed by the row-major index of their base cell c = floor(source index) + 1 in a (D+1) x (H+1) x (W+1) lattice
// (c = 0 <=> base -1), radix-sorted (stable), and every sorted item gets a 16-byte record {fx, fy, fz, offset of its
// C gradient values}; cs[c] = number of items with a key below c (CSR offsets: a running maximum over the cell ends).
// Kernel: a voxel v receives from the cells c = v + d, d in {0,1}^3, with weight f (d = 0: v is the item's upper
// corner) or 1 - f (d = 1) per axis.  A thread owns XS consecutive x voxels and one float4 of channels: the cells
// x0 .. x0+XS of one (dz, dy) pair are consecutive keys, so their items are ONE contiguous range [cs[k], cs[k+XS+1]):
// four ranges per thread, walked together (one item of each per round, all loads of a round issued first).
// No atomics, no memset of the gradient volume, fixed summation order (sorted order) -> bit-reproducible.
// (First version: one voxel per thread, walks one after the other: 1.2 ms at level 1 = four dependent load chains per
// voxel at 16 waves per CU; the x strips put 4x the work behind every chain and read every item once per row pair.)
// ---------------------------------------------------------------------------------------------
struct PullRec {
  float fx, fy, fz;
  int32_t goff;
};

__device__ __forceinline__ bool pull_cell(const float *pt, int j, float disp, int D, int H, int W, int ac, int b,
                                          uint32_t &key, float &fx, float &fy, float &fz) {
  const Corner c = sample_corner(pt, j, disp, D, H, W, ac);
  const int x0 = clamp_int(c.x0f), y0 = clamp_int(c.y0f), z0 = clamp_int(c.z0f);
  fx = c.ix - c.x0f;
  fy = c.iy - c.y0f;
  fz = c.iz - c.z0f;
  if (!(z0 >= -1 && z0 < D && y0 >= -1 && y0 < H && x0 >= -1 && x0 < W)) return false;  // touches no voxel (or NaN)
  key = (uint32_t)((((int64_t)b * (D + 1) + z0 + 1) * (H + 1) + y0 + 1) * (W + 1) + x0 + 1);
  return true;
}

__global__ __launch_bounds__(256) void pull_key_kernel(const float *__restrict__ points, uint32_t *__restrict__ keys,
                                                       int32_t *__restrict__ vals, int64_t total, int N, int D, int H,
                                                       int W, float disp, int ac, uint32_t sentinel, int with_j) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // item = pn * 7 + j
  if (i >= total) return;
  const int64_t pn = i / 7;
  const int j = (int)(i - pn * 7);
  uint32_t key;
  float fx, fy, fz;
  const float p3[3] = {points[pn * 3], points[pn * 3 + 1], points[pn * 3 + 2]};
  const bool ok = pull_cell(p3, j, disp, D, H, W, ac, (int)(pn / N), key, fx, fy, fz);
  keys[i] = ok ? (with_j ? key * 8u + (uint32_t)j : key) : sentinel;   // with_j: (cell, displacement) order
  vals[i] = (int32_t)i;
}

// sorted item -> record; the last item of a cell writes the cell's end (same arithmetic as the key kernel: both are
// compiled in this file)
__global__ __launch_bounds__(256) void pull_record_kernel(const float *__restrict__ points,
                                                          const uint32_t *__restrict__ keys,
                                                          const int32_t *__restrict__ items, PullRec *__restrict__ recs,
                                                          int32_t *__restrict__ ends, int64_t total, int N, int D, int H,
                                                          int W, int C, int col, int row_stride, float disp, int ac,
                                                          uint32_t sentinel) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const uint32_t k = keys[idx];
  if (k == sentinel) return;
  const int32_t id = items[idx];
  const int32_t pn = id / 7;
  const int j = id - pn * 7;
  const float p3[3] = {points[(int64_t)pn * 3], points[(int64_t)pn * 3 + 1], points[(int64_t)pn * 3 + 2]};
  uint32_t key;
  PullRec r;
  pull_cell(p3, j, disp, D, H, W, ac, pn / N, key, r.fx, r.fy, r.fz);
  r.goff = pn * row_stride + col + j * C;
  recs[idx] = r;
  if (idx == total - 1 || keys[idx + 1] != k) ends[k + 1] = (int32_t)idx + 1;  // ends[0] stays 0
}

// workgroup = BZ x BY x BXG strips of XS voxels x C/4 channel quads (256 threads): the voxels that share an item sit
// in the same workgroup most of the time, so its record and gradient slice are L1 hits after the first touch
template <int C, int XS, int BZ, int BY, int BXG>
__global__ __launch_bounds__(256) void gather_bwd_pull_kernel(float *__restrict__ gvol, const uint32_t *__restrict__ keys,
                                                              const PullRec *__restrict__ recs,
                                                              const int32_t *__restrict__ cs,
                                                              const float *__restrict__ gfeat, int n_items, int B, int D,
                                                              int H, int W) {
  constexpr int V = C / 4;
  static_assert(BZ * BY * BXG * V == 256, "brick must fill the workgroup");
  const int t = threadIdx.x;
  const int q4 = (t % V) * 4, si = t / V;
  const int nbx = (W + BXG * XS - 1) / (BXG * XS), nby = (H + BY - 1) / BY, nbz = (D + BZ - 1) / BZ;
  uint32_t bid = blockIdx.x;
  const int bx = bid % nbx; bid /= nbx;
  const int by = bid % nby; bid /= nby;
  const int bz = bid % nbz;
  const int b = bid / nbz;
  const int x0 = (bx * BXG + si % BXG) * XS, y = by * BY + (si / BXG) % BY, z = bz * BZ + si / (BXG * BY);
  if (x0 >= W || y >= H || z >= D) return;
  const int ncell = min(XS, W - x0) + 1;  // cells x0 .. x0 + ncell - 1 (cell coordinates end at W)
  uint32_t kbase[4];
  int lo[4], hi[4];
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const int dz = w >> 1, dy = w & 1;
    kbase[w] = (uint32_t)((((int64_t)b * (D + 1) + z + dz) * (H + 1) + y + dy) * (W + 1) + x0);
    lo[w] = cs[kbase[w]];
    hi[w] = cs[kbase[w] + ncell];
  }
  float4 acc[XS];
#pragma unroll
  for (int i = 0; i < XS; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  const int last = n_items - 1;
  // one item of each of the four ranges per round; all key / record loads of a round first (unconditional, from clamped
  // indices: a load inside a branch is waited for on its own), then the four gradient loads
  while (lo[0] < hi[0] || lo[1] < hi[1] || lo[2] < hi[2] || lo[3] < hi[3]) {
    uint32_t k[4];
    PullRec r[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int i = min(lo[w], last);
      k[w] = keys[i];
      r[w] = recs[i];
    }
    float4 g[4];
    bool v[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      v[w] = lo[w] < hi[w];
      g[w] = *reinterpret_cast<const float4 *>(gfeat + (v[w] ? r[w].goff : 0) + q4);
    }
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int dz = w >> 1, dy = w & 1;
      if (v[w]) {
        const int o = (int)(k[w] - kbase[w]);  // cell x0 + o: voxel x0 + o gets f, voxel x0 + o - 1 gets 1 - f
        const float wyz = (dy ? 1.f - r[w].fy : r[w].fy) * (dz ? 1.f - r[w].fz : r[w].fz);
        const float wa = r[w].fx * wyz, wb = (1.f - r[w].fx) * wyz;
#pragma unroll
        for (int i = 0; i < XS; ++i) {
          const float wt = (o == i) ? wa : ((o == i + 1) ? wb : 0.f);
          acc[i].x += wt * g[w].x; acc[i].y += wt * g[w].y; acc[i].z += wt * g[w].z; acc[i].w += wt * g[w].w;
        }
        ++lo[w];
      }
    }
  }
  float *out = gvol + ((((int64_t)b * D + z) * H + y) * W + x0) * C + q4;
#pragma unroll
  for (int i = 0; i < XS; ++i)
    if (x0 + i < W) *reinterpret_cast<float4 *>(out + (int64_t)i * C) = acc[i];   // (streaming stores here: no change of the step)
}

// x-strip length of the pull kernel per channel count (measured at config 3: 0.55 / 0.38 / 0.57 ms; strips of 4 / 2 / 2:
// 0.57 / 0.41 / 0.66, one voxel per thread: 1.29 / 0.60 / 0.66)
__host__ __device__ inline int pull_xs(int C) { return C == 16 ? 8 : 4; }

// The pull kernel's cost is set by its LONGEST serial walk: stats[0] = max number of items in one (row, x strip) range,
// stats[1] = number of occupied cells.  Points clustered on surfaces give walks of hundreds of items (level 3 at config 3:
// 3.5 ms against 0.49 ms for the atomic scatter over the item order); the caller reads the statistic of the PREVIOUS
// step (no synchronisation) and picks the form per level.
__global__ __launch_bounds__(256) void pull_stats_kernel(const int32_t *__restrict__ cs, int32_t *__restrict__ stats, int B, int D,
                                                         int H, int W, int XS) {
  const int nsx = (W + XS - 1) / XS;
  const int64_t total = (int64_t)B * (D + 1) * (H + 1) * nsx, i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int len = 0;
  if (i < total) {
    const int sx = (int)(i % nsx);
    const int64_t row = i / nsx;  // (b, cz, cy)
    const int x0 = sx * XS, ncell = min(XS, W - x0) + 1;
    const int64_t k = row * (W + 1) + x0;
    len = cs[k + ncell] - cs[k];
  }
  __shared__ int red[256];
  red[threadIdx.x] = len;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] = max(red[threadIdx.x], red[threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x == 0 && red[0] > 0) atomicMax(stats, red[0]);
}

int64_t pull_cells(int B, int D, int H, int W) { return (int64_t)B * (D + 1) * (H + 1) * (W + 1); }
int pull_key_bits(int64_t cells) {  // sentinel = cells must fit
  int nb = 1;
  while ((1LL << nb) <= cells) ++nb;
  return nb;
}
int64_t al256(int64_t x) { return (x + 255) / 256 * 256; }

int check_desc(const svr_gather_desc *d, bool bwd) {
  SVR_CHECK(d != nullptr, SVR_E_BADARG, "gather: null descriptor");
  SVR_CHECK(d->n_levels >= 1 && d->n_levels <= SVR_MAX_LEVELS, SVR_E_BADARG, "gather: n_levels=%d", d->n_levels);
  SVR_CHECK(d->B >= 0 && d->N >= 0, SVR_E_BADSHAPE, "gather: B=%d N=%d", d->B, d->N);
  SVR_CHECK(d->row_stride % 4 == 0, SVR_E_ALIGN, "gather: row_stride %d not a multiple of 4", d->row_stride);
  for (int l = 0; l < d->n_levels; ++l) {
    const svr_level &L = d->level[l];
    SVR_CHECK(L.vol != nullptr || (bwd && L.gvol == nullptr), SVR_E_BADARG, "gather: level %d has no volume", l);
    SVR_CHECK(L.C == 1 || L.C == 16 || L.C == 32 || L.C == 64 || L.C == 128, SVR_E_UNSUPPORTED,
              "gather: level %d: C=%d (supported: 1,16,32,64,128)", l, L.C);
    SVR_CHECK(L.D > 0 && L.H > 0 && L.W > 0, SVR_E_BADSHAPE, "gather: level %d: empty volume", l);
    if (bwd && L.gvol == nullptr && L.vol == nullptr) continue;  // level skipped by the backward (e.g. a projected level: the
                                                                 // compact kept-column matrix has no columns for it)
    SVR_CHECK(L.col >= 0 && L.col + 7 * L.C <= d->row_stride, SVR_E_BADSHAPE,
              "gather: level %d: columns [%d,%d) exceed row stride %d", l, L.col, L.col + 7 * L.C, d->row_stride);
    SVR_CHECK(L.C == 1 || L.col % 4 == 0, SVR_E_ALIGN, "gather: level %d: col %d not 16-byte aligned", l, L.col);
    SVR_CHECK(((uintptr_t)L.vol & 15) == 0 && ((uintptr_t)L.gvol & 15) == 0, SVR_E_ALIGN,
              "gather: level %d: volume pointer not 16-byte aligned", l);
  }
  return SVR_OK;
}

LevelArgs level_args(const svr_level &L) { return LevelArgs{L.vol, L.gvol, L.C, L.D, L.H, L.W, L.col, L.order, L.item_order}; }

}  // namespace

#define DISPATCH_C(Cval, ...)                         \
  switch (Cval) {                                     \
    case 1: { constexpr int CC = 1; __VA_ARGS__; } break;     \
    case 16: { constexpr int CC = 16; __VA_ARGS__; } break;   \
    case 32: { constexpr int CC = 32; __VA_ARGS__; } break;   \
    case 64: { constexpr int CC = 64; __VA_ARGS__; } break;   \
    case 128: { constexpr int CC = 128; __VA_ARGS__; } break; \
  }

extern "C" int svr_gather_trilinear_fwd(const svr_gather_desc *d, const float *points, float *features,
                                        void *stream) {
  if (int rc = check_desc(d, false)) return rc;
  if ((int64_t)d->B * d->N == 0) return SVR_OK;  // empty point set
  SVR_CHECK(points && features, SVR_E_BADARG, "gather_fwd: null points/features");
  SVR_CHECK(((uintptr_t)features & 15) == 0, SVR_E_ALIGN, "gather_fwd: features not 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  int64_t BN = (int64_t)d->B * d->N;
  if (BN == 0) return SVR_OK;
  // one launch; widest levels first so the long workgroups are scheduled early
  FusedArgs A;
  A.n = 0;
  unsigned blocks = 0;
  // 32-bit element / row offsets in the shared-geometry body: every volume and the feature matrix below 2^31 elements
  bool shared = BN * (int64_t)d->row_stride < (1LL << 31) && !(d->flags & SVR_GATHER_WIDE_OFFSETS);
  for (int l = 0; l < d->n_levels; ++l)
    if ((int64_t)d->B * d->level[l].D * d->level[l].H * d->level[l].W * d->level[l].C >= (1LL << 31)) shared = false;
  A.shared = shared ? 1 : 0;
  for (int pass = 0; pass < 2; ++pass)
    for (int l = d->n_levels - 1; l >= 0; --l) {
      if ((d->level[l].C >= 4) != (pass == 0)) continue;
      LevelArgs L = level_args(d->level[l]);
      int V = L.C >= 4 ? L.C / 4 : 1;
      A.L[A.n] = L;
      A.block_start[A.n] = blocks;
      blocks += svr::xcd_grid((shared && L.C >= 16) ? svr::cdiv(BN * 7, 256) : svr::cdiv(BN * 7 * V, 256));  // multiple of 8
      ++A.n;
    }
  A.block_start[A.n] = blocks;
  // Off by default: measured at config 3 it helps UNSORTED points (3.04 -> 2.57 ms) but costs Morton-sorted ones, the
  // production order (2.00 -> 2.20 ms: with round-robin placement the lines a neighbour XCD just fetched are Infinity
  // Cache hits, and eight XCDs streaming eight different samples open more DRAM pages).  SVR_GATHER_XCD=1 enables it.
  static const int xcd_order = getenv("SVR_GATHER_XCD") ? 1 : 0;   // read once per process, not per launch
  A.xcd = xcd_order;
  A.pad_start = 0;
  for (int l = 0; l < d->n_levels; ++l) {
    int end = d->level[l].col + 7 * d->level[l].C;
    if (end > A.pad_start) A.pad_start = end;
  }
  hipLaunchKernelGGL(gather_fwd_fused_kernel, dim3(blocks), dim3(256), 0, s, A, points, features, d->order, BN, d->N,
                     d->row_stride, d->displacement, d->align_corners);
  return svr::launch_status("gather_fwd");
}

extern "C" int svr_gather_trilinear_bwd(const svr_gather_desc *d, const float *points, const float *gfeatures,
                                        float *gpoints, void *stream) {
  if (int rc = check_desc(d, true)) return rc;
  if ((int64_t)d->B * d->N == 0) return SVR_OK;  // empty point set: nothing to scatter
  SVR_CHECK(points && gfeatures, SVR_E_BADARG, "gather_bwd: null points/gfeatures");
  hipStream_t s = (hipStream_t)stream;
  int64_t BN = (int64_t)d->B * d->N;
  if (BN == 0) return SVR_OK;
  if (gpoints) {
    hipError_t e = hipMemsetAsync(gpoints, 0, (size_t)BN * 3 * sizeof(float), s);
    SVR_CHECK(e == hipSuccess, (int)e, "gather_bwd: memset failed: %s", hipGetErrorString(e));
  }
  FusedArgs FA;
  FA.n = 0;
  unsigned fblocks = 0;
  for (int l = d->n_levels - 1; l >= 0; --l) {
    LevelArgs L = level_args(d->level[l]);
    bool gv = L.gvol != nullptr;
    if (!gv && !gpoints) continue;
    SVR_CHECK(L.vol != nullptr || !gpoints, SVR_E_BADARG, "gather_bwd: level %d needs vol for the point gradient", l);
    if (gpoints) {  // rare path (points require grad): per-thread float4 kernel, point gradient only
      int V = L.C >= 4 ? L.C / 4 : 1;
      int64_t total = BN * 7 * V;
      unsigned grid = (unsigned)svr::cdiv(total, 256);
      DISPATCH_C(L.C, hipLaunchKernelGGL((gather_bwd_kernel<CC, false, true>), dim3(grid), dim3(256), 0, s, L, points,
                                         gfeatures, gpoints, total, d->N, d->row_stride, d->displacement,
                                         d->align_corners));
    }
    if (!gv) continue;
    if (d->level[l].plan && !(d->flags & SVR_GATHER_DETERMINISTIC)) {  // atomic-free pull form: gvol is overwritten
      const svr_pull_plan *P = d->level[l].plan;
      SVR_CHECK(L.C == 16 || L.C == 32 || L.C == 64, SVR_E_UNSUPPORTED, "gather_bwd: level %d: pull plan with C=%d", l, L.C);
      SVR_CHECK(P->keys && P->recs && P->heads && P->n_items == 7 * BN, SVR_E_BADARG, "gather_bwd: level %d: bad pull plan", l);
      SVR_CHECK(P->n_items < (1LL << 31) && BN * (int64_t)d->row_stride < (1LL << 31), SVR_E_UNSUPPORTED,
                "gather_bwd: level %d: pull plan needs 32-bit item / gradient offsets", l);
      const int n = (int)P->n_items;
      const PullRec *recs = (const PullRec *)P->recs;
#define SVR_PULL(CC, XS, BZ, BY, BXG)                                                                                     \
  {                                                                                                                       \
    const int64_t blocks = (int64_t)d->B * svr::cdiv(L.D, BZ) * svr::cdiv(L.H, BY) * svr::cdiv(L.W, BXG * XS);            \
    SVR_CHECK(blocks < (1LL << 31), SVR_E_UNSUPPORTED, "gather_bwd: level %d: %ld pull workgroups", l, (long)blocks);     \
    hipLaunchKernelGGL((gather_bwd_pull_kernel<CC, XS, BZ, BY, BXG>), dim3((unsigned)blocks), dim3(256), 0, s, L.gvol,    \
                       P->keys, recs, P->heads, gfeatures, n, d->B, L.D, L.H, L.W);                                       \
  }
      // strips of 8 / 4 / 4 voxels (measured at config 3: 0.55 / 0.38 / 0.57 ms; strips of 4 / 2 / 2: 0.57 / 0.41 / 0.66,
      // one voxel per thread: 1.29 / 0.60 / 0.66)
      if (L.C == 16) SVR_PULL(16, 8, 4, 4, 4)
      else if (L.C == 32) SVR_PULL(32, 4, 2, 4, 4)
      else SVR_PULL(64, 4, 2, 2, 4)
#undef SVR_PULL
      continue;
    }
    if (d->flags & SVR_GATHER_DETERMINISTIC) {
      hipLaunchKernelGGL(gather_bwd_serial_kernel, dim3((unsigned)d->B), dim3(128), 0, s, L, points, gfeatures, d->N,
                         d->row_stride, d->displacement, d->align_corners);
      continue;
    }
    if (L.C == 1) {  // level 0 (raw grid): scalar atomics
      int64_t total = BN * 7;
      hipLaunchKernelGGL((gather_bwd_kernel<1, true, false>), dim3((unsigned)svr::cdiv(total, 256)), dim3(256), 0, s, L,
                         points, gfeatures, gpoints, total, d->N, d->row_stride, d->displacement, d->align_corners);
      continue;
    }
    FA.L[FA.n] = L;
    FA.block_start[FA.n] = fblocks;
    SVR_CHECK(L.D < 1022 && L.H < 1022 && L.W < 1022, SVR_E_UNSUPPORTED, "gather_bwd: level %d: dims above 1021", l);
    if (L.items) {
      SVR_CHECK(7 * BN < (1LL << 31) && BN * (int64_t)d->row_stride < (1LL << 31), SVR_E_UNSUPPORTED,
                "gather_bwd: level %d: item order needs 32-bit item / gradient offsets", l);
      fblocks += (unsigned)svr::cdiv(bwd_items_waves(L.C, 7 * BN) * 64, 256);
    } else {
      fblocks += (unsigned)svr::cdiv(bwd_runs_waves(L.C, d->B, d->N) * 64, 256);
    }
    ++FA.n;
  }
  if (FA.n > 0) {
    FA.block_start[FA.n] = fblocks;
    hipLaunchKernelGGL(gather_bwd_fused_kernel, dim3(fblocks), dim3(256), 0, s, FA, points, gfeatures, d->order, d->B, d->N,
                       d->row_stride, d->displacement, d->align_corners);
  }
  return svr::launch_status("gather_bwd");
}

extern "C" int svr_gather_corner_indices(const svr_gather_desc *d, int32_t level, const float *points,
                                         int32_t *out, void *stream) {
  if (int rc = check_desc(d, false)) return rc;
  SVR_CHECK(level >= 0 && level < d->n_levels, SVR_E_BADARG, "corner_indices: level %d", level);
  SVR_CHECK(points && out, SVR_E_BADARG, "corner_indices: null pointer");
  const svr_level &L = d->level[level];
  int64_t total = (int64_t)d->B * 7 * d->N;
  if (total == 0) return SVR_OK;
  hipLaunchKernelGGL(corner_index_kernel, dim3((unsigned)svr::cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     points, out, total, d->N, L.D, L.H, L.W, d->displacement, d->align_corners);
  return svr::launch_status("corner_indices");
}

extern "C" int64_t svr_gather_pull_plan_workspace(int32_t B, int32_t N) {
  // items: 3 x uint32 arrays + radix temp; the raw cell ends (cells + 1 ints, at most 2^31 cells) are sized by
  // svr_gather_pull_plan_workspace_cells below -- callers add both
  const int64_t T = (int64_t)7 * B * N;
  if (T <= 0) return 256;
  return 3 * al256(T * 4) + al256((int64_t)svr::sort_pairs_u32_temp_bytes(T, 32)) + 256;
}

extern "C" int64_t svr_gather_pull_plan_workspace_cells(int32_t B, int32_t D, int32_t H, int32_t W) {
  const int64_t cells = pull_cells(B, D, H, W);
  return al256((cells + 1) * 4) + al256((int64_t)svr::scan_max_i32_temp_bytes(cells + 1)) + 256;
}

extern "C" int svr_gather_pull_plan(const float *points, int32_t B, int32_t N, int32_t D, int32_t H, int32_t W, int32_t C,
                                    int32_t col, int32_t row_stride, int32_t align_corners, float displacement,
                                    uint32_t *keys, void *recs, int32_t *heads, int32_t *items_out, int32_t *stats,
                                    void *workspace, void *stream) {
  const int64_t T = (int64_t)7 * B * N;
  SVR_CHECK(B >= 0 && N >= 0 && D > 0 && H > 0 && W > 0, SVR_E_BADSHAPE, "pull_plan: B=%d N=%d dims %dx%dx%d", B, N, D, H, W);
  SVR_CHECK(heads && workspace, SVR_E_BADARG, "pull_plan: null heads / workspace");
  hipStream_t s = (hipStream_t)stream;
  const int64_t cells = pull_cells(B, D, H, W);
  SVR_CHECK(cells < (1LL << 31) - 1 && T < (1LL << 31) && (int64_t)B * N * row_stride < (1LL << 31), SVR_E_UNSUPPORTED,
            "pull_plan: %ld cells / %ld items / %ld gradient floats exceed 32-bit offsets", (long)cells, (long)T,
            (long)((int64_t)B * N * row_stride));
  char *w = (char *)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  int32_t *ends = (int32_t *)w;   // raw cell ends, then the running maximum goes to `heads`
  w += al256((cells + 1) * 4);
  void *scan_tmp = (void *)w;
  const size_t scan_bytes = svr::scan_max_i32_temp_bytes(cells + 1);
  w += al256((int64_t)scan_bytes);
  hipError_t e = hipMemsetAsync(ends, 0, (size_t)(cells + 1) * sizeof(int32_t), s);
  SVR_CHECK(e == hipSuccess, (int)e, "pull_plan: memset failed: %s", hipGetErrorString(e));
  if (T > 0) {
    SVR_CHECK(points && keys && recs, SVR_E_BADARG, "pull_plan: null pointer");
    uint32_t *keys_in = (uint32_t *)w;
    w += al256(T * 4);
    int32_t *vals_in = (int32_t *)w;
    w += al256(T * 4);
    int32_t *items = items_out ? items_out : (int32_t *)w;   // the sorted item ids double as svr_level.item_order
    w += al256(T * 4);
    const uint32_t sentinel = (uint32_t)cells;
    const int bits = pull_key_bits(cells);
    hipLaunchKernelGGL(pull_key_kernel, dim3((unsigned)svr::cdiv(T, 256)), dim3(256), 0, s, points, keys_in, vals_in, T, N, D,
                       H, W, displacement, align_corners, sentinel, 0);
    e = svr::sort_pairs_u32((void *)w, svr::sort_pairs_u32_temp_bytes(T, bits), keys_in, keys, vals_in, items, T, bits, s);
    SVR_CHECK(e == hipSuccess, (int)e, "pull_plan: radix sort failed: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(pull_record_kernel, dim3((unsigned)svr::cdiv(T, 256)), dim3(256), 0, s, points, keys, items,
                       (PullRec *)recs, ends, T, N, D, H, W, C, col, row_stride, displacement, align_corners, sentinel);
  }
  e = svr::scan_max_i32(scan_tmp, scan_bytes, ends, heads, cells + 1, s);
  SVR_CHECK(e == hipSuccess, (int)e, "pull_plan: scan failed: %s", hipGetErrorString(e));
  if (stats) {
    e = hipMemsetAsync(stats, 0, 2 * sizeof(int32_t), s);
    SVR_CHECK(e == hipSuccess, (int)e, "pull_plan: memset failed: %s", hipGetErrorString(e));
    const int XS = pull_xs(C);
    const int64_t strips = (int64_t)B * (D + 1) * (H + 1) * svr::cdiv(W, XS);
    hipLaunchKernelGGL(pull_stats_kernel, dim3((unsigned)svr::cdiv(strips, 256)), dim3(256), 0, s, (const int32_t *)heads, stats, B,
                       D, H, W, XS);
  }
  return svr::launch_status("pull_plan");
}

extern "C" int svr_gather_item_order(const float *points, int32_t B, int32_t N, int32_t D, int32_t H, int32_t W,
                                     int32_t align_corners, float displacement, int32_t with_j, int32_t *items,
                                     void *workspace, void *stream) {
  const int64_t T = (int64_t)7 * B * N;
  SVR_CHECK(B >= 0 && N >= 0 && D > 0 && H > 0 && W > 0, SVR_E_BADSHAPE, "item_order: B=%d N=%d dims %dx%dx%d", B, N, D, H, W);
  if (T == 0) return SVR_OK;
  SVR_CHECK(points && items && workspace, SVR_E_BADARG, "item_order: null pointer");
  const int64_t cells = pull_cells(B, D, H, W), nkeys = with_j ? cells * 8 : cells;
  SVR_CHECK(nkeys < (1LL << 31) - 1 && T < (1LL << 31), SVR_E_UNSUPPORTED, "item_order: %ld keys / %ld items exceed 32 bits",
            (long)nkeys, (long)T);
  hipStream_t s = (hipStream_t)stream;
  char *w = (char *)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  uint32_t *keys_in = (uint32_t *)w;
  w += al256(T * 4);
  int32_t *vals_in = (int32_t *)w;
  w += al256(T * 4);
  uint32_t *keys_out = (uint32_t *)w;
  w += al256(T * 4);
  const int bits = pull_key_bits(nkeys);
  hipLaunchKernelGGL(pull_key_kernel, dim3((unsigned)svr::cdiv(T, 256)), dim3(256), 0, s, points, keys_in, vals_in, T, N, D, H,
                     W, displacement, align_corners, (uint32_t)nkeys, with_j ? 1 : 0);
  hipError_t e = svr::sort_pairs_u32((void *)w, svr::sort_pairs_u32_temp_bytes(T, bits), keys_in, keys_out, vals_in, items, T,
                                     bits, s);
  SVR_CHECK(e == hipSuccess, (int)e, "item_order: radix sort failed: %s", hipGetErrorString(e));
  return svr::launch_status("item_order");
}

extern "C" int svr_gather_project_bwd(const float *points, const float *dh, int64_t lddh, int32_t B, int32_t N, int32_t D,
                                      int32_t H, int32_t W, int32_t align_corners, float displacement, const int32_t *items,
                                      float *dP, void *stream) {
  const int64_t T = (int64_t)7 * B * N;
  if (T == 0) return SVR_OK;
  SVR_CHECK(points && dh && items && dP, SVR_E_BADARG, "project_bwd: null pointer");
  SVR_CHECK(D > 0 && H > 0 && W > 0 && D < 1022 && H < 1022 && W < 1022, SVR_E_BADSHAPE, "project_bwd: dims %dx%dx%d", D, H, W);
  SVR_CHECK(T < (1LL << 31) && (int64_t)B * N * lddh < (1LL << 31) && lddh >= 256, SVR_E_UNSUPPORTED,
            "project_bwd: needs 32-bit item / row offsets and rows of >= 256 floats");
  LevelArgs L{nullptr, dP, 256, D, H, W, 0, nullptr, items};
  const int64_t waves = bwd_items_waves_cpl(256, 4, T, kProjReps);
  hipLaunchKernelGGL(gather_bwd_proj_kernel, dim3((unsigned)svr::cdiv(waves * 64, 256)), dim3(256), 0, (hipStream_t)stream, L, points,
                     dh, T, N, (int)lddh, displacement, align_corners, waves);
  return svr::launch_status("project_bwd");
}

// ---- two-pass projected scatter (see gather_bwd_proj_store_kernel)
extern "C" int64_t svr_gather_project_plan_workspace(int32_t B, int32_t N) {
  const int64_t T = (int64_t)7 * B * N;
  if (T <= 0) return 256;
  return 2 * al256(T * 4) + al256((int64_t)svr::sort_pairs_u32_temp_bytes(T, 32)) + al256((T + 1) * 4) +
         al256((int64_t)svr::scan_sum_excl_i32_temp_bytes(T + 1)) + 512;
}

extern "C" int64_t svr_gather_project_slots(int32_t B, int32_t N, int32_t D, int32_t H, int32_t W) {
  const int64_t T = (int64_t)7 * B * N, nkeys = pull_cells(B, D, H, W) * 8;
  return (T < nkeys ? T : nkeys) + svr::cdiv(T, 2 * 64 * kProjReps) + 1;   // distinct keys + one split per wave chunk
}

extern "C" int svr_gather_project_plan(const float *points, int32_t B, int32_t N, int32_t D, int32_t H, int32_t W,
                                       int32_t align_corners, float displacement, int32_t *items, uint32_t *keys, int32_t *sidx,
                                       int32_t *first_slot, void *workspace, void *stream) {
  const int64_t T = (int64_t)7 * B * N;
  SVR_CHECK(B >= 0 && N >= 0 && D > 0 && H > 0 && W > 0, SVR_E_BADSHAPE, "project_plan: B=%d N=%d dims %dx%dx%d", B, N, D, H, W);
  if (T == 0) return SVR_OK;
  SVR_CHECK(points && items && keys && sidx && first_slot && workspace, SVR_E_BADARG, "project_plan: null pointer");
  const int64_t nkeys = pull_cells(B, D, H, W) * 8;
  SVR_CHECK(nkeys < (1LL << 31) - 1 && T < (1LL << 31) - 1, SVR_E_UNSUPPORTED, "project_plan: %ld keys / %ld items exceed 32 bits",
            (long)nkeys, (long)T);
  hipStream_t s = (hipStream_t)stream;
  char *w = (char *)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  uint32_t *keys_in = (uint32_t *)w;
  w += al256(T * 4);
  int32_t *vals_in = (int32_t *)w;
  w += al256(T * 4);
  const int bits = pull_key_bits(nkeys);
  void *sort_tmp = w;
  w += al256((int64_t)svr::sort_pairs_u32_temp_bytes(T, 32));
  int32_t *flags = (int32_t *)w;
  w += al256((T + 1) * 4);
  hipLaunchKernelGGL(pull_key_kernel, dim3((unsigned)svr::cdiv(T, 256)), dim3(256), 0, s, points, keys_in, vals_in, T, N, D, H,
                     W, displacement, align_corners, (uint32_t)nkeys, 1);
  hipError_t e = svr::sort_pairs_u32(sort_tmp, svr::sort_pairs_u32_temp_bytes(T, bits), keys_in, keys, vals_in, items, T, bits, s);
  SVR_CHECK(e == hipSuccess, (int)e, "project_plan: radix sort failed: %s", hipGetErrorString(e));
  hipLaunchKernelGGL(proj_flag_kernel, dim3((unsigned)svr::cdiv(T + 1, 256)), dim3(256), 0, s, (const uint32_t *)keys, flags, T,
                     2 * 64 * kProjReps, (uint32_t)nkeys);
  e = svr::scan_sum_excl_i32((void *)w, svr::scan_sum_excl_i32_temp_bytes(T + 1), flags, sidx, T + 1, s);
  SVR_CHECK(e == hipSuccess, (int)e, "project_plan: scan failed: %s", hipGetErrorString(e));
  hipLaunchKernelGGL(proj_table_kernel, dim3((unsigned)svr::cdiv(nkeys + 1, 256)), dim3(256), 0, s, (const uint32_t *)keys,
                     (const int32_t *)sidx, first_slot, nkeys, T);
  return svr::launch_status("project_plan");
}

extern "C" int svr_gather_project_bwd2(const float *points, const float *dh, int64_t lddh, int32_t B, int32_t N, int32_t D,
                                       int32_t H, int32_t W, int32_t align_corners, float displacement, const int32_t *items,
                                       const int32_t *sidx, const int32_t *first_slot, float *partials, float *dP, void *stream) {
  const int64_t T = (int64_t)7 * B * N;
  SVR_CHECK(dP && D > 0 && H > 0 && W > 0 && D < 1022 && H < 1022 && W < 1022, SVR_E_BADSHAPE, "project_bwd2: dims %dx%dx%d", D, H, W);
  hipStream_t s = (hipStream_t)stream;
  if (T > 0) {
    SVR_CHECK(points && dh && items && sidx && first_slot && partials, SVR_E_BADARG, "project_bwd2: null pointer");
    SVR_CHECK(T < (1LL << 31) - 1 && (int64_t)B * N * lddh < (1LL << 31) && lddh >= 256, SVR_E_UNSUPPORTED,
              "project_bwd2: needs 32-bit item / row offsets and rows of >= 256 floats");
    LevelArgs L{nullptr, partials, 256, D, H, W, 0, sidx, items};   // gvol = the partial-sum buffer, order = the slot numbers
    const int64_t waves = bwd_items_waves_cpl(256, 4, T, kProjReps);
    hipLaunchKernelGGL(gather_bwd_proj_store_kernel, dim3((unsigned)svr::cdiv(waves * 64, 256)), dim3(256), 0, s, L, points, dh, T,
                       N, (int)lddh, displacement, align_corners, waves);
    const int64_t rows = (int64_t)B * D * H * W * 7;
    hipLaunchKernelGGL(proj_combine_kernel, dim3((unsigned)svr::cdiv(rows, 4)), dim3(256), 0, s, (const float *)partials, first_slot,
                       dP, B, D, H, W);
  }
  return svr::launch_status("project_bwd2");
}
