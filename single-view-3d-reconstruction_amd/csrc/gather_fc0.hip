// Fused trilinear gather -> fc_0 for gfx950: the feature rows never travel through HBM.
//
// Replaces, in one kernel, the six F.grid_sample calls + torch.cat of model/ifnet.py:156-197 AND the first Conv1d of the
// point MLP (model/ifnet.py:43-45,55: fc_0 + ReLU).  The separate kernels (gather.hip's fused forward, gemm_f16x3.hip)
// write the (B*N, 2592) feature matrix -- 4.1 GB at 128^3 x 50k x 8 -- and read it back: 2.05 + 2.13 ms, both bound by
// those bytes.  Here a workgroup owns FC_TM = 64 consecutive (Morton-sorted) points and all 256 output columns (two
// workgroups per CU; 128-point tiles at one workgroup per CU are a build option, -DFC_TM=128):
//   * 4 PRODUCER waves gather one K-slab (<= 64 feature columns of one level: one or two displacements x a channel
//     range) for the tile's points -- same geometry and summation order as gather.hip, so the values are bit-identical --
//     split it into the f16 hi / lo planes of the 3-product split (f16x3.h) and store it in LDS;
//   * 4 CONSUMER waves (one per SIMD, FC_TM x 64 outputs each) multiply the previous slab from the other LDS buffer with
//     W's pre-split planes, which they read straight from L2 in MFMA fragment layout (no LDS staging, no redundancy
//     between the consumers), three v_mfma_f32_32x32x16_f16 per 32 x 32 x 16 block;
//   * one s_barrier per slab hands the buffers over.
// The kernel is bound by the bytes the producers pull through the vector L1 (93 KB per point of corner reads, most of
// them cache hits) -- the matrix cores wait for the gather, not the other way round.
// Levels whose rows the backward still needs (the ones that are not projected, ifnet.py) are ALSO written to a feature
// matrix: either in gather.hip's full row layout or -- what the training step uses -- a COMPACT kept-column matrix
// (its own column per level and row stride: 800 columns instead of a 2592-wide row at the 128-architecture).
//
// Measurement switches (parts of the kernel turned off: wrong results by construction) exist only in builds with
// -DSVR_FC0_MEASURE (tools/exp/prof_fc0.sh); the production kernel carries none.
//
// bf16-STORAGE variant (template flag BF; svr_gather_fc0_bf16_*: the throughput mode of bf16_path.hip, never the default):
// volumes in bf16 (8-byte corner loads for the same four channels per lane), f32 corner sums in ATen's order, ONE rounding
// to bf16 (the feature row bf16_path.hip's gather would have written, bit for bit), one LDS plane, W as one bf16 plane and
// one v_mfma_f32_32x32x16_bf16 per block instead of three f16 products, h0 stored in bf16.  No kept columns.
//
// Build with -ffp-contract=off (gather_common.h).
#include "common.h"
#include "gather_common.h"
#include "f16x3.h"
#include <algorithm>
#include <cstdlib>

using namespace svr;

namespace {

#ifdef SVR_FC0_MEASURE
__device__ unsigned long long *fc_stamps = nullptr;   // measurement builds: timeline / debug buffer (svr_gather_fc0_stamps)
__device__ const float *fc_dbg_pt0 = nullptr;         // ... and the point pointer of the tile whose geometry table is dumped
__device__ int fc_dbg_C = 64;                         // ... at the level with this many channels
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define GLOBAL_AS __attribute__((address_space(1)))
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16_rne(float a, float b) {  // round to nearest even (v_cvt_pk_bf16_f32), NaN safe
  f32x2_t v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

constexpr int FK = 16;                 // reduction elements per MFMA step
constexpr int FLW = 8;                 // dwords per LDS row of one k-step: 16 halves = two 16-byte slots (fc_slot_dw), as gemm_f16x3.hip
#ifndef FC_TM
#define FC_TM 64
#endif
constexpr int FTM = FC_TM;             // points per workgroup.  64 (two workgroups per CU: 128 VGPRs, 42 KB of LDS each,
                                       // 16 waves per CU hide the producers' load latency; one pass in flight per producer
                                       // wave) 2.58 ms at config 3 against 2.83 for 128 (one workgroup per CU, 255 VGPRs,
                                       // three passes in flight): W is read twice as often (every 64 points), still faster
#ifndef FC_NPW
#define FC_NPW 4
#endif
constexpr int NPW = FC_NPW;            // producer waves per workgroup: 4 (512 threads, two workgroups per CU at 128 VGPRs) or
                                       // 8 (768 threads = 3 waves per SIMD at 168 VGPRs, one workgroup per CU: round 4 -- the
                                       // in-kernel timeline showed a producer pass to be one ~1 600-cycle dependency chain and
                                       // a consumer k-step to be one ~1 500-cycle W-fragment load latency; what hides them is
                                       // passes / loads in flight, i.e. registers per wave)
constexpr int NPT = NPW * 64;          // producer threads
constexpr int NTHR = 256 + NPT;        // threads per workgroup (4 consumer waves first)
constexpr int RPW = FTM / NPW;         // rows per producer wave
constexpr int FMT = FTM / 32;          // 32-row MFMA tiles per consumer wave
constexpr int FTN = 256;               // output columns (fc_0's width)
constexpr int FPLANE = FTM * FLW;      // dwords per plane and k-step
constexpr int FKSTEP = 2 * FPLANE + 8;   // hi + lo; + 8 dwords: the four k-steps of one producer store (64-column slabs) land on different banks
constexpr int FSLAB_K = 4;             // k-steps per slab buffer
constexpr int FSLAB = FSLAB_K * FKSTEP;
constexpr int FC_MAX_SLABS = 56;
// LDS staging of the coarse levels (round 3).  Counters (tools/pmc_counters.sh): the texture addresser is busy 69 % of the
// kernel's time, the matrix pipe 29 %, LDS 24 % -- the 8 corner loads per (point, displacement, 4 channels) are what bounds
// it, and 56 of the 81 KB a point reads belong to the two 128-channel levels, where a Morton tile of 64 points touches a box
// of ~30 (8^3) / ~200 (16^3) voxels 7 x 8 x 64 times.  A level with D, H, W <= 16 and C % 64 == 0 is therefore visited one
// 64-channel half at a time: a STAGE slab (no k-steps) copies the tile's bounding box x 64 channels into LDS (at most
// FC_STAGE_VOX voxels: 40 KB; one coalesced float4 per lane), the seven displacement slabs behind it take their corners
// from there with ds_read_b128 (16 lanes = one voxel's 256 bytes: all 64 banks, conflict free) -- same values, same order
// of operations, bit-identical features.  The box of every (tile, level) comes from fc0_boxes_kernel (one wave per tile, in
// front of the launch); a tile whose box is larger, or that straddles two samples, keeps the global loads (box.nvox = 0).
#ifndef FC_GEO_CHECK
#define FC_GEO_CHECK 0   // debug (measurement builds): both geometry forms, compared value by value inside the kernel
#endif
#ifndef FC_GEO_LDS
// 1: per-level sample geometry of a tile in LDS (per row and AXIS: 9 KB) instead of 42 producer registers, which lets a producer
// wave keep two passes in flight inside 128 VGPRs: 2.35 instead of 2.49 ms stand-alone.  NOT the default: with one or two passes
// in flight (FC_DEPTH 1 / 2) some (row, displacement) items of the UPPER half-wave come out wrong (8-10 of 12 level checks of
// tools/exp/dbg_fc0.py, the same rows from run to run within a build), with three (FC_DEPTH=3) every check passes.  Established
// with measurement builds (tools/exp/dbg_geo*.py, FC_GEO_CHECK): the table is correct right after it is written (levels 2-3, ten
// tiles, against a host restatement), two consecutive reads of an entry always agree (no concurrent writer), and when BOTH forms
// are computed the LDS values equal the register values lane by lane (and the output is then right).  Extra s_waitcnt vmcnt(0) /
// lgkmcnt(0) around the table, unmerged reads and unsigned lane arithmetic do not change it.  Cause not found -> not shipped.
#define FC_GEO_LDS 0
#endif
constexpr int FC_STAGE_VOX = FC_GEO_LDS ? 144 : 160;   // (144: the geometry table takes 9 KB of the 80 KB a workgroup may use)
constexpr int FC_STAGE_DW = FC_STAGE_VOX * 64;   // dwords of the staging region (64 channels per voxel)
constexpr int FC_NSTAGE = 2;                     // at most two staged levels
struct FcBox { int nvox, b, z0, y0, x0, by, bx, pad; };   // per (tile, staged level): 32 bytes, scalar loads
#ifndef FC_DEPTH
#define FC_DEPTH ((FC_NPW == 8 || FC_GEO_LDS) ? 2 : 1)
#endif
#ifndef FC_KEEP_NT
#define FC_KEEP_NT 1   // nontemporal stores of the kept feature columns (2.636 -> 2.598 ms with levels 1-3 kept; 0: ordinary stores)
#endif
#ifndef FC_PRIO
#define FC_PRIO 0  // s_setprio of the producer waves (the second-dispatched half of the workgroup loses VALU arbitration by age)
#endif
#ifndef FC_PK
#define FC_PK 1    // corner sum / f16 split as packed f32 instructions (v_pk_mul/add/fma_f32) or as plain ones
#endif
#ifndef FC_FMA
#define FC_FMA 0   // corner sum: 0 = ATen's rounding (bit-identical to F.grid_sample), 1 = one v_pk_fma_f32 per step
#endif
// Geometry table (FC_GEO_LDS): per tile row 36 dwords = byte offsets [axis x, y, z][variant 0, -d, +d][corner 0, 1], then the
// weights in the same order (0 where the corner lies outside the volume).  A displacement moves ONE axis, so the 7 x 3 axis
// evaluations of a row's displacements are 9 distinct ones; a pass reads 3 + 3 eight-byte entries instead of 14 ds_bpermute
// and keeps no geometry in registers -- which is what lets a producer wave have TWO passes in flight inside 128 VGPRs.
constexpr int GEO_ROW_DW = 36;
constexpr int FC_GEO_DW = FC_GEO_LDS ? FTM * GEO_ROW_DW : 0;
constexpr int FC_LDS_BYTES = 2 * FSLAB * 4 + FC_STAGE_DW * 4 + FC_GEO_DW * 4;   // two slab buffers + the staging region + geometry
static_assert(2 * FC_LDS_BYTES <= 160 * 1024 || FTM != 64, "two workgroups per CU");

struct FcLevel {
  const float *vol;
  int C, D, H, W, col;
  int kcol;  // first column of the level in the KEPT matrix (col for the full row layout)
};
struct FcSlab {  // 32-bit fields: scalar loads.  (Sub-dword fields were fetched with VECTOR loads + s_waitcnt vmcnt(0), which
                 // drained the consumers' W prefetch at every slab boundary.)
  int level, j0, nj, lp;  // lp: lanes per point (columns / 4); 0 marks the C == 1 slab (7 columns + 9 zeros)
  int k0, nk;             // first k-step and number of k-steps
  int c0, keep;           // first channel; keep: also store the values to the feature matrix
  int stage, geo;         // stage: -1, or the staged-level index of a level whose corners come from the LDS staging region;
                          // lp == -1 marks the STAGE slab itself (nk == 0).  geo: first displacement slab of its level --
                          // the producers evaluate the level's sample geometry there (level_geometry) and keep it
};
struct FcArgs {
  FcLevel L[SVR_MAX_LEVELS];
  FcSlab S[FC_MAX_SLABS];
  int n_slabs, KF;  // KF: fused reduction length (multiple of 16)
  int n_stage, stage_level[FC_NSTAGE];
};

// fused K order -> column of W's (= the feature row's) layout, -1 for the padding of the C == 1 slab
__device__ __forceinline__ int fused_col(const FcArgs &A, int kk) {
  for (int s = 0; s < A.n_slabs; ++s) {
    const FcSlab S = A.S[s];
    const int r = kk - S.k0 * FK;
    if (r < 0 || r >= S.nk * FK) continue;
    const FcLevel L = A.L[S.level];
    if (S.lp == 0) return r < 7 ? L.col + r : -1;
    const int nc = S.lp * 4 / S.nj;
    return L.col + (S.j0 + r / nc) * L.C + S.c0 + r % nc;
  }
  return -1;
}

// Fragment-major W planes: the 16 halves x 32 rows of one MFMA B-fragment are 1 KB contiguous (lane l of the consumer wave
// reads 16 bytes at lane * 16), k-step major: [k-step][hi / lo][row tile of 32][lane 64][8 halves].  A row-major plane
// made every fragment load touch 32 cache lines for 32 bytes each and the L1 re-fetched every line four times.
__device__ __forceinline__ int64_t wfrag_index(int kk, int plane, int n, int ntiles, int nplanes = 2) {
  return ((((int64_t)(kk >> 4) * nplanes + plane) * ntiles + (n >> 5)) * 64 + ((kk >> 3) & 1) * 32 + (n & 31)) * 8 + (kk & 7);
}

// W[N][K] f32 -> fragment-major f16 planes in the fused K order: hi(Ws), lo(Ws)
__global__ void split_w_fused_kernel(FcArgs A, const float *__restrict__ W, int64_t ldw, const uint32_t *__restrict__ amax,
                                     uint16_t *__restrict__ p0, int N) {
  const int kk = blockIdx.x * blockDim.x + threadIdx.x, n = blockIdx.y;
  if (kk >= A.KF || n >= N) return;
  const int src = fused_col(A, kk);
  const float w = src >= 0 ? W[(int64_t)n * ldw + src] * w_scale(amax[0], false) : 0.f;
  const _Float16 h = (_Float16)w;  // round to nearest even, like pack_f16
  const _Float16 l = (_Float16)(w - (float)h);
  p0[wfrag_index(kk, 0, n, N / 32)] = __builtin_bit_cast(uint16_t, h);
  p0[wfrag_index(kk, 1, n, N / 32)] = __builtin_bit_cast(uint16_t, l);
}

// bf16-storage variant: ONE fragment-major plane of bf16(W) (round to nearest even: the bits of svr_cast_f32_to_bf16)
__global__ void split_w_fused_bf16_kernel(FcArgs A, const float *__restrict__ W, int64_t ldw, uint16_t *__restrict__ p0, int N) {
  const int kk = blockIdx.x * blockDim.x + threadIdx.x, n = blockIdx.y;
  if (kk >= A.KF || n >= N) return;
  const int src = fused_col(A, kk);
  const float w = src >= 0 ? W[(int64_t)n * ldw + src] : 0.f;
  p0[wfrag_index(kk, 0, n, N / 32, 1)] = (uint16_t)(pack_bf16_rne(w, 0.f) & 0xffffu);
}

// LDS rows of the feature tile: slot (16 bytes = 8 halves) h of a row sits at 2 row + (h ^ bit 3 of the row), so that a
// consumer's fragment is ONE ds_read_b128 (256 B/clk; the padded 8-byte aligned rows of the first version made it a
// ds_read2_b64 at 128 B/clk) and its 16-lane groups hit 16 different slots (see gemm_f16x3.hip / conv3d_bf16.hip).
__device__ __forceinline__ int fc_slot_dw(int row, int h) { return (row * 2 + (h ^ ((row >> 3) & 1))) * 4; }

__device__ __forceinline__ f16x8 lds_frag(const uint32_t *plane, int row, int lh) {
  union { uint4 q; f16x8 v; } f;
  f.q = *reinterpret_cast<const uint4 *>(plane + fc_slot_dw(row, lh));
  return f.v;
}

// Levels with C >= 16.  The producers are bound by the instructions they issue (one wave per SIMD, 4 cycles per wave64
// VALU instruction), so the code is split by how often each part has to run:
//   level_geometry (ONCE per tile and level; round 4 -- rounds 2-3 redid it in every one of the level's 4-14 slabs, ~250
//     instructions per slab and wave against ~100 per pass: 38-55 % of a slab's issue slots): producer wave pw owns rows
//     [RPW pw, RPW pw + RPW) of the tile; its 7 RPW (row, displacement) items are evaluated completely, one per lane and
//     round, in displacement-major order (item = j RPW + row; set = item / 64, lane = item % 64) and stay in registers
//     for all slabs of the level: the eight corner weights (wx*wy)*wz with 0 for corners outside the volume, and the byte
//     offsets of the CLAMPED corner coordinates (four (z,y) row offsets + two x offsets, without the slab's first
//     channel), so a pass needs neither bounds tests nor address selects;
//   produce_slab (per slab): LP lanes per point (LP * 4 columns), NJ displacements side by side; LP / 2 passes of
//     64 / LP rows; the owning lane's 14 values arrive by ds_bpermute, 8 float4 loads (uniform base + 32-bit offset),
//     sum_k v_k * w_k in ATen's corner order.  A corner outside the volume contributes v * 0 with v read from a clamped,
//     i.e. existing, voxel: the sum is bit-identical to skipping it (ATen, gather.hip) for finite volumes.
// Corner sum: FC_FMA = 0 rounds product and sum separately, in ATen's order -- the gathered values are bit-identical to
// F.grid_sample / gather.hip (v_pk_mul_f32 + v_pk_add_f32: 34 instructions per pass); FC_FMA = 1 contracts each step into
// one v_pk_fma_f32 (16 per pass; one rounding per step instead of two: <= 1 ulp of the f32 sum closer to the exact value).
// Index arithmetic, weights and floor() are not contracted in either mode: corner indices stay bit-exact.
#define LDS_AS __attribute__((address_space(3)))
constexpr int GEO_N = 14;                     // per item: ezy[4], ex[2], wk[8]
constexpr int GEO_IPS = 64 / RPW;             // displacements per register set (64 items)
constexpr int GEO_SETS = (7 + GEO_IPS - 1) / GEO_IPS;
struct Geo { int v[GEO_SETS][GEO_N]; };

// rowb: the lane's row of the tile (lane % RPW of the wave's rows, clamped to the last point) | its sample << 8 -- the same
// for every level and round, computed once in front of the slab loop (a 64-bit pn / N per slab cost ~100 VALU)
template <bool BF, bool STAGED>
__device__ __forceinline__ void level_geometry(const FcLevel L, const float *__restrict__ pt0, int rowb, float disp, int ac, int lane,
                                               const FcBox box, Geo &G) {
  constexpr uint32_t EB = BF ? 2u : 4u;   // bytes per stored channel value
  const int C = L.C;
#pragma unroll
  for (int r = 0; r < GEO_SETS; ++r) {
    const int item = r * 64 + lane, j = min(item / RPW, 6);
    const int row = rowb & 255, b = rowb >> 8;
    // (uniform base + 32-bit lane offset: a per-lane 64-bit pointer would live -- spilled -- across the whole slab loop)
    const GLOBAL_AS float *pg = (const GLOBAL_AS float *)pt0;
    const uint32_t po = (uint32_t)row * 3u;
    const float pt[3] = {pg[po], pg[po + 1u], pg[po + 2u]};
    const Corner c = sample_corner(pt, j, disp, L.D, L.H, L.W, ac);
    const Weights w = corner_weights(c);
    int xc[2], yc[2], zc[2];
    bool vx[2], vy[2], vz[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      vx[a] = w.x0 + a >= 0 && w.x0 + a < L.W;
      vy[a] = w.y0 + a >= 0 && w.y0 + a < L.H;
      vz[a] = w.z0 + a >= 0 && w.z0 + a < L.D;
      xc[a] = min(max(w.x0 + a, 0), L.W - 1);
      yc[a] = min(max(w.y0 + a, 0), L.H - 1);
      zc[a] = min(max(w.z0 + a, 0), L.D - 1);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float wt = (w.wx[k & 1] * w.wy[(k >> 1) & 1]) * w.wz[k >> 2];
      G.v[r][6 + k] = __float_as_int((vx[k & 1] && vy[(k >> 1) & 1] && vz[k >> 2]) ? wt : 0.f);
    }
    if constexpr (STAGED) {   // byte offsets inside the staged box: [z][y][x][64 channels of this half]
      (void)b;
#pragma unroll
      for (int i = 0; i < 4; ++i) G.v[r][i] = (((zc[i >> 1] - box.z0) * box.by + (yc[i & 1] - box.y0)) * box.bx) * 256;
#pragma unroll
      for (int a = 0; a < 2; ++a) G.v[r][4 + a] = (xc[a] - box.x0) * 256;
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) G.v[r][i] = (int)((uint32_t)(((b * L.D + zc[i >> 1]) * L.H + yc[i & 1]) * L.W * C) * EB);  // host: < 2^30 elements
#pragma unroll
      for (int a = 0; a < 2; ++a) G.v[r][4 + a] = xc[a] * C * (int)EB;
    }
  }
}

// FC_GEO_LDS: the wave's RPW rows x 9 (axis, variant) evaluations, one per lane and round, into the geometry table.  Only this wave
// reads its rows' entries (LDS operations of a wave are ordered: no barrier).  Same operations in the same order as
// sample_corner / corner_weights, so corner indices and weights are bit-identical.
template <bool BF, bool STAGED>
__device__ __forceinline__ void level_geometry_lds(const FcLevel L, const float *__restrict__ pt0, int last, int b0, int rem0, int N,
                                                   float disp, int ac, int pw, int lane, const FcBox box, uint32_t *__restrict__ geo) {
#if defined(FC_DBG_WAIT) && (FC_DBG_WAIT & 2)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
  constexpr uint32_t EB = BF ? 2u : 4u;
  const GLOBAL_AS float *pg = (const GLOBAL_AS float *)pt0;
  // lane -> (row = lane % RPW, slot = lane / RPW); round r evaluates (axis, variant) number r (64 / RPW) + slot, the last round's
  // surplus slots redo number 8 (same values, same address): shifts and masks only
  constexpr int SLOTS = 64 / RPW, ROUNDS = (9 + SLOTS - 1) / SLOTS;
  const unsigned ul = (unsigned)lane & 63u;
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const int rl = (int)(ul % (unsigned)RPW);
    const int av = min(r * SLOTS + (int)(ul / (unsigned)RPW), 8), axis = (av * 11) >> 5, var = av - 3 * axis;
    const int row = min(RPW * pw + rl, last);
    float g = 2.0f * pg[(uint32_t)row * 3u + (uint32_t)(2 - axis)];
    if (var == 1) g = g + (-disp);
    if (var == 2) g = g + disp;
    const int S = axis == 0 ? L.W : (axis == 1 ? L.H : L.D);
    const float i = unnormalize(g, S, ac), i0f = floorf(i);
    const float w0 = (i0f + 1.0f) - i, w1 = i - i0f;
    const int i0 = clamp_int(i0f);
    const bool v0 = i0 >= 0 && i0 < S, v1 = i0 + 1 >= 0 && i0 + 1 < S;
    const int c0 = min(max(i0, 0), S - 1), c1 = min(max(i0 + 1, 0), S - 1);
    uint32_t o0, o1;
    if constexpr (STAGED) {   // byte offsets inside the staged box: [z][y][x][64 channels of this half]
      (void)b0; (void)rem0; (void)N;
      const int base = axis == 0 ? box.x0 : (axis == 1 ? box.y0 : box.z0);
      const int mul = axis == 0 ? 256 : (axis == 1 ? box.bx * 256 : box.by * box.bx * 256);
      o0 = (uint32_t)((c0 - base) * mul);
      o1 = (uint32_t)((c1 - base) * mul);
    } else {
      const int b = b0 + (int)((uint32_t)(rem0 + row) / (uint32_t)N);
      const uint32_t mul = axis == 0 ? (uint32_t)L.C : (axis == 1 ? (uint32_t)(L.W * L.C) : (uint32_t)(L.H * L.W * L.C));   // host: < 2^30 elements
      const uint32_t add = axis == 2 ? (uint32_t)(b * L.D) : 0u;
      o0 = (add + (uint32_t)c0) * mul * EB;
      o1 = (add + (uint32_t)c1) * mul * EB;
    }
    // (stored with the TYPES they are read back with in produce_slab -- u32x2_t offsets, f32x2_t weights: a float load may be
    // moved across an unsigned store to the same address under strict aliasing, and was: wrong rows, run to run different)
    LDS_AS char *d = (LDS_AS char *)geo + ((RPW * pw + rl) * GEO_ROW_DW + av * 2) * 4;
    *reinterpret_cast<LDS_AS u32x2_t *>(d) = u32x2_t{o0, o1};
    *reinterpret_cast<LDS_AS f32x2_t *>(d + 72) = f32x2_t{v0 ? w0 : 0.f, v1 ? w1 : 0.f};
  }
#if defined(FC_DBG_WAIT) && (FC_DBG_WAIT & 1)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#else
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the table is complete before this wave's passes read it
#endif
#ifdef SVR_FC0_MEASURE
  // debug dump (fc_stamps set, first tile only): level L.C == 64's table of this wave -> buffer rows [pw][16][36]
  if (fc_stamps && pt0 == fc_dbg_pt0 && L.C == fc_dbg_C) {
    uint32_t *dst = reinterpret_cast<uint32_t *>(fc_stamps) + pw * RPW * GEO_ROW_DW;
    for (int i = lane; i < RPW * GEO_ROW_DW; i += 64) dst[i] = geo[RPW * pw * GEO_ROW_DW + i];
  }
#endif
}

template <int LP, int NJ, bool BF, bool STAGED = false>
__device__ __forceinline__ void produce_slab(const FcLevel L, const FcSlab S, uint32_t *__restrict__ buf, int64_t m0, int64_t M,
                                             float *__restrict__ feat, int row_stride, int pw, int lane, const Geo &G,
                                             const uint32_t *stage = nullptr, const uint32_t *geo = nullptr) {
  constexpr int PPW0 = 64 / LP, PPW = PPW0 < RPW ? PPW0 : RPW, NP = RPW / PPW, LPI = LP / NJ, NC = LPI * 4, DEPTH = NP <= FC_DEPTH ? NP : (FC_DEPTH < 3 ? FC_DEPTH : 3);  // passes in flight (the fine levels miss the caches: all of a slab's passes)
  const int C = L.C;
  const GLOBAL_AS char *vol = (const GLOBAL_AS char *)L.vol;
  constexpr uint32_t EB = BF ? 2u : 4u;   // bytes per stored channel value
  // the register set that holds this slab's displacements (j0 is even for NJ == 2: both sit in one set)
  int ge[GEO_N];
  if constexpr (!FC_GEO_LDS || (FC_GEO_CHECK & 1)) {
#pragma unroll
    for (int k = 0; k < GEO_N; ++k) ge[k] = G.v[0][k];
#pragma unroll
    for (int r = 1; r < GEO_SETS; ++r)
      if (S.j0 / GEO_IPS == r) {   // (wave uniform)
#pragma unroll
        for (int k = 0; k < GEO_N; ++k) ge[k] = G.v[r][k];
      }
  }
  // (lane is opaque to the compiler here -- gather_fc0_kernel hides it per slab -- so its range is stated: unsigned shifts / masks
  // instead of the signed-division sequences with 16-bit SDWA pieces that the unknown range produced)
  const unsigned ul = (unsigned)lane & 63u;
  const int g = (int)(ul / (unsigned)LP), q = (int)(ul % (unsigned)LP), jj = (int)((unsigned)q / (unsigned)LPI), c4 = (int)((unsigned)q % (unsigned)LPI) * 4;
  // FC_GEO_LDS: byte addresses of this lane's three table entries (the displacement moves one axis) in row 0 of the wave
  const int jd = S.j0 + jj;
  const int gx8 = (jd == 1 ? 1 : (jd == 2 ? 2 : 0)) * 8, gy8 = (3 + (jd == 3 ? 1 : (jd == 4 ? 2 : 0))) * 8,
            gz8 = (6 + (jd == 5 ? 1 : (jd == 6 ? 2 : 0))) * 8;
  const LDS_AS char *grow = (const LDS_AS char *)geo + (RPW * pw + (g < RPW ? g : RPW - 1)) * (GEO_ROW_DW * 4);
  const int src0 = (((S.j0 + jj) % GEO_IPS) * RPW + g) << 2;   // + it * PPW rows
  const uint32_t xoff = STAGED ? (uint32_t)c4 * 4u : (uint32_t)(S.c0 + c4) * EB;
  const int col = jj * NC + c4;  // column inside the slab
  // row of pass `it` = RPW pw + g + it PPW: its swizzle bit (bit 3) is known at compile time when PPW <= 8 (g < PPW), else
  // (PPW = 16) it is bit 3 of g: at most one v_xor per pass
  static_assert((RPW == 8 || RPW % 16 == 0) && (PPW == 4 || PPW == 8 || PPW == 16), "swizzle bit of the producer rows");
  // (FKSTEP, FLW and the pass stride are multiples of 8 dwords: bit 2 of the dword offset IS the slot bit, `^ 4` toggles it)
  static_assert(FKSTEP % 8 == 0 && FLW == 8, "slot bit of the producer's destination");
  // RPW == 8: all rows of the wave share bit 3 (= bit 0 of pw)
  const int dsto = ((col >> 4) * FKSTEP + (RPW * pw + g) * FLW + ((col & 15) >> 1)) ^
                   (RPW == 8 ? (pw & 1) * 4 : (PPW == 16 ? ((g >> 3) & 1) * 4 : 0));
  const bool own = PPW0 <= RPW || g < RPW;   // (a lane group beyond the wave's rows -- LP = 4 at RPW = 8 -- stores nothing)
  GLOBAL_AS char *featt = (GLOBAL_AS char *)(feat + m0 * row_stride);
  const uint32_t fo0 = (uint32_t)((RPW * pw + g) * row_stride + L.kcol + (S.j0 + jj) * C + S.c0 + c4) * 4u;
  const int live = M - m0 < FTM ? (int)(M - m0) : FTM;
  struct Iter {
    f32x4 v[8];   // (only the loaded values are in flight: the weights are fetched by finish(), so a second pass in flight
  };              //  costs 32 registers, not 40)
  auto fetch = [&](Iter &I, int it) {
    const int src = src0 + it * (PPW * 4);
    uint32_t zy[4], x[2];
    if constexpr (FC_GEO_LDS) {
      const LDS_AS char *gp = grow + it * (PPW * GEO_ROW_DW * 4);
      const u32x2_t xo = *reinterpret_cast<const LDS_AS u32x2_t *>(gp + gx8), yo = *reinterpret_cast<const LDS_AS u32x2_t *>(gp + gy8),
                    zo = *reinterpret_cast<const LDS_AS u32x2_t *>(gp + gz8);
      zy[0] = zo.x + yo.x; zy[1] = zo.x + yo.y; zy[2] = zo.y + yo.x; zy[3] = zo.y + yo.y;
      x[0] = xo.x + xoff; x[1] = xo.y + xoff;
#if FC_GEO_CHECK == 2 && defined(SVR_FC0_MEASURE)
      {   // read the three entries AGAIN (nothing may write them in between): differences = a concurrent writer
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const LDS_AS char *gp2 = gp;
        asm volatile("" : "+v"(gp2));
        const u32x2_t xo2 = *reinterpret_cast<const LDS_AS u32x2_t *>(gp2 + gx8), yo2 = *reinterpret_cast<const LDS_AS u32x2_t *>(gp2 + gy8),
                      zo2 = *reinterpret_cast<const LDS_AS u32x2_t *>(gp2 + gz8);
        const uint32_t a6[6] = {xo.x, xo.y, yo.x, yo.y, zo.x, zo.y}, b6[6] = {xo2.x, xo2.y, yo2.x, yo2.y, zo2.x, zo2.y};
#pragma unroll
        for (int i = 0; i < 6; ++i)
          if (a6[i] != b6[i] && fc_stamps) {
            const unsigned long long n = atomicAdd(fc_stamps, 1ull);
            if (n < 200) {
              unsigned long long *o = fc_stamps + 8 + n * 4;
              o[0] = ((unsigned long long)blockIdx.x << 32) | (unsigned)(S.level << 24 | pw << 16 | lane << 8 | it << 4 | i);
              o[1] = ((unsigned long long)a6[i] << 32) | b6[i];
              o[2] = ((unsigned long long)(unsigned)S.j0 << 32) | (unsigned)(LP << 8 | NJ);
              o[3] = 0;
            }
          }
      }
#endif
#if (FC_GEO_CHECK == 1 || FC_GEO_CHECK == 3) && defined(SVR_FC0_MEASURE)
      {
        uint32_t r6[6];
#pragma unroll
        for (int i = 0; i < 4; ++i) r6[i] = (uint32_t)__builtin_amdgcn_ds_bpermute(src, ge[i]);
#pragma unroll
        for (int a = 0; a < 2; ++a) r6[4 + a] = (uint32_t)__builtin_amdgcn_ds_bpermute(src, ge[4 + a]) + xoff;
        const uint32_t l6[6] = {zy[0], zy[1], zy[2], zy[3], x[0], x[1]};
#pragma unroll
        for (int i = 0; i < 6; ++i)
          if (l6[i] != r6[i] && fc_stamps && (PPW0 <= RPW || g < RPW)) {
            const unsigned long long n = atomicAdd(fc_stamps, 1ull);
            if (n < 200) {
              unsigned long long *o = fc_stamps + 8 + n * 4;
              o[0] = ((unsigned long long)blockIdx.x << 32) | (unsigned)(S.level << 24 | pw << 16 | lane << 8 | it << 4 | i);
              o[1] = ((unsigned long long)l6[i] << 32) | r6[i];
              o[2] = ((unsigned long long)(unsigned)S.j0 << 32) | (unsigned)(LP << 8 | NJ);
              o[3] = 0;
            }
          }
        if constexpr (FC_GEO_CHECK == 1) {
#pragma unroll
          for (int i = 0; i < 4; ++i) zy[i] = r6[i];
          x[0] = r6[4]; x[1] = r6[5];
        }
      }
#endif
    } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) zy[i] = (uint32_t)__builtin_amdgcn_ds_bpermute(src, ge[i]);
#pragma unroll
    for (int a = 0; a < 2; ++a) x[a] = (uint32_t)__builtin_amdgcn_ds_bpermute(src, ge[4 + a]) + xoff;
    }
    if constexpr (BF) {   // four bf16 channels = one 8-byte load; widened to f32 exactly (bf16 = the upper half of an f32)
      u32x2_t raw[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) raw[k] = *reinterpret_cast<const GLOBAL_AS u32x2_t *>(vol + (zy[k >> 1] + x[k & 1]));
#pragma unroll
      for (int k = 0; k < 8; ++k)
        I.v[k] = f32x4{__uint_as_float(raw[k].x << 16), __uint_as_float(raw[k].x & 0xffff0000u), __uint_as_float(raw[k].y << 16),
                       __uint_as_float(raw[k].y & 0xffff0000u)};
    } else if constexpr (STAGED) {
      const LDS_AS char *sb = (const LDS_AS char *)stage;
#pragma unroll
      for (int k = 0; k < 8; ++k) I.v[k] = *reinterpret_cast<const LDS_AS f32x4 *>(sb + (zy[k >> 1] + x[k & 1]));
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) I.v[k] = *reinterpret_cast<const GLOBAL_AS f32x4 *>(vol + (zy[k >> 1] + x[k & 1]));
    }
  };
  auto finish = [&](const Iter &I, int it) {
    float w[8];
    if constexpr (FC_GEO_LDS) {   // (wx wy) wz in ATen's order, from the three per-axis pairs (a pair is 0 outside the volume)
      const LDS_AS char *gp = grow + it * (PPW * GEO_ROW_DW * 4) + 72;
#if defined(FC_DBG_WAIT) && (FC_DBG_WAIT & 4)
      asm volatile("" : "+v"(gp));     // (debug: keeps the weight reads from being merged with the offset reads into ds_read2_b64)
#endif
      const f32x2_t wx = *reinterpret_cast<const LDS_AS f32x2_t *>(gp + gx8), wy = *reinterpret_cast<const LDS_AS f32x2_t *>(gp + gy8),
                    wz = *reinterpret_cast<const LDS_AS f32x2_t *>(gp + gz8);
      const float wxy[4] = {wx.x * wy.x, wx.y * wy.x, wx.x * wy.y, wx.y * wy.y};
#pragma unroll
      for (int k = 0; k < 8; ++k) w[k] = wxy[k & 3] * (k < 4 ? wz.x : wz.y);
#if (FC_GEO_CHECK == 1 || FC_GEO_CHECK == 3) && defined(SVR_FC0_MEASURE)
      {
        const int src = src0 + it * (PPW * 4);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float wr = __int_as_float(__builtin_amdgcn_ds_bpermute(src, ge[6 + k]));
          if (__float_as_uint(wr) != __float_as_uint(w[k]) && !(wr == 0.f && w[k] == 0.f) && fc_stamps && (PPW0 <= RPW || g < RPW)) {
            const unsigned long long n = atomicAdd(fc_stamps, 1ull);
            if (n < 200) {
              unsigned long long *o = fc_stamps + 8 + n * 4;
              o[0] = ((unsigned long long)blockIdx.x << 32) | (unsigned)(S.level << 24 | pw << 16 | lane << 8 | it << 4 | (8 + k));
              o[1] = ((unsigned long long)__float_as_uint(w[k]) << 32) | __float_as_uint(wr);
              o[2] = ((unsigned long long)(unsigned)S.j0 << 32) | (unsigned)(LP << 8 | NJ);
              o[3] = ((unsigned long long)__float_as_uint(wz.x) << 32) | __float_as_uint(wz.y);
            }
          }
          if constexpr (FC_GEO_CHECK == 1) w[k] = wr;
        }
      }
#endif
    } else {
      const int src = src0 + it * (PPW * 4);
#pragma unroll
      for (int k = 0; k < 8; ++k) w[k] = __int_as_float(__builtin_amdgcn_ds_bpermute(src, ge[6 + k]));
    }
    f32x4 acc;
    if constexpr (FC_PK == 0 && !BF) {   // plain f32 instructions: same values, same order
      float a[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if constexpr (FC_FMA != 0) {
          a[c] = I.v[0][c] * w[0];
#pragma unroll
          for (int k = 1; k < 8; ++k) a[c] = __builtin_fmaf(I.v[k][c], w[k], a[c]);
        } else {
          a[c] = 0.f;
#pragma unroll
          for (int k = 0; k < 8; ++k) a[c] = a[c] + I.v[k][c] * w[k];
        }
        asm volatile("" : "+v"(a[c]));   // (keeps the SLP vectoriser from packing the four chains again)
      }
      acc = f32x4{a[0], a[1], a[2], a[3]};
    } else if constexpr (FC_FMA != 0 && !BF) {   // one rounding per step (v_pk_fma_f32)
      acc = I.v[0] * w[0];
#pragma unroll
      for (int k = 1; k < 8; ++k) {
        const f32x2_t wk = {w[k], w[k]};
        const f32x2_t lo = __builtin_elementwise_fma(wk, f32x2_t{I.v[k].x, I.v[k].y}, f32x2_t{acc.x, acc.y});
        const f32x2_t hi = __builtin_elementwise_fma(wk, f32x2_t{I.v[k].z, I.v[k].w}, f32x2_t{acc.z, acc.w});
        acc = f32x4{lo.x, lo.y, hi.x, hi.y};
      }
    } else {
      acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 8; ++k) acc = acc + I.v[k] * w[k];
    }
    uint32_t *d = buf + (dsto ^ ((PPW == 16 || RPW == 8) ? 0 : (((it * PPW) >> 3) & 1) * 4)) + it * (PPW * FLW);
    if (!own) return;
    if constexpr (BF) {   // the feature values in bf16 (one rounding), one plane
      *reinterpret_cast<uint2 *>(d) = make_uint2(pack_bf16_rne(acc.x, acc.y), pack_bf16_rne(acc.z, acc.w));
      return;
    }
    uint32_t h0, l0, h1, l1;
    if constexpr (FC_PK == 0) {   // split_x with plain f32 instructions (same values)
      h0 = pack_f16(acc.x, acc.y);
      h1 = pack_f16(acc.z, acc.w);
      const f32x2 u0 = unpack_f16(h0), u1 = unpack_f16(h1);
      float r[4] = {(acc.x - u0.x) * 2048.f, (acc.y - u0.y) * 2048.f, (acc.z - u1.x) * 2048.f, (acc.w - u1.y) * 2048.f};
#pragma unroll
      for (int c = 0; c < 4; ++c) asm volatile("" : "+v"(r[c]));
      l0 = pack_f16(r[0], r[1]);
      l1 = pack_f16(r[2], r[3]);
    } else {
      split_x(acc.x, acc.y, h0, l0);
      split_x(acc.z, acc.w, h1, l1);
    }
    *reinterpret_cast<uint2 *>(d) = make_uint2(h0, h1);
    *reinterpret_cast<uint2 *>(d + FPLANE) = make_uint2(l0, l1);
    if (S.keep && RPW * pw + it * PPW + g < live) {
#if FC_KEEP_NT
      // streaming store: the kept columns (1.28 GB) are read once, by the backward -- they should not push volumes out of L2
      __builtin_nontemporal_store(acc, reinterpret_cast<GLOBAL_AS f32x4 *>(featt + (fo0 + (uint32_t)(it * PPW * row_stride * 4))));
#else
      *reinterpret_cast<GLOBAL_AS f32x4 *>(featt + (fo0 + (uint32_t)(it * PPW * row_stride * 4))) = acc;
#endif
    }
  };
  Iter I[DEPTH];
#pragma unroll
  for (int it = 0; it < DEPTH; ++it) fetch(I[it], it);
#pragma unroll
  for (int it = 0; it < NP; ++it) {
    finish(I[it % DEPTH], it);
    if (it + DEPTH < NP) fetch(I[it % DEPTH], it + DEPTH);
  }
}

// The C == 1 level (the raw input grid): 7 columns + 9 zero columns, one k-step.  One (row, displacement) item per
// producer thread and round (4 rounds of 256; "displacement 7" stands for the zero columns), all 8 x 4 loads in flight.
template <bool BF>
__device__ __forceinline__ void produce_c1(const FcLevel L, const FcSlab S, uint32_t *__restrict__ buf,
                                           const float *__restrict__ points, int64_t m0, int64_t M, int N, float disp, int ac,
                                           float *__restrict__ feat, int row_stride, int tp) {
  const GLOBAL_AS float *vol = (const GLOBAL_AS float *)L.vol;
  const GLOBAL_AS uint16_t *vol16 = (const GLOBAL_AS uint16_t *)L.vol;
  constexpr int C1R = FTM * 8 / NPT;  // rounds of NPT (row, displacement) items
  static_assert(C1R >= 1 && FTM * 8 % NPT == 0, "C == 1 level: whole rounds of producer threads");
  float u[C1R][8], wk[C1R][8];
#pragma unroll
  for (int r = 0; r < C1R; ++r) {
    const int item = r * NPT + tp, row = item >> 3, j = item & 7;
    const int64_t pn = min(m0 + row, M - 1);
    const uint32_t vb = (uint32_t)(pn / N) * (uint32_t)(L.D * L.H * L.W);
    const Corner c = sample_corner(points + pn * 3, j < 7 ? j : 0, disp, L.D, L.H, L.W, ac);
    const Weights w = corner_weights(c);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int z = w.z0 + (k >> 2), y = w.y0 + ((k >> 1) & 1), x = w.x0 + (k & 1);
      const bool valid = z >= 0 && z < L.D && y >= 0 && y < L.H && x >= 0 && x < L.W && j < 7;
      const int zc = min(max(z, 0), L.D - 1), yc = min(max(y, 0), L.H - 1), xc = min(max(x, 0), L.W - 1);
      wk[r][k] = valid ? (w.wx[k & 1] * w.wy[(k >> 1) & 1]) * w.wz[k >> 2] : 0.f;
      const uint32_t vo = vb + (uint32_t)((zc * L.H + yc) * L.W + xc);
      u[r][k] = BF ? __uint_as_float((uint32_t)vol16[vo] << 16) : vol[vo];
    }
  }
  uint16_t *b16 = reinterpret_cast<uint16_t *>(buf);
#pragma unroll
  for (int r = 0; r < C1R; ++r) {
    const int item = r * NPT + tp, row = item >> 3, j = item & 7;
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) acc = acc + u[r][k] * wk[r][k];
    if (j == 7) acc = 0.f;  // (a volume with non-finite values must not leak into the zero columns)
    const _Float16 h = (_Float16)acc;
    const _Float16 l = (_Float16)((acc - (float)h) * 2048.f);
    const uint16_t hb = BF ? (uint16_t)(pack_bf16_rne(acc, 0.f) & 0xffffu) : __builtin_bit_cast(uint16_t, h);
    const int s0 = fc_slot_dw(row, 0), s1 = fc_slot_dw(row, 1);   // halves 0 .. 7 / 8 .. 15 of the row
    if (j < 7) {
      b16[s0 * 2 + j] = hb;
      if (!BF) b16[(FPLANE + s0) * 2 + j] = __builtin_bit_cast(uint16_t, l);
      if (!BF && S.keep && m0 + row < M) feat[(m0 + row) * row_stride + L.kcol + j] = acc;
    } else {  // halves 7 .. 15 of the row: zeros
      uint32_t z = 0u;
      asm volatile("" : "+v"(z));   // (a hoisted uint4 of zeros was kept -- spilled -- across the whole slab loop)
      b16[s0 * 2 + 7] = (uint16_t)z;
      if (!BF) b16[(FPLANE + s0) * 2 + 7] = (uint16_t)z;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        buf[s1 + p] = z;
        if (!BF) buf[FPLANE + s1 + p] = z;
      }
    }
  }
}

// STAGE slab: the tile's bounding box of level L, channels [c0, c0 + 64), into the staging region: one float4 per lane and
// round (16 lanes = one voxel), all loads of the thread issued first.  tp = producer thread (0 .. 255).
__device__ __forceinline__ void stage_box(const FcLevel L, int c0, const FcBox box, uint32_t *__restrict__ stage, int tp) {
  if (box.nvox == 0) return;   // (uniform) the tile keeps its global loads
  const GLOBAL_AS char *vol = (const GLOBAL_AS char *)L.vol;
  constexpr int R = FC_STAGE_VOX * 16 / NPT;
  static_assert(FC_STAGE_VOX * 16 % NPT == 0, "staging rounds");
  const int n16 = box.nvox * 16, byx = box.by * box.bx;
  const float ibyx = 1.f / (float)byx, ibx = 1.f / (float)box.bx;
  f32x4 reg[R];
#pragma unroll
  for (int i = 0; i < R; ++i) {
    const int idx = min(tp + NPT * i, n16 - 1), v = idx >> 4, q = idx & 15;
    const int vz = (int)(((float)v + 0.5f) * ibyx), r = v - vz * byx;   // exact for v, byx <= 160
    const int vy = (int)(((float)r + 0.5f) * ibx), vx = r - vy * box.bx;
    const uint32_t off = (uint32_t)((((box.b * L.D + box.z0 + vz) * L.H + box.y0 + vy) * L.W + box.x0 + vx) * L.C + c0 + q * 4) * 4u;
    reg[i] = *reinterpret_cast<const GLOBAL_AS f32x4 *>(vol + off);
  }
#pragma unroll
  for (int i = 0; i < R; ++i) {
    const int idx = tp + NPT * i;
    if (idx < n16) *reinterpret_cast<f32x4 *>(stage + idx * 4) = reg[i];   // [voxel][64 channels]: idx * 4 dwords
  }
}

template <bool BF>
__device__ __forceinline__ void produce(const FcArgs &A, int s, uint32_t *buf, const float *points, int64_t m0, int64_t M,
                                        int N, float disp, int ac, float *feat, int row_stride, int pw, int lane, int dbg,
                                        uint32_t *stage, const FcBox *__restrict__ boxes, Geo &G, int rowb, uint32_t *geo, int last,
                                        int b0, int rem0) {
  const FcSlab S = A.S[s];
  const FcLevel L = A.L[S.level];
  if constexpr (!BF) {
    if (S.stage >= 0) {   // (uniform) a staged level: the STAGE slab itself, or one of its displacement slabs
      const FcBox box = boxes[S.stage];
      if (S.lp < 0) {
        stage_box(L, S.c0, box, stage, pw * 64 + lane);
        return;
      }
      if (box.nvox > 0) {
        if (S.geo) {
          if constexpr (FC_GEO_LDS) level_geometry_lds<BF, true>(L, points + m0 * 3, last, b0, rem0, N, disp, ac, pw, lane, box, geo);
          if constexpr (!FC_GEO_LDS || (FC_GEO_CHECK & 1)) level_geometry<BF, true>(L, points + m0 * 3, rowb, disp, ac, lane, box, G);
        }
        produce_slab<16, 1, BF, true>(L, S, buf, m0, M, feat, row_stride, pw, lane, G, stage, geo);
        return;
      }
    }
  }
#ifdef SVR_FC0_MEASURE
  if ((dbg >> (8 + S.level)) & 1) return;
#else
  (void)dbg;
#endif
  if (S.lp == 0) {
    produce_c1<BF>(L, S, buf, points, m0, M, N, disp, ac, feat, row_stride, pw * 64 + lane);
    return;
  }
  if (S.geo) {
    if constexpr (FC_GEO_LDS) level_geometry_lds<BF, false>(L, points + m0 * 3, last, b0, rem0, N, disp, ac, pw, lane, FcBox{}, geo);
    if constexpr (!FC_GEO_LDS || (FC_GEO_CHECK & 1)) level_geometry<BF, false>(L, points + m0 * 3, rowb, disp, ac, lane, FcBox{}, G);
  }
  switch (S.lp * 4 + S.nj) {
    case 16 * 4 + 1: produce_slab<16, 1, BF>(L, S, buf, m0, M, feat, row_stride, pw, lane, G, nullptr, geo); break;
    case 16 * 4 + 2: produce_slab<16, 2, BF>(L, S, buf, m0, M, feat, row_stride, pw, lane, G, nullptr, geo); break;
    case 8 * 4 + 1: produce_slab<8, 1, BF>(L, S, buf, m0, M, feat, row_stride, pw, lane, G, nullptr, geo); break;
    case 8 * 4 + 2: produce_slab<8, 2, BF>(L, S, buf, m0, M, feat, row_stride, pw, lane, G, nullptr, geo); break;
    case 4 * 4 + 1: produce_slab<4, 1, BF>(L, S, buf, m0, M, feat, row_stride, pw, lane, G, nullptr, geo); break;
  }
}

__device__ __forceinline__ void slab_barrier() {  // LDS stores of this wave done, then the workgroup barrier
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// The slab table is read from DEVICE memory (args_store_kernel copies the by-value struct there): kernel arguments live in
// uncached host-visible memory, and one dynamically indexed s_load from them per slab cost ~0.8 us (0.4 ms per launch).
__global__ void args_store_kernel(FcArgs A, FcArgs *__restrict__ dst) {
  const uint32_t *src = reinterpret_cast<const uint32_t *>(&A);
  for (unsigned i = threadIdx.x; i < sizeof(FcArgs) / 4; i += blockDim.x) reinterpret_cast<uint32_t *>(dst)[i] = src[i];
}

// `dbg` is a constant 0 unless the file is built with -DSVR_FC0_MEASURE (then: environment SVR_FC0_DBG), which switches
// parts of the kernel OFF for measurements -- the results are then wrong by construction: bit 0 producers idle, bit 1
// consumers idle, bit 2 no epilogue, bit 3 return at once, bit 8 + l: level l is not gathered (tools/exp/prof_fc0.sh;
// DESIGN.md section 5b quotes the numbers).
#ifdef SVR_FC0_MEASURE
#define FC_DBG(x) (x)
// In-kernel timeline (measurement builds only; MI355X_MICROARCH.md "In-kernel stamps"): lane 0 of every wave of the tiles
// [FC_ST_TILE0, FC_ST_TILE0 + FC_ST_TILES) appends (event id << 56 | slab << 48 | s_memtime) to its own row of a buffer that
// nothing else reads (svr_gather_fc0_stamps sets the pointer; tools/exp/fc0_timeline.py reads it back).

constexpr int FC_ST_TILE0 = 3000, FC_ST_TILES = 8, FC_ST_N = 1024;
#define FC_STAMP_INIT()                                                                                          \
  unsigned long long *st_p = nullptr;                                                                            \
  int st_i = 0;                                                                                                  \
  if (fc_stamps && tile >= FC_ST_TILE0 && tile < FC_ST_TILE0 + FC_ST_TILES && lane == 0)                        \
    st_p = fc_stamps + ((tile - FC_ST_TILE0) * 8 + wave) * FC_ST_N;
#define FC_STAMP(id, slab)                                                                                       \
  do {                                                                                                           \
    if (st_p && st_i < FC_ST_N)                                                                                  \
      st_p[st_i++] = ((unsigned long long)(id) << 56) | ((unsigned long long)((slab) & 255) << 48) |             \
                     (__builtin_amdgcn_s_memtime() & 0xffffffffffffull);                                         \
  } while (0)
#else
#define FC_DBG(x) 0
#define FC_STAMP_INIT()
#define FC_STAMP(id, slab)
#endif
template <bool BF>
#ifndef FC_MINWAVES
#define FC_MINWAVES (FC_NPW == 8 ? 3 : (FTM == 64 ? 4 : 2))
#endif
__global__ __launch_bounds__(NTHR, FC_MINWAVES) void gather_fc0_kernel(const FcArgs *__restrict__ Ap, const float *__restrict__ points,
                                                            const uint16_t *__restrict__ W0,
                                                            const uint32_t *__restrict__ amax, const float *__restrict__ bias,
                                                            float *__restrict__ Y, int64_t ldy, float *__restrict__ feat,
                                                            int row_stride, int pad_start, int64_t M, int N, float disp, int ac,
                                                            int relu, int dbg_arg, const FcBox *__restrict__ boxes, int xcd) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const int dbg = FC_DBG(dbg_arg);
  const FcArgs &A = *Ap;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  // tile of this workgroup: the block id, or (xcd: grid padded to a multiple of 8) a tile order that is contiguous per XCD
  const int64_t tile = xcd ? xcd_logical(blockIdx.x, gridDim.x) : (int64_t)blockIdx.x;
  const int64_t m0 = tile * FTM;
  if (m0 >= M) return;   // (whole workgroup, in front of the first barrier)
  const int S = A.n_slabs;
  if (dbg & 8) return;
  FC_STAMP_INIT();
  FC_STAMP(0, 0);
#ifdef FC_NO_PRODUCER
  if (wave >= 4) return;
#endif
  if (wave >= 4) {
    // ------------------------------------------------------------------ producers
    const int pw = __builtin_amdgcn_readfirstlane(wave) - 4;   // (uniform: everything derived from it is scalar arithmetic)
    if constexpr (FC_PRIO != 0) __builtin_amdgcn_s_setprio(FC_PRIO);
    if (pad_start >= 0 && t - 256 < FTM && m0 + (t - 256) < M)  // kept rows: the padding columns behind the last level are zeros
      for (int cc = pad_start; cc < row_stride; ++cc) feat[(m0 + (t - 256)) * row_stride + cc] = 0.f;
    // slab s + 1 is produced into the buffer the consumers are not reading, then the barrier hands both over (ONE call site:
    // with a second, peeled call for slab 0 the compiler inlined all five slab shapes twice and spilled 268 B / lane)
    Geo G;   // the sample geometry of the level in work (level_geometry), alive across that level's slabs (register form)
    int rowb = 0;
    const int last = (int)min<int64_t>(M - 1 - m0, FTM - 1);     // last valid row of the tile
    const int b0 = (int)(m0 / N), rem0 = (int)(m0 - (int64_t)b0 * N);   // (uniform) sample of row 0 and its position in it
    uint32_t *geo = lds + 2 * FSLAB + FC_STAGE_DW;
    if constexpr (!FC_GEO_LDS || (FC_GEO_CHECK & 1)) {
      static_assert(FTM <= 256, "row in the low byte of rowb");
      const int row = min(RPW * pw + lane % RPW, last);
      rowb = row | ((b0 + (int)((uint32_t)(rem0 + row) / (uint32_t)N)) << 8);
      asm volatile("" : "+v"(rowb));    // one register, not its recomputable pieces hoisted and spilled
    }
    for (int s = -1; s < S; ++s) {
      // (the lane index is made opaque per slab: with seven slab shapes the lane-derived constants of ALL of them were hoisted
      // in front of this loop and 12 of them spilled; recomputed per slab they cost a few dozen VALU instructions)
      int lane_s = lane;
      int rowb_s = rowb;
      asm volatile("" : "+v"(lane_s), "+v"(rowb_s));   // (rowb too: the point address derived from it was hoisted and spilled)
      FC_STAMP(1, s + 1);
      if (s + 1 < S && !(dbg & 1))
        produce<BF>(A, s + 1, lds + ((s + 1) & 1) * FSLAB, points, m0, M, N, disp, ac, feat, row_stride, pw, lane_s, dbg, lds + 2 * FSLAB,
                    boxes + tile * FC_NSTAGE, G, rowb_s, geo, last, b0, rem0);
      FC_STAMP(2, s + 1);
      slab_barrier();
      FC_STAMP(3, s + 1);
    }
    return;
  }
#ifdef FC_NO_CONSUMER   // (register-count experiments: what do the producers need on their own?)
  return;
#endif
  // -------------------------------------------------------------------- consumers: wave wc -> columns [64 wc, 64 wc + 64)
  const int wc = wave, l31 = lane & 31, lh = lane >> 5;
  const int KF = A.KF, nk = KF / FK;
  const uint16_t *wp = W0 + ((2 * wc) * 64 + lane) * 8;  // + jt * 512 halves, + plane * 8 * 512, + k-step * 2 * 8 * 512
  // W fragments of four k-steps in registers, rotating by NAME (a copy would have to wait for the load it moves)
  uint4 b0[2][2], b1[2][2], b2[2][2], b3[2][2];  // [tile][hi / lo]
  constexpr int NPL = BF ? 1 : 2;   // W planes per k-step
  auto loadb = [&](uint4 (&r)[2][2], int kidx) {
    const uint16_t *q = wp + (kidx < nk ? kidx : nk - 1) * (NPL * (FTN / 32) * 512);
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
      r[jt][0] = *reinterpret_cast<const uint4 *>(q + jt * 512);
      if constexpr (!BF) r[jt][1] = *reinterpret_cast<const uint4 *>(q + (FTN / 32) * 512 + jt * 512);
    }
  };
  f32x16 acc[FMT][2];
#pragma unroll
  for (int i = 0; i < FMT; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  constexpr int BDIST = (FTM == 64 && NPW == 4) ? 1 : 3;  // k-steps the W loads run ahead (2 or 4 register sets)
  loadb(b0, 0);
  if constexpr (BDIST == 3) {
    loadb(b1, 1);
    loadb(b2, 2);
  }
  slab_barrier();  // slab 0 is in LDS
  int s = 0, kin = 0, ksl = A.S[0].nk, kidx = 0;
  // a slab without k-steps (a STAGE slab) is one more hand-over and nothing else
  auto skip_empty = [&]() {
    while (s < S && ksl == 0) {
      slab_barrier();
      ++s;
      ksl = s < S ? A.S[s].nk : 0;
    }
  };
  skip_empty();
  // `mma_on`: an always-true scalar the compiler cannot see through.  With the k-step body unconditional the register
  // allocator spills 260 B / lane of the accumulators (one basic block for the whole unrolled loop: the launch takes 11.7
  // instead of 2.6 ms); behind a uniform branch -- which is how the kernel was developed, the branch used to be a
  // measurement switch -- it keeps 128 VGPRs without scratch.  One s_cbranch per k-step.
  int mma_on = !(dbg & 2);
  asm volatile("" : "+s"(mma_on));
  // k-step kidx on the fragments `cur`; `fre` (used one step ago) is refilled with step kidx + 3
  auto step = [&](const uint4 (&cur)[2][2], uint4 (&fre)[2][2]) {
    FC_STAMP(20, s);
    if (mma_on) {
      loadb(fre, kidx + BDIST);
      const uint32_t *pa = lds + (s & 1) * FSLAB + kin * FKSTEP;
      if constexpr (BF) {   // one bf16 product per block
#pragma unroll
        for (int i = 0; i < FMT; ++i) {
          union { f16x8 h; bf16x8_t b; } ua;
          ua.h = lds_frag(pa, i * 32 + l31, lh);
#pragma unroll
          for (int jt = 0; jt < 2; ++jt) {
            union { uint4 q; bf16x8_t b; } ub;
            ub.q = cur[jt][0];
            acc[i][jt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ua.b, ub.b, acc[i][jt], 0, 0, 0);
          }
        }
      } else {
      f16x8 b[3][2];
#pragma unroll
      for (int jt = 0; jt < 2; ++jt) {
        union { uint4 q; f16x8 v; } u0, u1;
        u0.q = cur[jt][0];
        u1.q = cur[jt][1];
        b[0][jt] = u0.v;
        b[1][jt] = u1.v;
        b[2][jt] = scale_2m11(u0.v);
      }
#pragma unroll
      for (int i = 0; i < FMT; ++i) {
        const f16x8 ah = lds_frag(pa, i * 32 + l31, lh), al = lds_frag(pa + FPLANE, i * 32 + l31, lh);
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
          acc[i][jt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, b[2][jt], acc[i][jt], 0, 0, 0);  // lo(x) hi(w)
          acc[i][jt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b[1][jt], acc[i][jt], 0, 0, 0);  // hi(x) lo(w)
          acc[i][jt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b[0][jt], acc[i][jt], 0, 0, 0);  // hi(x) hi(w)
        }
      }
      }
    }
    ++kidx;
    if (++kin == ksl) {  // slab consumed: hand the buffer back, the next one is ready behind the barrier
      FC_STAMP(21, s);
      slab_barrier();
      FC_STAMP(22, s);
      ++s;
      kin = 0;
      ksl = s < S ? A.S[s].nk : 0;
      skip_empty();
    }
  };
  while (kidx < nk) {
    if constexpr (BDIST == 3) {
      step(b0, b3);
      if (kidx >= nk) break;
      step(b1, b0);
      if (kidx >= nk) break;
      step(b2, b1);
      if (kidx >= nk) break;
      step(b3, b2);
    } else {
      step(b0, b1);
      if (kidx >= nk) break;
      step(b1, b0);
    }
  }
  FC_STAMP(23, s);
  if (dbg & 4) return;
  const float inv = BF ? 1.f : w_scale(amax[0], true);
#pragma unroll
  for (int i = 0; i < FMT; ++i)
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
      const int n = 64 * wc + jt * 32 + l31;
      const float bv = bias ? bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t m = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m < M) {
          float v = acc[i][jt][r] * inv + bv;
          if (relu) v = fmaxf(v, 0.f);
          if constexpr (BF) reinterpret_cast<uint16_t *>(Y)[m * ldy + n] = (uint16_t)(pack_bf16_rne(v, 0.f) & 0xffffu);   // ldy in bf16 elements
          else Y[m * ldy + n] = v;
        }
      }
    }
  FC_STAMP(24, s);
}

// Bounding box of the clamped corner coordinates (the ones produce_slab reads) of a tile's 64 points x 7 displacements at
// every staged level: one wave per tile.  nvox = 0: more than FC_STAGE_VOX voxels, or the tile straddles two samples.
__global__ __launch_bounds__(64) void fc0_boxes_kernel(const FcArgs *__restrict__ Ap, const float *__restrict__ points, int64_t M,
                                                       int N, float disp, int ac, FcBox *__restrict__ boxes) {
  static_assert(FTM == 64, "one lane per row of the tile");
  const FcArgs &A = *Ap;
  const int lane = threadIdx.x;
  const int64_t pn = min((int64_t)blockIdx.x * FTM + lane, M - 1);
  const int b = (int)(pn / N);
  for (int si = 0; si < A.n_stage; ++si) {
    const FcLevel L = A.L[A.stage_level[si]];
    int lo[3] = {1 << 30, 1 << 30, 1 << 30}, hi[3] = {-1, -1, -1};
    for (int j = 0; j < 7; ++j) {
      const Corner c = sample_corner(points + pn * 3, j, disp, L.D, L.H, L.W, ac);
      const Weights w = corner_weights(c);
      const int i0[3] = {w.z0, w.y0, w.x0}, n[3] = {L.D, L.H, L.W};
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        lo[a] = min(lo[a], min(max(i0[a], 0), n[a] - 1));
        hi[a] = max(hi[a], min(max(i0[a] + 1, 0), n[a] - 1));
      }
    }
    int bl = b, bh = b;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        lo[a] = min(lo[a], __shfl_xor(lo[a], o));
        hi[a] = max(hi[a], __shfl_xor(hi[a], o));
      }
      bl = min(bl, __shfl_xor(bl, o));
      bh = max(bh, __shfl_xor(bh, o));
    }
    if (lane == 0) {
      FcBox x;
      x.b = b; x.z0 = lo[0]; x.y0 = lo[1]; x.x0 = lo[2];
      x.by = hi[1] - lo[1] + 1; x.bx = hi[2] - lo[2] + 1;
      const int nv = (hi[0] - lo[0] + 1) * x.by * x.bx;
      x.nvox = (bl == bh && nv <= FC_STAGE_VOX) ? nv : 0;
      x.pad = 0;
      boxes[(int64_t)blockIdx.x * FC_NSTAGE + si] = x;
    }
  }
}

// SVR_FC0_STAGE=0: no LDS staging of the coarse levels (A/B switch, read once; prepare and run must agree, so it is
// process-wide)
bool fc0_staging() {
  static const bool on = !(getenv("SVR_FC0_STAGE") && getenv("SVR_FC0_STAGE")[0] == '0');
  return on;
}

// slab table of a descriptor; false if a level's channel count has no slab shape.  stage: coarse levels through LDS
bool build_slabs(const svr_gather_desc *d, uint32_t keep_mask, FcArgs &A, const int32_t *keep_cols = nullptr, bool stage = false) {
  int ns = 0, k = 0;
  A.n_stage = 0;
  for (int i = 0; i < FC_NSTAGE; ++i) A.stage_level[i] = 0;
  int geo_level = -1;   // level whose geometry slab has been flagged
  auto add = [&](int level, int j0, int nj, int lp, int c0, int cols, int st = -1) {
    if (ns >= FC_MAX_SLABS) return false;
    FcSlab &S = A.S[ns++];
    S.level = level;
    S.j0 = j0;
    S.nj = nj;
    S.lp = lp;
    S.k0 = k;
    S.nk = cols / FK;
    S.c0 = c0;
    S.keep = (int)((keep_mask >> level) & 1);
    S.stage = st;
    S.geo = (lp > 0 && level != geo_level) ? 1 : 0;   // first displacement slab of the level
    if (S.geo) geo_level = level;
    k += cols / FK;
    return true;
  };
  // the C == 1 level first: the only slab that is not overlapped with MFMAs should be the cheapest
  for (int pass = 0; pass < 2; ++pass)
    for (int l = 0; l < d->n_levels; ++l) {
      const svr_level &lv = d->level[l];
      A.L[l] = FcLevel{lv.vol, lv.C, lv.D, lv.H, lv.W, lv.col, keep_cols ? keep_cols[l] : lv.col};
      const int C = lv.C;
      if ((C == 1) != (pass == 0)) continue;
      bool ok = true;
      if (C == 1) {
        ok = add(l, 0, 7, 0, 0, 16);
      } else if (C == 16) {
        for (int j = 0; j < 6 && ok; j += 2) ok = add(l, j, 2, 8, 0, 32);
        ok = ok && add(l, 6, 1, 4, 0, 16);
      } else if (C == 32) {
        for (int j = 0; j < 6 && ok; j += 2) ok = add(l, j, 2, 16, 0, 64);
        ok = ok && add(l, 6, 1, 8, 0, 32);
      } else if (C >= 64 && C % 64 == 0) {
        if (stage && FTM == 64 && lv.D <= 16 && lv.H <= 16 && lv.W <= 16 && A.n_stage < FC_NSTAGE) {
          // staged level: one 64-channel half at a time -- the STAGE slab, then its seven displacement slabs
          const int st = A.n_stage++;
          A.stage_level[st] = l;
          for (int c0 = 0; c0 < C && ok; c0 += 64) {
            ok = add(l, 0, 0, -1, c0, 0, st);
            for (int j = 0; j < 7 && ok; ++j) ok = add(l, j, 1, 16, c0, 64, st);
          }
        } else {
          for (int j = 0; j < 7 && ok; ++j)
            for (int c0 = 0; c0 < C && ok; c0 += 64) ok = add(l, j, 1, 16, c0, 64);
        }
      } else {
        return false;
      }
      if (!ok) return false;
    }
  A.n_slabs = ns;
  A.KF = k * FK;
  return ns > 0;
}

int check_desc(const svr_gather_desc *d, const char *who) {
  SVR_CHECK(d && d->n_levels >= 1 && d->n_levels <= SVR_MAX_LEVELS && d->B > 0 && d->N >= 0, SVR_E_BADARG, "%s: bad descriptor", who);
  SVR_CHECK(d->order == nullptr, SVR_E_UNSUPPORTED, "%s: a processing order is not supported (sort the points instead)", who);
  int n_c1 = 0;
  for (int l = 0; l < d->n_levels; ++l) {
    const svr_level &lv = d->level[l];
    SVR_CHECK(lv.vol && lv.D > 0 && lv.H > 0 && lv.W > 0, SVR_E_BADARG, "%s: level %d: bad volume", who, l);
    SVR_CHECK((int64_t)d->B * lv.D * lv.H * lv.W * lv.C < (1LL << 30), SVR_E_UNSUPPORTED, "%s: level %d has >= 2^30 elements", who, l);
    SVR_CHECK(((uintptr_t)lv.vol & 15) == 0 || lv.C == 1, SVR_E_ALIGN, "%s: level %d: volume must be 16-byte aligned", who, l);
    SVR_CHECK(lv.C == 1 || lv.col % 4 == 0, SVR_E_ALIGN, "%s: level %d: column %d", who, l, lv.col);
    n_c1 += lv.C == 1;
  }
  SVR_CHECK(n_c1 <= 1, SVR_E_UNSUPPORTED, "%s: more than one single-channel level", who);
  return SVR_OK;
}

}  // namespace

extern "C" int32_t svr_gather_fc0_supported(const svr_gather_desc *d) {
  if (!d || d->n_levels < 1 || d->n_levels > SVR_MAX_LEVELS || d->order) return 0;
  FcArgs A;
  if (!build_slabs(d, 0, A)) return 0;
  for (int l = 0; l < d->n_levels; ++l)
    if ((int64_t)d->B * d->level[l].D * d->level[l].H * d->level[l].W * d->level[l].C >= (1LL << 30)) return 0;
  return 1;
}

extern "C" int64_t svr_gather_fc0_workspace(const svr_gather_desc *d, int32_t n_out) {
  FcArgs A;
  if (!d || !build_slabs(d, 0, A)) return 0;
  const int64_t tiles = cdiv((int64_t)d->B * d->N, FTM);
  return 2 * (int64_t)n_out * A.KF * (int64_t)sizeof(uint16_t) + 1024 + (int64_t)sizeof(FcArgs) + 256 +
         tiles * FC_NSTAGE * (int64_t)sizeof(FcBox);
}

namespace {

struct FcWorkspace {
  uint32_t *amax;
  uint16_t *p0;
  FcArgs *Ad;
  FcBox *boxes;   // [tiles][FC_NSTAGE]
};
FcWorkspace carve(void *workspace, int32_t n_out, int KF) {
  FcWorkspace w;
  w.amax = (uint32_t *)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  w.p0 = (uint16_t *)(w.amax + 64);
  uint16_t *p1 = w.p0 + (int64_t)n_out * KF;
  w.Ad = (FcArgs *)(((uintptr_t)(p1 + (int64_t)n_out * KF) + 255) & ~(uintptr_t)255);
  w.boxes = (FcBox *)(((uintptr_t)(w.Ad + 1) + 255) & ~(uintptr_t)255);
  return w;
}

// validates the kept-matrix arguments; kend = end of the last kept level's columns
int check_keep(const svr_gather_desc *d, const FcArgs &A, const float *feat, int64_t ldf, uint32_t keep_levels, int64_t *kend) {
  const int64_t M = (int64_t)d->B * d->N;
  SVR_CHECK(keep_levels == 0 || (feat && ldf % 4 == 0 && ((uintptr_t)feat & 15) == 0 && M * ldf < (1LL << 31)),
            SVR_E_BADARG, "gather_fc0: keep_levels needs a 16-byte aligned feature matrix with < 2^31 elements");
  *kend = 0;
  for (int l = 0; l < d->n_levels; ++l) {
    if (!((keep_levels >> l) & 1)) continue;
    const int kc = A.L[l].kcol, w = 7 * d->level[l].C;
    SVR_CHECK(kc >= 0 && kc + w <= ldf && (d->level[l].C == 1 || kc % 4 == 0), SVR_E_BADSHAPE,
              "gather_fc0: kept level %d: columns [%d,%d) in a row of %ld", l, kc, kc + w, (long)ldf);
    for (int m = 0; m < l; ++m)
      if ((keep_levels >> m) & 1)
        SVR_CHECK(kc + w <= A.L[m].kcol || A.L[m].kcol + 7 * d->level[m].C <= kc, SVR_E_BADSHAPE,
                  "gather_fc0: kept levels %d and %d overlap", m, l);
    *kend = std::max<int64_t>(*kend, kc + w);
  }
  return SVR_OK;
}

}  // namespace

extern "C" int svr_gather_fc0_prepare(const svr_gather_desc *d, const float *W, int64_t ldw, int32_t n_out, float *feat,
                                      int64_t ldf, const int32_t *keep_cols, uint32_t keep_levels, void *workspace,
                                      void *stream) {
  int rc = check_desc(d, "gather_fc0_prepare");
  if (rc != SVR_OK) return rc;
  if ((int64_t)d->B * d->N == 0) return SVR_OK;
  SVR_CHECK(W && workspace, SVR_E_BADARG, "gather_fc0_prepare: null pointer");
  SVR_CHECK(n_out == FTN, SVR_E_UNSUPPORTED, "gather_fc0_prepare: %d output columns (the kernel is built for %d)", n_out, FTN);
  if (ldf <= 0) ldf = d->row_stride;
  FcArgs A;
  SVR_CHECK(build_slabs(d, keep_levels, A, keep_cols, fc0_staging()), SVR_E_UNSUPPORTED,
            "gather_fc0_prepare: a level's channel count has no slab shape (1, 16, 32, 64 k)");
  int64_t kend;
  if ((rc = check_keep(d, A, feat, ldf, keep_levels, &kend)) != SVR_OK) return rc;
  hipStream_t s = (hipStream_t)stream;
  const FcWorkspace ws = carve(workspace, n_out, A.KF);
  // W's columns: every column of the layout the slabs cover (the row's padding columns never enter the product)
  int64_t kw = 0;
  for (int l = 0; l < d->n_levels; ++l) kw = std::max<int64_t>(kw, d->level[l].col + 7 * d->level[l].C);
  (void)hipMemsetAsync(ws.amax, 0, sizeof(uint32_t), s);
  hipLaunchKernelGGL(w_amax_kernel, dim3((unsigned)std::min<int64_t>(cdiv(n_out * kw, 1024), 1024)), dim3(256), 0, s, W, ldw,
                     (int64_t)n_out, kw, ws.amax);
  hipLaunchKernelGGL(split_w_fused_kernel, dim3((unsigned)cdiv(A.KF, 64), (unsigned)n_out), dim3(64), 0, s, A, W, ldw, ws.amax,
                     ws.p0, (int)n_out);
  hipLaunchKernelGGL(args_store_kernel, dim3(1), dim3(256), 0, s, A, ws.Ad);
  return launch_status("gather_fc0_prepare");
}

extern "C" int svr_gather_fc0_run(const svr_gather_desc *d, const float *points, const float *bias, float *Y, int64_t ldy,
                                  int32_t n_out, float *feat, int64_t ldf, const int32_t *keep_cols, uint32_t keep_levels,
                                  int32_t epilogue, void *workspace, void *stream) {
  int rc = check_desc(d, "gather_fc0_run");
  if (rc != SVR_OK) return rc;
  const int64_t M = (int64_t)d->B * d->N;
  if (M == 0) return SVR_OK;
  SVR_CHECK(points && Y && workspace, SVR_E_BADARG, "gather_fc0_run: null pointer");
  SVR_CHECK(n_out == FTN, SVR_E_UNSUPPORTED, "gather_fc0_run: %d output columns (the kernel is built for %d)", n_out, FTN);
  SVR_CHECK(epilogue == SVR_EPI_NONE || ((epilogue == SVR_EPI_BIAS || epilogue == SVR_EPI_BIAS_RELU) && bias), SVR_E_BADARG,
            "gather_fc0_run: epilogue %d", epilogue);
  if (ldf <= 0) ldf = d->row_stride;
  FcArgs A;  // (the slab table is rebuilt on the host only for its sizes and the argument checks; the kernel reads the
             // copy svr_gather_fc0_prepare stored in the workspace)
  SVR_CHECK(build_slabs(d, keep_levels, A, keep_cols, fc0_staging()), SVR_E_UNSUPPORTED,
            "gather_fc0_run: a level's channel count has no slab shape (1, 16, 32, 64 k)");
  int64_t kend;
  if ((rc = check_keep(d, A, feat, ldf, keep_levels, &kend)) != SVR_OK) return rc;
  int64_t kw = 0;
  for (int l = 0; l < d->n_levels; ++l) kw = std::max<int64_t>(kw, d->level[l].col + 7 * d->level[l].C);
  // kept matrix in the full row layout: the padding starts behind the last LEVEL (kept or not), as svr_gather_trilinear_fwd
  const int64_t pad_start = keep_cols ? kend : kw;
  const FcWorkspace ws = carve(workspace, n_out, A.KF);
  static const hipError_t lds_attr =  // once per process (an immutable kernel attribute, the library's only global state)
      hipFuncSetAttribute((const void *)gather_fc0_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, FC_LDS_BYTES);
  SVR_CHECK(lds_attr == hipSuccess, (int)lds_attr, "gather_fc0_run: cannot reserve %d bytes of LDS: %s", FC_LDS_BYTES,
            hipGetErrorString(lds_attr));
  const float *eb = epilogue == SVR_EPI_NONE ? nullptr : bias;
  const int relu = epilogue == SVR_EPI_BIAS_RELU ? 1 : 0;
#ifdef SVR_FC0_MEASURE
  const int dbg = getenv("SVR_FC0_DBG") ? atoi(getenv("SVR_FC0_DBG")) : 0;
#else
  const int dbg = 0;
#endif
  if (A.n_stage > 0)   // the tiles' bounding boxes at the staged levels (the points change from call to call)
    hipLaunchKernelGGL(fc0_boxes_kernel, dim3((unsigned)cdiv(M, FTM)), dim3(64), 0, (hipStream_t)stream, ws.Ad, points, M, d->N,
                       d->displacement, d->align_corners, ws.boxes);
  // SVR_FC0_XCD=1: tiles contiguous per XCD (Morton neighbours share an L2) instead of dealt round-robin over the eight
  static const int xcd = (getenv("SVR_FC0_XCD") && getenv("SVR_FC0_XCD")[0] == '1') ? 1 : 0;
  const unsigned grid = xcd ? xcd_grid(cdiv(M, FTM)) : (unsigned)cdiv(M, FTM);
  hipLaunchKernelGGL(gather_fc0_kernel<false>, dim3(grid), dim3(NTHR), FC_LDS_BYTES, (hipStream_t)stream, ws.Ad, points,
                     ws.p0, ws.amax, eb, Y, ldy, feat, (int)ldf, keep_levels ? (int)pad_start : -1, M, d->N, d->displacement,
                     d->align_corners, relu, dbg, ws.boxes, xcd);
  return launch_status("gather_fc0_run");
}

extern "C" int svr_gather_fc0_fwd(const svr_gather_desc *d, const float *points, const float *W, int64_t ldw, const float *bias,
                                  float *Y, int64_t ldy, int32_t n_out, float *feat, int64_t ldf, const int32_t *keep_cols,
                                  uint32_t keep_levels, int32_t epilogue, void *workspace, void *stream) {
  int rc = svr_gather_fc0_prepare(d, W, ldw, n_out, feat, ldf, keep_cols, keep_levels, workspace, stream);
  if (rc != SVR_OK) return rc;
  return svr_gather_fc0_run(d, points, bias, Y, ldy, n_out, feat, ldf, keep_cols, keep_levels, epilogue, workspace, stream);
}

#ifdef SVR_FC0_MEASURE
// measurement builds only (not declared in include/svr_hip.h): buffer of FC_ST_TILES * 8 * FC_ST_N u64 for the in-kernel timeline
extern "C" int svr_gather_fc0_stamps(void *buf) {
  unsigned long long *p = (unsigned long long *)buf;
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(fc_stamps), &p, sizeof(p));
}
extern "C" int svr_gather_fc0_dbg_level(int C) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(fc_dbg_C), &C, sizeof(C)); }
extern "C" int svr_gather_fc0_dbg_points(const void *pts) {
  const float *p = (const float *)pts;
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(fc_dbg_pt0), &p, sizeof(p));
}
#endif

// ------------------------------------------------------------------------------------------------------------------
// bf16-storage variant (see the file header): levels' `vol` pointers are bf16 volumes, W is given in f32 and rounded to
// bf16 here, h0 (B*N, 256) is written in bf16.
// ------------------------------------------------------------------------------------------------------------------
extern "C" int svr_gather_fc0_bf16_prepare(const svr_gather_desc *d, const float *W, int64_t ldw, int32_t n_out, void *workspace,
                                           void *stream) {
  int rc = check_desc(d, "gather_fc0_bf16_prepare");
  if (rc != SVR_OK) return rc;
  if ((int64_t)d->B * d->N == 0) return SVR_OK;
  SVR_CHECK(W && workspace, SVR_E_BADARG, "gather_fc0_bf16_prepare: null pointer");
  SVR_CHECK(n_out == FTN, SVR_E_UNSUPPORTED, "gather_fc0_bf16_prepare: %d output columns (the kernel is built for %d)", n_out, FTN);
  FcArgs A;
  SVR_CHECK(build_slabs(d, 0, A), SVR_E_UNSUPPORTED, "gather_fc0_bf16_prepare: a level's channel count has no slab shape (1, 16, 32, 64 k)");
  for (int l = 0; l < d->n_levels; ++l)
    SVR_CHECK(((uintptr_t)d->level[l].vol & 7) == 0 || d->level[l].C == 1, SVR_E_ALIGN, "gather_fc0_bf16_prepare: level %d: 8-byte alignment", l);
  hipStream_t s = (hipStream_t)stream;
  const FcWorkspace ws = carve(workspace, n_out, A.KF);
  hipLaunchKernelGGL(split_w_fused_bf16_kernel, dim3((unsigned)cdiv(A.KF, 64), (unsigned)n_out), dim3(64), 0, s, A, W, ldw, ws.p0, (int)n_out);
  hipLaunchKernelGGL(args_store_kernel, dim3(1), dim3(256), 0, s, A, ws.Ad);
  return launch_status("gather_fc0_bf16_prepare");
}

extern "C" int svr_gather_fc0_bf16_run(const svr_gather_desc *d, const float *points, const float *bias, uint16_t *Y, int64_t ldy,
                                       int32_t n_out, int32_t epilogue, void *workspace, void *stream) {
  int rc = check_desc(d, "gather_fc0_bf16_run");
  if (rc != SVR_OK) return rc;
  const int64_t M = (int64_t)d->B * d->N;
  if (M == 0) return SVR_OK;
  SVR_CHECK(points && Y && workspace, SVR_E_BADARG, "gather_fc0_bf16_run: null pointer");
  SVR_CHECK(n_out == FTN, SVR_E_UNSUPPORTED, "gather_fc0_bf16_run: %d output columns (the kernel is built for %d)", n_out, FTN);
  SVR_CHECK(epilogue == SVR_EPI_NONE || ((epilogue == SVR_EPI_BIAS || epilogue == SVR_EPI_BIAS_RELU) && bias), SVR_E_BADARG,
            "gather_fc0_bf16_run: epilogue %d", epilogue);
  FcArgs A;
  SVR_CHECK(build_slabs(d, 0, A), SVR_E_UNSUPPORTED, "gather_fc0_bf16_run: a level's channel count has no slab shape (1, 16, 32, 64 k)");
  const FcWorkspace ws = carve(workspace, n_out, A.KF);
  static const hipError_t lds_attr =
      hipFuncSetAttribute((const void *)gather_fc0_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, FC_LDS_BYTES);
  SVR_CHECK(lds_attr == hipSuccess, (int)lds_attr, "gather_fc0_bf16_run: cannot reserve %d bytes of LDS: %s", FC_LDS_BYTES,
            hipGetErrorString(lds_attr));
  const float *eb = epilogue == SVR_EPI_NONE ? nullptr : bias;
  const int relu = epilogue == SVR_EPI_BIAS_RELU ? 1 : 0;
  hipLaunchKernelGGL(gather_fc0_kernel<true>, dim3((unsigned)cdiv(M, FTM)), dim3(NTHR), FC_LDS_BYTES, (hipStream_t)stream, ws.Ad, points,
                     ws.p0, ws.amax, eb, reinterpret_cast<float *>(Y), ldy, (float *)nullptr, 0, -1, M, d->N, d->displacement,
                     d->align_corners, relu, 0, (const FcBox *)nullptr, 0);
  return launch_status("gather_fc0_bf16_run");
}
