/*
 * svr_hip.h -- C ABI of libsvr_hip.so: the MI355X (gfx950) kernels behind the IF-Net
 * occupancy-query hot path of nihalsid/single-view-3d-reconstruction.
 *
 * The reference has no FFI layer of its own: the path sits behind torch nn.Modules
 * (model/ifnet.py:10-199, model/projection.py:21-218) and every arithmetic step is a stock
 * torch op.  Each entry point below therefore names the torch call site(s) it replaces.
 *
 * Conventions (all entry points):
 *   - plain device pointers + sizes, no torch types; the CALLER owns every buffer
 *     (inputs, outputs, workspaces) and keeps it alive until the stream work completes;
 *   - asynchronous: work is only enqueued on `stream` (a hipStream_t passed as void*),
 *     no device synchronisation, no allocation, no default-stream work;
 *   - return 0 on success, a negative SVR_E_* code for a bad argument, or a positive
 *     hipError_t for a launch failure; svr_last_error() gives the text (thread local);
 *   - volumes are channels-last: (B, D, H, W, C) float32, C contiguous;
 *   - "feature rows": (B*N, FS) float32, FS = row stride >= svr_feature_width(); column
 *     layout is level-major, see svr_feature_layout().
 */
#ifndef SVR_HIP_H
#define SVR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SVR_OK 0
#define SVR_E_BADARG (-1)
#define SVR_E_BADSHAPE (-2)
#define SVR_E_ALIGN (-3)
#define SVR_E_UNSUPPORTED (-4)
#define SVR_E_IO (-5)
#define SVR_E_NOTFOUND (-6)

#define SVR_MAX_LEVELS 6

int svr_version(void); /* 100 = round 1, 200 = this header (svr_level / svr_gather_desc grew: check svr_sizeof_*) */
const char *svr_last_error(void);

/* ---------------------------------------------------------------------------------------
 * Trilinear multi-level feature gather  (replaces: coordinate prep model/ifnet.py:156-161,
 * the six F.grid_sample calls :162,168,175,181,187,193, torch.cat :197 and the reshape
 * model/ifnet.py:43-45; 32-variant :93-118).
 * ------------------------------------------------------------------------------------- */
/* Backward plan of one level for the atomic-free "pull" scatter (svr_gather_pull_plan): the 7*B*N (point,
 * displacement) items sorted by the row-major key of their base cell, one 16-byte record per item and the first
 * item of every occupied cell.  All three arrays are device memory owned by the caller.                    */
typedef struct svr_pull_plan {
  const uint32_t *keys;  /* (7*B*N) sorted cell keys; items that touch no voxel carry the largest key      */
  const void *recs;      /* (7*B*N) x {float fx, fy, fz; int32 gfeat_offset} in sorted order               */
  const int32_t *heads;  /* (B*(D+1)*(H+1)*(W+1) + 1) CSR offsets: heads[c] = number of items with a key < c     */
  int64_t n_items;       /* 7*B*N                                                                          */
} svr_pull_plan;

typedef struct svr_level {
  const float *vol; /* (B, D, H, W, C) channels-last                                  */
  float *gvol;      /* backward only: gradient volume, same shape, accumulated into   */
  int32_t C, D, H, W;
  int32_t col;      /* first column of this level inside a feature row                */
  const int32_t *order; /* backward only, optional: (B*N) visiting order for THIS level from
                           svr_points_voxel_order (overrides svr_gather_desc.order)   */
  const int32_t *item_order; /* backward only, optional: (7*B*N) item ids from svr_gather_item_order -- the atomic
                           scatter then walks (point, displacement) items sorted jointly by base cell (7x longer
                           runs on the coarse levels); overrides `order`             */
  const svr_pull_plan *plan; /* backward only, optional (host pointer, read at call time): scatter this level
                           atomic-free in pull form.  gvol is then OVERWRITTEN (it need not be zeroed) and
                           `order` is ignored.  C in {16, 32, 64}.                     */
} svr_level;

typedef struct svr_gather_desc {
  int32_t n_levels;
  int32_t B, N;           /* batch, query points per sample                          */
  int32_t row_stride;     /* FS: floats per feature row (>= used width, multiple of 4) */
  int32_t align_corners;  /* 0: 128-architecture, 1: 32-architecture                   */
  float displacement;     /* 0.0722 / 0.035 (model/ifnet.py:144,82)                    */
  const int32_t *order;   /* optional (B*N) processing order from svr_points_morton_order, or NULL */
  svr_level level[SVR_MAX_LEVELS];
  int32_t flags;          /* SVR_GATHER_* bits, 0 = production defaults                */
} svr_gather_desc;

/* flags: test / measurement switches; results are the same with or without them.
 *   WIDE_OFFSETS   forward: take the 64-bit-offset gather body that is otherwise only selected when a
 *                  volume or the feature matrix has >= 2^31 elements (so tests can reach it at small sizes);
 *   DETERMINISTIC  backward: scatter without float atomics in a fixed summation order (one wave per
 *                  (sample, level), serial over the points like ATen's CPU grid_sampler_3d_backward):
 *                  bit-reproducible run to run, orders of magnitude slower -- for tests only.          */
#define SVR_GATHER_WIDE_OFFSETS 1
#define SVR_GATHER_DETERMINISTIC 2

/* Layout pin for bindings in other languages: sizeof(svr_level) / sizeof(svr_gather_desc) as this library was
 * compiled; a binding asserts its own struct sizes against these before the first call.                   */
int64_t svr_sizeof_level(void);
int64_t svr_sizeof_gather_desc(void);

/* order[i] = index (b*N+n) of the i-th point in (sample, Morton code at 64^3) order.  Not a
 * reference op: it only changes the order in which the gather / scatter kernels visit points
 * (L2 locality, run-combining of atomics); outputs keep the caller's point order.
 * workspace: svr_points_morton_order_workspace() bytes.                                      */
/* order[i] = index of the i-th point in (sample, row-major index, x fastest, of the BASE VOXEL of its undisplaced
 * trilinear sample in a D x H x W volume) order: points that scatter into the same 8 corners of that
 * level become consecutive, which is what the backward run-combining needs (the base-voxel lattice is
 * shifted by half a voxel, differently at every level, so one global order cannot serve all levels).
 * Same workspace size as svr_points_morton_order.                                               */
int svr_points_voxel_order(const float *points, int32_t *order, int32_t B, int32_t N, int32_t D, int32_t H,
                           int32_t W, int32_t align_corners, void *workspace, void *stream);
int64_t svr_points_morton_order_workspace(int32_t B, int32_t N);
int svr_points_morton_order(const float *points, int32_t *order, float *sorted_points /* (B,N,3) or NULL */,
                            int32_t B, int32_t N, void *workspace, void *stream);

/* Plan for the pull-form backward scatter of ONE level (volume D x H x W, C channels at feature column `col`).
 * The scatter of a sparse level is bound by the float-atomic rate (~1.3 TB/s); in pull form every voxel sums the
 * items of its <= 8 neighbouring base cells in a fixed order and is written once with plain stores: no atomics, no
 * zero-initialised gradient volume, bit-reproducible.  Not a reference op (ATen's CPU backward scatters serially).
 * keys / recs / heads: see svr_pull_plan (heads needs B*(D+1)*(H+1)*(W+1)+1 int32).  B*N*row_stride and the cell
 * count must be below 2^31.  workspace: svr_gather_pull_plan_workspace(B, N) +
 * svr_gather_pull_plan_workspace_cells(B, D, H, W) bytes.                                                    */
int64_t svr_gather_pull_plan_workspace(int32_t B, int32_t N);
int64_t svr_gather_pull_plan_workspace_cells(int32_t B, int32_t D, int32_t H, int32_t W);
int svr_gather_pull_plan(const float *points, int32_t B, int32_t N, int32_t D, int32_t H, int32_t W, int32_t C,
                         int32_t col, int32_t row_stride, int32_t align_corners, float displacement,
                         uint32_t *keys, void *recs, int32_t *heads, int32_t *items /* optional (7*B*N): the sorted item
                         ids = svr_level.item_order of the same level */, int32_t *stats /* optional, 2 device ints:
                         [0] = longest serial walk of the pull kernel (items in one (row, x strip) range) */,
                         void *workspace, void *stream);

/* items[i] = id (b*N+n)*7 + j of the i-th (point, displacement) item in (sample, row-major base cell of ITS displaced
 * sample in a D x H x W volume) order; items that touch no voxel come last.  For svr_level.item_order.
 * workspace: svr_gather_pull_plan_workspace(B, N) bytes.                                                     */
int svr_gather_item_order(const float *points, int32_t B, int32_t N, int32_t D, int32_t H, int32_t W,
                          int32_t align_corners, float displacement, int32_t with_j /* 1: order by (cell, j) */,
                          int32_t *items, void *workspace, void *stream);
/* Backward-only projection of a wide level (gather.hip, gather_bwd_proj_kernel): the scatter commutes with fc_0's
 * product, so the level's 7*C feature columns need neither dX nor dW of the point MLP over all points:
 *   dP[b][v][j][0:256] += w(item, v) * dh[(b*N+n)][0:256]   for every item (n, j) of svr_gather_item_order(with_j = 1)
 * (dP: (B, D*H*W, 7, 256) float32, zero-initialised by the caller; dh: the gradient wrt fc_0's pre-activation, row
 * stride lddh >= 256).  The caller finishes with two GEMMs over VOXELS: dvol = dP W0_l, dW0_l = dP^T vol.       */
int svr_gather_project_bwd(const float *points, const float *dh, int64_t lddh, int32_t B, int32_t N, int32_t D, int32_t H,
                           int32_t W, int32_t align_corners, float displacement, const int32_t *items, float *dP,
                           void *stream);

/* Two-pass form of svr_gather_project_bwd (no float atomics, no memset of dP, bit-reproducible): pass 1 stores the 8 corner
 * sums of every run -- a maximal stretch of equal (sample, cell, displacement) keys inside a 256-item chunk of the sorted
 * items -- to partials[slot][8][256]; pass 2 gives every (voxel, displacement) row of dP the sum of the <= 8 cells that touch
 * it, with plain stores (dP is OVERWRITTEN).  svr_gather_project_plan sorts the items (as svr_gather_item_order with_j = 1)
 * and builds keys (7*B*N sorted u32), sidx (7*B*N + 1: run starts in front of an item) and first_slot
 * (B*(D+1)*(H+1)*(W+1)*8 + 1); partials needs svr_gather_project_slots(...) * 8 * 256 floats (not initialised).
 * workspace: svr_gather_project_plan_workspace(B, N) bytes.                                                              */
int64_t svr_gather_project_plan_workspace(int32_t B, int32_t N);
int64_t svr_gather_project_slots(int32_t B, int32_t N, int32_t D, int32_t H, int32_t W);
int svr_gather_project_plan(const float *points, int32_t B, int32_t N, int32_t D, int32_t H, int32_t W, int32_t align_corners,
                            float displacement, int32_t *items, uint32_t *keys, int32_t *sidx, int32_t *first_slot,
                            void *workspace, void *stream);
int svr_gather_project_bwd2(const float *points, const float *dh, int64_t lddh, int32_t B, int32_t N, int32_t D, int32_t H,
                            int32_t W, int32_t align_corners, float displacement, const int32_t *items, const int32_t *sidx,
                            const int32_t *first_slot, float *partials, float *dP, void *stream);

/* features[b*N+n][level.col + j*C + c] = trilinear sample j of channel c (zeros padding);
 * columns past the last level (up to row_stride) are written as zeros.                      */
int svr_gather_trilinear_fwd(const svr_gather_desc *d, const float *points /*(B,N,3)*/,
                             float *features, void *stream);
/* Fused gather -> first point-MLP layer (gather_fc0.hip): the feature rows of svr_gather_trilinear_fwd are produced
 * 128 points x one K-slab at a time in LDS and multiplied straight into fc_0's 128 x 256 output tile (3-product f16
 * split, as svr_linear_fwd_f16x3), so the (B*N, row_stride) feature matrix is never written or read:
 *   Y[b*N+n][0:n_out] = epi( features[b*N+n][:] . W[0:n_out][:]^T )         (model/ifnet.py:156-197 + :43-45,55)
 * W (n_out, >= last used column), row stride ldw, in the feature row's column layout (svr_level.col); n_out = 256.
 * keep_levels: bit l set = the gathered values of level l are ALSO stored to `features` -- the rows a backward still
 * needs; 0 = inference, `features` may be NULL.  The kept matrix has row stride ldf floats (<= 0: d->row_stride) and
 * level l starts at column keep_cols[l] (host array of n_levels entries, read at call time; NULL: svr_level.col, i.e.
 * the layout of svr_gather_trilinear_fwd) -- a compact matrix of the kept levels only (800 columns instead of a
 * 2592-wide row for the 128-architecture's training step).  With any bit set the columns behind the last kept level
 * (keep_cols given) / the last level (NULL) up to ldf are zero-filled.  d->order must be NULL (sort the points instead);
 * channel counts 1 (at most one level), 16, 32 or multiples of 64; every volume < 2^30 elements;
 * svr_gather_fc0_supported(d) tells.  workspace: svr_gather_fc0_workspace(d, n_out) bytes.                       */
int32_t svr_gather_fc0_supported(const svr_gather_desc *d);
int64_t svr_gather_fc0_workspace(const svr_gather_desc *d, int32_t n_out);
int svr_gather_fc0_fwd(const svr_gather_desc *d, const float *points, const float *W, int64_t ldw, const float *bias, float *Y,
                       int64_t ldy, int32_t n_out, float *features, int64_t ldf, const int32_t *keep_cols, uint32_t keep_levels,
                       int32_t epilogue, void *workspace, void *stream);
/* The same in two calls, for callers that query ONE pyramid with ONE weight matrix many times (dense-grid inference,
 * model/ifnet.py:215-229: one chunk of the lattice per call): svr_gather_fc0_prepare splits W into the kernel's f16
 * planes and stores the slab table in `workspace` (three small launches), svr_gather_fc0_run is the gather -> fc_0
 * kernel alone and may be repeated with other points / B*N <= the prepared descriptor's, same volumes, same layout.
 * svr_gather_fc0_fwd = prepare + run.                                                                             */
int svr_gather_fc0_prepare(const svr_gather_desc *d, const float *W, int64_t ldw, int32_t n_out, float *features, int64_t ldf,
                           const int32_t *keep_cols, uint32_t keep_levels, void *workspace, void *stream);
int svr_gather_fc0_run(const svr_gather_desc *d, const float *points, const float *bias, float *Y, int64_t ldy, int32_t n_out,
                       float *features, int64_t ldf, const int32_t *keep_cols, uint32_t keep_levels, int32_t epilogue,
                       void *workspace, void *stream);
/* bf16-STORAGE variant of the fused kernel (the throughput mode of svr_gather_trilinear_fwd_bf16 + svr_linear_fwd_bf16 in
 * one launch; never the default): the levels' `vol` pointers are bf16 volumes (8-byte aligned), W (n_out, ldw) is given in
 * f32 and rounded to bf16 (round to nearest even), the feature values are rounded to bf16 once -- the bits
 * svr_gather_trilinear_fwd_bf16 writes -- and multiplied on the bf16 matrix cores with f32 accumulation; Y (B*N, ldy) bf16.
 * workspace: svr_gather_fc0_workspace(d, n_out) bytes; prepare once per pyramid and weight, run per point set.            */
int svr_gather_fc0_bf16_prepare(const svr_gather_desc *d, const float *W, int64_t ldw, int32_t n_out, void *workspace,
                                void *stream);
int svr_gather_fc0_bf16_run(const svr_gather_desc *d, const float *points, const float *bias, uint16_t *Y, int64_t ldy,
                            int32_t n_out, int32_t epilogue, void *workspace, void *stream);
/* gvol[level] += scatter of gfeatures (autograd of grid_sample wrt the volume);
 * levels with gvol == NULL are skipped.  gpoints (B,N,3) may be NULL; when given it is
 * OVERWRITTEN with the gradient wrt the query points.                                   */
int svr_gather_trilinear_bwd(const svr_gather_desc *d, const float *points, const float *gfeatures,
                             float *gpoints, void *stream);
/* Integer base corner (z0,y0,x0) of every sample: out (B,7,N,3) int32 -- the bit-exact gate. */
int svr_gather_corner_indices(const svr_gather_desc *d, int32_t level, const float *points,
                              int32_t *out, void *stream);

/* ---------------------------------------------------------------------------------------
 * bf16-STORAGE throughput mode of the query path (north_star "bf16 occupancy logits", BASELINE configs[1];
 * reference hooks: the dtype-generic grid_sample / Conv1d calls model/ifnet.py:161-193,55-59 under
 * util/arguments.py:30 --precision 16).  Volumes, feature rows and MLP activations are bf16 in memory; corner
 * weights, sums, MFMA accumulators, bias and ReLU are f32; every stored value is rounded once (nearest even).
 * The sample geometry is the f32 code of the default path: corner indices stay bit-exact.  A separate mode --
 * never the default, never held to the fp32 1e-4 gate (bf16_path.hip).
 * ------------------------------------------------------------------------------------- */
int svr_cast_f32_to_bf16(const float *in, uint16_t *out, int64_t n, void *stream);
/* As svr_gather_trilinear_fwd, with svr_level.vol pointing to bf16 volumes (B,D,H,W,C) and bf16 feature rows
 * (row_stride in ELEMENTS, multiple of 8).  One of the levels must have C == 1 (its kernel writes the padding
 * columns) unless the levels fill the row.                                                               */
int svr_gather_trilinear_fwd_bf16(const svr_gather_desc *d, const float *points, uint16_t *features, void *stream);
/* Y[M,N] (bf16) = epi( X[M,K] (bf16, ldx) W[N,K]^T (bf16, ldw) ), f32 accumulation on v_mfma_f32_32x32x16_bf16;
 * epilogue NONE / BIAS / BIAS_RELU with an f32 bias; 32 | K, rows 16-byte aligned.                         */
int svr_linear_fwd_bf16(const uint16_t *X, int64_t ldx, const uint16_t *W, int64_t ldw, const float *bias, uint16_t *Y,
                        int64_t ldy, int64_t M, int64_t N, int64_t K, int epilogue, void *stream);
/* logits[r(m)] (f32) = H[m,:] (bf16) . w (f32) + b,  r(m) = row_map[m] or m (see svr_fc_out_fwd)           */
int svr_fc_out_fwd_bf16(const uint16_t *H, int64_t ldh, const float *w, const float *b, float *logits,
                        const int32_t *row_map, int64_t M, int64_t K, void *stream);

/* ---------------------------------------------------------------------------------------
 * Dense f32 GEMMs on MFMA (replaces nn.Conv1d(.,.,1) fc_0/fc_1/fc_2 model/ifnet.py:19-21,55-57
 * and their autograd).  Row-major everywhere.
 * ------------------------------------------------------------------------------------- */
#define SVR_EPI_NONE 0
#define SVR_EPI_BIAS 1       /* + bias[n]                       */
#define SVR_EPI_BIAS_RELU 2  /* relu(. + bias[n])               */
#define SVR_EPI_MASK 3       /* . * (mask[m][n] > 0)  (ReLU backward with the saved output) */

/* Y[M,N] = epi( X[M,K](ldx) * W[N,K](ldw)^T )                                           */
int svr_linear_fwd(const float *X, int64_t ldx, const float *W, int64_t ldw, const float *bias,
                   float *Y, int64_t ldy, int64_t M, int64_t N, int64_t K, int epilogue,
                   const float *mask, int64_t ldmask, void *stream);
/* dX[M,K] = epi( dY[M,N](lddy) * W[N,K](ldw) ), epilogue NONE or MASK (mask is [M,K])    */
int svr_linear_bwd_data(const float *dY, int64_t lddy, const float *W, int64_t ldw, float *dX,
                        int64_t lddx, int64_t M, int64_t N, int64_t K, int epilogue,
                        const float *mask, int64_t ldmask, void *stream);
/* dW[N,K](lddw) = dY[M,N]^T * X[M,K];  db[N] = column sums of dY (db may be NULL).
 * workspace: svr_linear_bwd_weight_workspace() bytes.  Deterministic (slab + ordered sum). */
int64_t svr_linear_bwd_weight_workspace(int64_t M, int64_t N, int64_t K);
int svr_linear_bwd_weight(const float *dY, int64_t lddy, const float *X, int64_t ldx, float *dW,
                          int64_t lddw, float *db, int64_t M, int64_t N, int64_t K,
                          void *workspace, void *stream);

/* Forward product at f32 accuracy on the bf16 matrix cores: x = hi + mid + lo (three bf16 terms = 24
 * mantissa bits), six MFMA products, f32 accumulation; agrees with svr_linear_fwd to ~2e-7 relative.
 * epilogue NONE / BIAS / BIAS_RELU.  workspace: svr_linear_fwd_bf16x6_workspace(N, K) bytes.          */
int64_t svr_linear_fwd_bf16x6_workspace(int64_t N, int64_t K);
int svr_linear_fwd_bf16x6(const float *X, int64_t ldx, const float *W, int64_t ldw, const float *bias,
                          float *Y, int64_t ldy, int64_t M, int64_t N, int64_t K, int epilogue,
                          void *workspace, void *stream);
/* Same result with the 3-product f16 split (x = hi + lo in f16: 22 mantissa bits; W normalised by a power of two,
 * lo terms kept normal by exact 2^11 scaling, see gemm_f16x3.hip): as accurate as an f32 GEMM (~3e-7 of f64) at
 * half the matrix-core work of bf16x6.  Domain: |X| < 65504 (f16 range).
 * workspace: svr_linear_fwd_f16x3_workspace(N, K) bytes.
 * PREPARE / RUN (the four split-precision entry points svr_linear_fwd_f16x3, svr_linear_bwd_data_bf16x3,
 * svr_conv3d_k3_fwd_f16x3, svr_conv3d_k3_bwd_data_bf16x3): with the data operand (X / dY / in / dout) NULL the call only
 * prepares the workspace from W (scale + split planes: the launches that depend on the parameters alone); with W NULL it
 * runs on a workspace an earlier call prepared.  A training step prepares every layer once, on a side stream, while the
 * first kernels of the step run (model/ifnet.py), instead of 2-4 launch-bound kernels in front of every layer.         */
int64_t svr_linear_fwd_f16x3_workspace(int64_t N, int64_t K);
int svr_linear_fwd_f16x3(const float *X, int64_t ldx, const float *W, int64_t ldw, const float *bias,
                          float *Y, int64_t ldy, int64_t M, int64_t N, int64_t K, int epilogue,
                          void *workspace, void *stream);

/* The same two backward products on the bf16 matrix cores with a 3-term split (x = hi + mid, products
 * hi*hi + hi*mid + mid*hi, f32 accumulation): ~1.5e-5 relative error per product, ~5x fewer matrix-core
 * cycles than the exact-f32 MFMA.  Backward only -- the forward pass (logits, ReLU masks) stays exact f32.
 * bwd_data workspace: svr_linear_bwd_data_bf16x3_workspace(N, K) bytes (split + transposed weight planes). */
int64_t svr_linear_bwd_data_bf16x3_workspace(int64_t N, int64_t K);
int svr_linear_bwd_data_bf16x3(const float *dY, int64_t lddy, const float *W, int64_t ldw, float *dX,
                               int64_t lddx, int64_t M, int64_t N, int64_t K, int epilogue,
                               const float *mask, int64_t ldmask, void *workspace, void *stream);
int64_t svr_linear_bwd_weight_bf16x3_workspace(int64_t M, int64_t N, int64_t K);
int svr_linear_bwd_weight_bf16x3(const float *dY, int64_t lddy, const float *X, int64_t ldx, float *dW,
                                 int64_t lddw, float *db, int64_t M, int64_t N, int64_t K,
                                 void *workspace, void *stream);

/* The same two backward products at f32 LEVEL on the f16 matrix cores ("f16x3s": the scaled 3-product f16 split).  The
 * reference computes its backward in fp32 (util/arguments.py:30, precision 32); bf16x3 above carries 16 mantissa bits per
 * operand, this carries 22 -- at the same three matrix instructions per block.  f16 has 5 exponent bits, so a gradient
 * operand is first multiplied by an exact power of two that brings its |max| into [2^13, 2^14):
 *   svr_amax_f32: amax[0] = bit pattern of max |X[m][n]| (one pass; N, ld multiples of 4, rows 16-byte aligned), or the
 *   amax_dx output of svr_linear_bwd_data_f16x3 (|max| of the dX it stored: the next layer's dY needs no extra pass).
 * Every amax OUTPUT (svr_amax_f32's, amax_dx, amax_din, ...) is an atomic maximum: the word must hold 0 (or a lower
 * bound) on entry -- the library launches no memset for it.
 * amax_dy == NULL: no scaling (values must then lie in f16's range).  The weight operand is normalised the same way when
 * its planes are prepared (PREPARE / RUN as above: dY NULL = prepare from W, W NULL = run); the activation operand X of
 * the weight gradient is taken as it is (|x| < 65504; below |x| ~ 0.1 its correction term has an absolute error floor of
 * 3e-8 |dY|).  Errors against f64: ~3e-7 (the exact-f32 kernels' own level), bf16x3: ~1e-5.                              */
int svr_amax_f32(const float *X, int64_t ld, int64_t M, int64_t N, uint32_t *amax, void *stream);
int64_t svr_linear_bwd_data_f16x3_workspace(int64_t N, int64_t K);
int svr_linear_bwd_data_f16x3(const float *dY, int64_t lddy, const float *W, int64_t ldw, float *dX,
                              int64_t lddx, int64_t M, int64_t N, int64_t K, int epilogue,
                              const float *mask, int64_t ldmask, const uint32_t *amax_dy, uint32_t *amax_dx,
                              void *workspace, void *stream);
int64_t svr_linear_bwd_weight_f16x3_workspace(int64_t M, int64_t N, int64_t K);
int svr_linear_bwd_weight_f16x3(const float *dY, int64_t lddy, const float *X, int64_t ldx, float *dW,
                                int64_t lddw, float *db, int64_t M, int64_t N, int64_t K,
                                const uint32_t *amax_dy, void *workspace, void *stream);

/* fc_out (Conv1d(hidden,1,1), model/ifnet.py:35,58-59): logits[r(m)] = H[m,:].w + b, where
 * r(m) = row_map[m] if row_map != NULL (rows were processed in Morton order: scatter the logits back
 * to the caller's point order) else m.                                                            */
int svr_fc_out_fwd(const float *H, int64_t ldh, const float *w, const float *b, float *logits,
                   const int32_t *row_map, int64_t M, int64_t K, void *stream);
/* With g[m] = dlogits[r(m)]:  dH[m,k] = g[m]*w[k]*(H[m,k]>0);  dw[k] = sum_m g[m]*H[m,k];
 * db = sum g.  workspace: svr_fc_out_bwd_workspace() bytes.  amax_dh (may be NULL; must hold 0 on entry):
 * receives the bit pattern of max |dH| -- the scale of the next layer's "f16x3s" products, see
 * svr_linear_bwd_data_f16x3.                                                                 */
int64_t svr_fc_out_bwd_workspace(int64_t M, int64_t K);
int svr_fc_out_bwd(const float *H, int64_t ldh, const float *w, const float *dlogits,
                   const int32_t *row_map, float *dH, int64_t lddh, float *dw, float *db, int64_t M,
                   int64_t K, uint32_t *amax_dh, void *workspace, void *stream);

/* BCE-with-logits, reduction 'none' -> sum over points -> mean over batch
 * (trainer/trainer_ifnet.py:46).  loss: 1 float; dlogits (B,N) = gscale*(sigmoid(z)-y)/B
 * (dlogits may be NULL).  workspace: B doubles.                                           */
int svr_bce_logits_sum_mean(const float *logits, const float *targets, float *loss, float *dlogits,
                            int64_t B, int64_t N, float gscale, void *workspace, void *stream);

/* ---------------------------------------------------------------------------------------
 * 3x3x3 convolution, padding 1, channels-last (replaces nn.Conv3d(.,.,3,padding=1) + ReLU
 * model/ifnet.py:126-135,164-191 and autograd).  Weights in the packed layout made by
 * svr_conv3d_pack_weight: fwd  Wp[tap][ci][co],  bwd-data  Wp[tap'][co][ci] (flipped taps).
 * ------------------------------------------------------------------------------------- */
int svr_conv3d_pack_weight(const float *W /*(Co,Ci,3,3,3)*/, float *Wp_fwd, float *Wp_bwd,
                           int32_t Ci, int32_t Co, void *stream);
int svr_conv3d_unpack_wgrad(const float *dWp /*[tap][ci][co]*/, float *dW /*(Co,Ci,3,3,3)*/,
                            int32_t Ci, int32_t Co, void *stream);
/* out(B,D,H,W,Co) = epi( conv(in(B,D,H,W,Ci), Wp) ): epilogue BIAS / BIAS_RELU / NONE / MASK.
 * Used for forward (Wp_fwd) and for backward-data (Wp_bwd, Ci<->Co swapped by the caller). */
int svr_conv3d_k3(const float *in, const float *Wp, const float *bias, float *out, int32_t B,
                  int32_t D, int32_t H, int32_t W, int32_t Ci, int32_t Co, int epilogue,
                  const float *mask, void *stream);
/* conv_in (Ci == 1, model/ifnet.py:126,165) forward with the statistics of the following BatchNorm3d fused in:
 * out(B,D,H,W,Co) = epi(conv(in(B,D,H,W,1), Wp[27][1][Co]) + bias), stats[0:Co] = mean, stats[Co:2Co] = biased variance
 * of `out` (float64: what svr_bn_stats(out) returns) without re-reading the output.  Co in {16, 32}.
 * workspace: svr_conv3d_c1_fwd_stats_workspace(B, D, H, W, Co) bytes.                                       */
int64_t svr_conv3d_c1_fwd_stats_workspace(int32_t B, int32_t D, int32_t H, int32_t W, int32_t Co);
int svr_conv3d_c1_fwd_stats(const float *in, const float *Wp, const float *bias, float *out, double *stats,
                            int32_t B, int32_t D, int32_t H, int32_t W, int32_t Co, int epilogue,
                            void *workspace, void *stream);
/* Forward at f32 accuracy on the bf16 matrix cores (bf16x6, see svr_linear_fwd_bf16x6): takes the UNPACKED
 * weights W(Co,Ci,3,3,3); epilogue NONE / BIAS / BIAS_RELU; Ci % 16 == 0.
 * workspace: svr_conv3d_fwd_bf16x6_workspace(Ci, Co) bytes.                                              */
int64_t svr_conv3d_fwd_bf16x6_workspace(int32_t Ci, int32_t Co);
int svr_conv3d_k3_fwd_bf16x6(const float *in, const float *W, const float *bias, float *out, int32_t B,
                             int32_t D, int32_t H, int32_t Wd, int32_t Ci, int32_t Co, int epilogue,
                             void *workspace, void *stream);
/* Same contract with the 3-product f16 split (see svr_linear_fwd_f16x3): f32-level accuracy for |in| < 65504 at half
 * the matrix-core work of bf16x6.  workspace: svr_conv3d_fwd_f16x3_workspace(Ci, Co) bytes.               */
int64_t svr_conv3d_fwd_f16x3_workspace(int32_t Ci, int32_t Co);
int svr_conv3d_k3_fwd_f16x3(const float *in, const float *W, const float *bias, float *out, int32_t B,
                             int32_t D, int32_t H, int32_t Wd, int32_t Ci, int32_t Co, int epilogue,
                             void *workspace, void *stream);
/* ... + the BatchNorm statistics of `out` from the kernel's epilogue: part[blocks][2][Co] float64 partial sums (sum, sum of
 * squares of the stored values), blocks = svr_conv3d_fwd_f16x3_stats_blocks(same shape); finish with svr_bn_finalize_parts.
 * (model/ifnet.py:170-172 etc.: the last convolution of a stage -> ReLU -> BatchNorm3d.)                              */
int32_t svr_conv3d_fwd_f16x3_stats_blocks(int32_t B, int32_t D, int32_t H, int32_t Wd, int32_t Ci, int32_t Co);
int svr_conv3d_k3_fwd_f16x3_stats(const float *in, const float *W, const float *bias, float *out, double *part, int32_t B,
                                  int32_t D, int32_t H, int32_t Wd, int32_t Ci, int32_t Co, int epilogue,
                                  void *workspace, void *stream);
/* Backward-data on the bf16 matrix cores with the 3-term split (see svr_linear_bwd_data_bf16x3):
 * din(B,D,H,W,Ci) = epi( conv^T(dout(B,D,H,W,Co), W(Co,Ci,3,3,3)) ), epilogue NONE or MASK (mask like din).
 * Takes the UNPACKED weights; workspace: svr_conv3d_bwd_data_bf16x3_workspace(Ci, Co) bytes.  Ci even.   */
int64_t svr_conv3d_bwd_data_bf16x3_workspace(int32_t Ci, int32_t Co);
int svr_conv3d_k3_bwd_data_bf16x3(const float *dout, const float *W, float *din, int32_t B, int32_t D,
                                  int32_t H, int32_t Wd, int32_t Ci, int32_t Co, int epilogue,
                                  const float *mask, void *workspace, void *stream);
/* dWp[tap][ci][co] = sum_m in[m+tap][ci]*dout[m][co]; db[co] = sum_m dout[m][co] (may be NULL). */
int64_t svr_conv3d_k3_bwd_weight_workspace(int32_t B, int32_t D, int32_t H, int32_t W, int32_t Ci,
                                           int32_t Co);
int svr_conv3d_k3_bwd_weight(const float *in, const float *dout, float *dWp, float *db, int32_t B,
                             int32_t D, int32_t H, int32_t W, int32_t Ci, int32_t Co,
                             void *workspace, void *stream);
/* Same result on the bf16 matrix cores with the 3-term split (voxels are the MFMA reduction; ~1e-5 relative);
 * Ci, Co % 4 == 0.  workspace: svr_conv3d_k3_bwd_weight_bf16x3_workspace(...) bytes.                      */
int64_t svr_conv3d_k3_bwd_weight_bf16x3_workspace(int32_t B, int32_t D, int32_t H, int32_t W, int32_t Ci,
                                                  int32_t Co);
int svr_conv3d_k3_bwd_weight_bf16x3(const float *in, const float *dout, float *dWp, float *db, int32_t B,
                                    int32_t D, int32_t H, int32_t W, int32_t Ci, int32_t Co,
                                    void *workspace, void *stream);
/* The same, written straight in the parameter's layout dW(Co,Ci,3,3,3) (autograd's layout of nn.Conv3d.weight.grad):
 * the slab reduction, the layout change and the bias-gradient reduction are one launch.                          */
int svr_conv3d_k3_bwd_weight_bf16x3_param(const float *in, const float *dout, float *dW, float *db, int32_t B,
                                          int32_t D, int32_t H, int32_t W, int32_t Ci, int32_t Co,
                                          void *workspace, void *stream);
/* The encoder's two backward products at f32 LEVEL on the scaled f16 split ("f16x3s", see svr_linear_bwd_data_f16x3):
 * amax_dout = svr_amax_f32 of dout (or the amax output of the kernel that made it; NULL: no scaling), amax_din (may be
 * NULL) receives |max| of the stored din.  bwd_data: PREPARE / RUN like the bf16x3 entry (dout NULL = prepare from W,
 * W NULL = run), workspace svr_conv3d_bwd_data_f16x3_workspace(Ci, Co) bytes, Co % 16 == 0, Ci even.  bwd_weight:
 * workspace svr_conv3d_k3_bwd_weight_bf16x3_workspace(...) bytes, Ci, Co % 4 == 0; param_layout != 0: dW(Co,Ci,3,3,3),
 * else the packed [tap][ci][co].                                                                                    */
int64_t svr_conv3d_bwd_data_f16x3_workspace(int32_t Ci, int32_t Co);
int svr_conv3d_k3_bwd_data_f16x3(const float *dout, const float *W, float *din, int32_t B, int32_t D,
                                 int32_t H, int32_t Wd, int32_t Ci, int32_t Co, int epilogue,
                                 const float *mask, const uint32_t *amax_dout, uint32_t *amax_din,
                                 void *workspace, void *stream);
int svr_conv3d_k3_bwd_weight_f16x3(const float *in, const float *dout, float *dW, float *db, int32_t B,
                                   int32_t D, int32_t H, int32_t W, int32_t Ci, int32_t Co, int32_t param_layout,
                                   const uint32_t *amax_dout, void *workspace, void *stream);

/* ---------------------------------------------------------------------------------------
 * BatchNorm3d (training or eval) + MaxPool3d(2), channels-last
 * (replaces nn.BatchNorm3d model/ifnet.py:138-142,165-192 and nn.MaxPool3d(2) :136,169-188).
 * ------------------------------------------------------------------------------------- */
/* stats[0:C] = mean, stats[C:2C] = biased variance over (B,D,H,W) in float64.
 * workspace: svr_bn_stats_workspace() bytes.                                              */
int64_t svr_bn_stats_workspace(int64_t rows, int32_t C);
int svr_bn_stats(const float *x, double *stats, int64_t rows /*B*D*H*W*/, int32_t C,
                 void *workspace, void *stream);
/* svr_bn_stats followed by svr_bn_finalize(training = 1) in two launches instead of three (stats may be NULL).      */
int svr_bn_stats_finalize(const float *x, double *stats, const float *gamma, const float *beta, float *running_mean,
                          float *running_var, float *scale_shift, float *mean_f32, int64_t rows, int32_t C,
                          float eps, float momentum, void *workspace, void *stream);
/* The same from per-workgroup partial sums part[blocks][2][C] (float64 sum / sum of squares of the BatchNorm's input, left by
 * svr_conv3d_k3_fwd_f16x3_stats): no pass over the tensor at all.                                                   */
int svr_bn_finalize_parts(const double *part, int32_t blocks, double *stats, const float *gamma, const float *beta,
                          float *running_mean, float *running_var, float *scale_shift, float *mean_f32, int64_t rows,
                          int32_t C, float eps, float momentum, void *stream);
/* scale_shift[0:C] = gamma*invstd, [C:2C] = beta - mean*gamma*invstd, [2C:3C] = invstd (f32);
 * training != 0 also updates running_mean/var (momentum, unbiased var) like torch.          */
int svr_bn_finalize(const double *stats, const float *gamma, const float *beta, float *running_mean,
                    float *running_var, float *scale_shift, float *mean_f32, int64_t rows,
                    int32_t C, float eps, float momentum, int training, void *stream);
/* y = x*scale + shift (full resolution);  pooled (B,D/2,H/2,W/2,C) = 2x2x2 max of y (floor),
 * argmax (uint8 0..7, first maximum in z,y,x scan order) -- pooled/argmax may be NULL.      */
int svr_bn_apply_pool(const float *x, const float *scale_shift, float *y, float *pooled,
                      uint8_t *argmax, int32_t B, int32_t D, int32_t H, int32_t W, int32_t C,
                      void *stream);
/* Backward of [ReLU ->] BN -> {sample, pool}:
 *   dy_total = dy (may be NULL = 0) + unpool(dpooled via argmax) (dpooled may be NULL)
 *   sums[0:C] = sum dy_total, sums[C:2C] = sum dy_total*xhat   (float64, step 1)
 *   dx = gamma*invstd*(dy_total - sum1/n - xhat*sum2/n) * (x > 0 if relu_mask & 1)  (step 2)
 *   dgamma = sum2, dbeta = sum1.
 * relu_mask bit 1 (value 2): the forward used FROZEN statistics (eval mode, running mean / var from
 * svr_bn_finalize(training = 0)): autograd of F.batch_norm(training=False) is dx = gamma*invstd*dy_total;
 * dgamma / dbeta are the same sums.                                                          */
int svr_bn_bwd_reduce(const float *x, const float *dy, const float *dpooled, const uint8_t *argmax,
                      const float *mean_f32, const float *scale_shift, double *sums, int32_t B,
                      int32_t D, int32_t H, int32_t W, int32_t C, void *workspace, void *stream);
int svr_bn_bwd_apply(const float *x, const float *dy, const float *dpooled, const uint8_t *argmax,
                     const float *mean_f32, const float *scale_shift, const float *gamma,
                     const double *sums, float *dx, float *dgamma, float *dbeta, int32_t B,
                     int32_t D, int32_t H, int32_t W, int32_t C, int relu_mask, uint32_t *amax_dx,
                     void *stream);   /* amax_dx (may be NULL; 0 on entry): bit pattern of max |dx|, as svr_fc_out_bwd's */

/* ---------------------------------------------------------------------------------------
 * First encoder stage of the 128-architecture as one recomputed unit (stage1.hip):
 *   a = relu(conv_in(x) + bias), y = BatchNorm3d(a), pooled = MaxPool3d(2)(y)
 * (replaces model/ifnet.py:126,138,136 as called at :165-166,169, and their autograd).  `a` -- one input channel, 27 taps,
 * 16 outputs -- is recomputed from x in every pass instead of being stored: the forward reads x and writes y / pooled /
 * argmax only, the backward reads x, dy, dpooled and writes the parameter gradients only.
 *   svr_stage1_supported: Co == 16 and 32-bit offsets inside one sample.
 *   svr_stage1_fwd:  training != 0: batch statistics (stats[0:16] mean, [16:32] biased variance, float64; running
 *                    statistics updated like torch), else the running statistics; outputs as svr_bn_finalize +
 *                    svr_bn_apply_pool (scale_shift 3 x 16, mean_f32 16, y, pooled / argmax may be NULL).
 *   svr_stage1_bwd:  dy_total = dy (may be NULL) + unpool(dpooled via argmax) (may be NULL); sums (float64, 2 x 16) =
 *                    sum dy_total, sum dy_total*xhat; dgamma, dbeta; dW(16,1,3,3,3) (the PARAMETER's layout) and db[16]
 *                    (may be NULL) of conv_in; dout (may be NULL)
 *                    = d(loss)/d(conv_in output) for a caller that needs d(loss)/d(x).  relu_mask as svr_bn_bwd_apply.
 *   f16x3 != 0: the recomputed convolution runs as the 3-product f16 split of svr_conv3d_k3_fwd_f16x3 (f32-level
 *                    accuracy for |x| < 65504, 3 matrix instructions per 16 voxels instead of 7 exact-f32 ones); 0: exact
 *                    f32.  The backward must be called with the flag the forward was called with (it recomputes the same
 *                    activation: ReLU mask and xhat bit for bit).
 *   workspace: svr_stage1_workspace(B, D, H, W) bytes.  Wp = svr_conv3d_pack_weight's Wp_fwd ([27][1][16]).          */
int32_t svr_stage1_supported(int32_t B, int32_t D, int32_t H, int32_t W, int32_t Co);
int64_t svr_stage1_workspace(int32_t B, int32_t D, int32_t H, int32_t W);
int svr_stage1_fwd(const float *x, const float *Wp, const float *bias, const float *gamma, const float *beta,
                   float *running_mean, float *running_var, float *y, float *pooled, uint8_t *argmax,
                   float *scale_shift, float *mean_f32, double *stats, int32_t B, int32_t D, int32_t H, int32_t W,
                   int32_t Co, float eps, float momentum, int training, int f16x3, void *workspace, void *stream);
int svr_stage1_bwd(const float *x, const float *Wp, const float *bias, const float *dy, const float *dpooled,
                   const uint8_t *argmax, const float *mean_f32, const float *scale_shift, double *sums,
                   float *dgamma, float *dbeta, float *dW, float *db, float *dout, int32_t B, int32_t D,
                   int32_t H, int32_t W, int32_t Co, int relu_mask, int f16x3, void *workspace, void *stream);

/* ---------------------------------------------------------------------------------------
 * Depth -> point cloud -> voxel grid  (model/projection.py).
 * ------------------------------------------------------------------------------------- */
/* pc[b][v*Wi+u] = (c2f * [X,Y,Z,1]) then optionally (p - dims/2)/dims  (projection.py:150-163,
 * 199-206, 124-132).  consts: f, cx, cy, c2f[0][0], c2f[0][3], c2f[1][1], c2f[1][3], c2f[2][2],
 * c2f[2][3], dims0, dims1, dims2 (12 floats, host memory).  gdepth/gpc for the backward.   */
int svr_unproject_fwd(const float *depth, float *pc, int32_t B, int32_t Hi, int32_t Wi,
                      const float *consts, int normalize, void *stream);
int svr_unproject_bwd(const float *depth, const float *gpc, float *gdepth, int32_t B, int32_t Hi,
                      int32_t Wi, const float *consts, int normalize, void *stream);
/* acc (B,D0,D1,D2) += trilinear splat of valid points (projection.py:39-78); acc must be zeroed
 * by the caller.  base (B,N,3) int32 and valid (B,N) uint8 are optional outputs (bit-exact gate). */
int svr_voxelize_splat_fwd(const float *pts, float *acc, int32_t *base, uint8_t *valid, int32_t B,
                           int32_t N, int32_t D0, int32_t D1, int32_t D2, void *stream);
/* gpts (B,N,3) = gradient wrt the normalised points given gacc (gradient wrt acc).          */
int svr_voxelize_splat_bwd(const float *pts, const float *gacc, float *gpts, int32_t B, int32_t N,
                           int32_t D0, int32_t D1, int32_t D2, void *stream);
/* out = clamp(scale*in, 0, 1): scale 8 = the reference's x8 alias quirk (projection.py:75-80),
 * scale 1 = the final clamp of the blur (:116).  bwd: gin = scale*gout where 0 <= scale*in <= 1. */
int svr_scale_clamp01_fwd(const float *in, float *out, int64_t n, float scale, void *stream);
int svr_scale_clamp01_bwd(const float *in, const float *gout, float *gin, int64_t n, float scale,
                          void *stream);
/* One axis pass of the separable blur (projection.py:96-114): out = correlation along `axis`
 * (0 = first spatial) with taps[K] (device memory, K odd <= 15), zero padding K/2.           */
int svr_blur_axis_fwd(const float *in, const float *taps, float *out, int32_t B, int32_t D0,
                      int32_t D1, int32_t D2, int32_t axis, int32_t K, void *stream);
/* gin = adjoint pass of gout (gin may be NULL); gtaps[K] (float64, zeroed by the caller) +=
 * sum_i gout[i]*in[i + t - K/2]  (gtaps may be NULL).                                        */
int svr_blur_axis_bwd(const float *in, const float *taps, const float *gout, float *gin,
                      double *gtaps, int32_t B, int32_t D0, int32_t D1, int32_t D2, int32_t axis,
                      int32_t K, void *stream);

/* ---------------------------------------------------------------------------------------
 * 2-D convolution blocks of the UNet depth regressor (replaces nn.Conv2d(k4,s2,p1) / nn.Conv2d(k3,s1,p1),
 * nn.LeakyReLU(0.2) / nn.ReLU, nn.Upsample(scale_factor=2, mode="bilinear") and torch.cat of model/unet.py:38-60,
 * 66-118 and their autograd).  Channels-last (B,H,W,C) float32.  A block's input is  act( cat(src0, src1) ), optionally
 * upsampled x2 (align_corners False); svr_conv2d_im2col writes its patch matrix
 *   col[(b,oy,ox)][(ky*k+kx)*C + c],  C = C0 + C1,  (B*Ho*Wo) x (k*k*C),  Ho = (Hv + 2 - k)/stride + 1
 * and the convolution itself is svr_linear_fwd* on it (weights repacked to [Cout][(ky*k+kx)*C + c]); backward:
 * svr_linear_bwd_weight* on (dY, col), svr_linear_bwd_data* -> dcol, then svr_conv2d_col2im sums dcol back in gather
 * form (workspace dvirt: B*Hv*Wv*C floats), applies the upsample adjoint and the activation mask and writes the
 * gradients of the two sources (either may be NULL).  act: 0 none, 1 LeakyReLU(0.2), 2 ReLU.
 * ------------------------------------------------------------------------------------- */
typedef struct svr_conv2d_desc {
  const float *src0, *src1; /* (B,H,W,C0), (B,H,W,C1) or NULL */
  int32_t B, H, W, C0, C1;
  int32_t k, stride;        /* (4,2) or (3,1); padding 1 */
  int32_t act, upsample;
} svr_conv2d_desc;
int svr_conv2d_im2col(const svr_conv2d_desc *d, float *col, void *stream);
int svr_conv2d_col2im(const svr_conv2d_desc *d, const float *dcol, float *dvirt, float *dsrc0, float *dsrc1, void *stream);

/* The same blocks as IMPLICIT GEMMs (round 4; the default of model/unet.py's hip backend, conv2d_igemm.hip): no patch matrix.
 * The A operand of the split-precision MFMA GEMM is gathered from the channels-last input -- a reduction step is 16 channels of
 * one tap; activation and concatenation are applied on the way in.  Descriptors here have upsample = 0: a decoder block first
 * writes its input  V = upsample_x2(act(cat(src0, src1)))  once with svr_conv2d_virtual (1x the activation; any descriptor)
 * and convolves {src0 = V, act = 0}.
 *   svr_conv2d_prepare   W (Cout, C, k, k) as nn.Conv2d holds it -> f16 hi / lo planes of the forward product and (want_bwd)
 *                        of the backward-data product in `planes` (svr_conv2d_planes_bytes); amax: one device word, left with
 *                        max|W| (the scale of the split).  Once per weight version.
 *   svr_conv2d_fwd       Y (B, Ho, Wo, Cout) = conv(act(cat(src0, src1))) + bias (bias may be NULL).  amax_x (may be NULL): word
 *                        holding max|input| -- the input is then a GRADIENT operand, scaled by a power of two into f16's range
 *                        (f16x3s); amax_y (may be NULL): zeroed word, left with max|Y|.  k = 1, stride 1 (no padding) is a plain
 *                        row-major GEMM Y (M, Cout) = X (M, C) W^T with the reduction split of the deep layers: B = H = 1, W = M
 *                        (svr_amd.ops.linear_bwd_data_splitk: few output tiles, long reductions).
 *   svr_conv2d_bwd_data  dIn (B, H, W, C0 + C1) = gradient of the convolution's (activated, concatenated) input from
 *                        dY (B, Ho, Wo, Cout); amax_dy: word holding max|dY| (scaled f16 split), NULL = unscaled.  Stride-2
 *                        layers run as four stride-1 problems, one per parity class of the input pixel.
 *   svr_conv2d_finish_bwd  dIn (or, for an upsampled block, the gradient of V) -> gradients of src0 / src1: upsample adjoint
 *                        (gather form), activation derivative, channel split.  `d` is the BLOCK's descriptor (upsample as is).
 *   svr_conv2d_bwd_weight  dW (Cout, C, k, k) in the parameter's own layout and db (Cout, may be NULL) from dY and the gathered
 *                        input; workspace svr_conv2d_bwd_weight_workspace bytes.
 * Layers with few output tiles split the reduction over workgroups and sum the partial outputs in a fixed order (workspace
 * svr_conv2d_workspace_bytes; deterministic, no atomics).  Limits: B*H*W and B*Ho*Wo < 2^24.                                  */
/* Blocks with 1..4 output channels (the UNet's last layer, 64 -> channels_out: a GEMM tile would be 1/64 full) run on the vector
 * ALUs in plain f32 FMAs straight from W (Cout, C, k, k): no planes, no amax words.  svr_conv2d_small_supported: Cout in 1..4 and
 * Cout*C*k*k <= 8192 (the weights live in LDS).  Same descriptors (upsample = 0) and the same results layout as above.             */
int svr_conv2d_small_supported(int32_t Cout, int32_t C, int32_t k);
int svr_conv2d_small_fwd(const svr_conv2d_desc *d, const float *W, const float *bias, float *Y, int32_t Cout, void *stream);
int svr_conv2d_small_bwd_data(const svr_conv2d_desc *d, const float *W, const float *dY, int32_t Cout, float *dIn, void *stream);
int64_t svr_conv2d_small_bwd_weight_workspace(const svr_conv2d_desc *d, int32_t Cout);
int svr_conv2d_small_bwd_weight(const svr_conv2d_desc *d, const float *dY, int32_t Cout, float *dW, float *db, void *workspace,
                                void *stream);
int64_t svr_conv2d_planes_bytes(int32_t Cout, int32_t C, int32_t k);
int svr_conv2d_prepare(const float *W, int32_t Cout, int32_t C, int32_t k, int32_t stride, int32_t want_bwd, uint32_t *amax,
                       void *planes, void *stream);
/* All layers of a network at once (n <= 16; three launches instead of four per layer): arrays of n weights / shapes / plane
 * buffers; amax_base: n words 256 bytes apart (layer i's word = amax_base + 64 i), zeroed here.  Same planes as n single calls. */
int svr_conv2d_prepare_many(int32_t n, const float *const *W, const int32_t *Cout, const int32_t *C, const int32_t *k,
                            const int32_t *stride, const int32_t *want_bwd, uint32_t *amax_base, void *const *planes, void *stream);
int64_t svr_conv2d_workspace_bytes(const svr_conv2d_desc *d, int32_t Cout);
int svr_conv2d_virtual(const svr_conv2d_desc *d, float *V, void *stream);
int svr_conv2d_fwd(const svr_conv2d_desc *d, const void *planes, const uint32_t *amax_w, const float *bias, float *Y, int32_t Cout,
                   const uint32_t *amax_x, uint32_t *amax_y, void *workspace, void *stream);
int svr_conv2d_bwd_data(const svr_conv2d_desc *d, const void *planes, const uint32_t *amax_w, const float *dY,
                        const uint32_t *amax_dy, int32_t Cout, float *dIn, void *workspace, void *stream);
int svr_conv2d_finish_bwd(const svr_conv2d_desc *d, const float *dvirt, float *dsrc0, float *dsrc1, void *stream);
int64_t svr_conv2d_bwd_weight_workspace(const svr_conv2d_desc *d, int32_t Cout);
int svr_conv2d_bwd_weight(const svr_conv2d_desc *d, const float *dY, const uint32_t *amax_dy, int32_t Cout, float *dW, float *db,
                          void *workspace, void *stream);

/* ---------------------------------------------------------------------------------------
 * Occupancy labelling against a triangle mesh (replaces check_mesh_contains,
 * data_processing/libmesh/inside_mesh.py:5-155, and the Cython TriangleHash, libmesh/triangle_hash.pyx:8-85,
 * called per training step by data_processing/mesh_occupancies.py:24-53 when subsample_points > 0).
 * Host side (plain C++, no device work): the 2-D triangle hash as a CSR table.
 *   svr_mesh_hash_entries: number of (cell, triangle) entries; writes scale[3], translate[3] of the mesh rescale to
 *                          [0.5, res-0.5]^3 (inside_mesh.py:17-22) to scale_translate[6].  Negative = SVR_E_*.
 *   svr_mesh_hash_build:   tri (n_faces*9 rescaled float64 coordinates), cell_start (res*res+1), tri_ids (entries;
 *                          triangles in index order inside a cell, like the reference's push_back order).
 * verts (n_verts,3) float64 and faces (n_faces,3) int32 are HOST arrays; all outputs are HOST arrays the caller
 * then copies to the device.
 * Device side: svr_mesh_contains -- points (n,3) float32 (points_f64 = 0) or float64 (1) on the device;
 * contains[i] = inside by both z-ray directions, holes[i] = the two directions disagree (uint8 0/1).  float64
 * arithmetic in the reference's operation order: the booleans equal the reference's bit for bit.
 * ------------------------------------------------------------------------------------- */
int64_t svr_mesh_hash_entries(const double *verts, int64_t n_verts, const int32_t *faces, int64_t n_faces, int32_t res,
                              double *scale_translate);
int svr_mesh_hash_build(const double *verts, int64_t n_verts, const int32_t *faces, int64_t n_faces, int32_t res,
                        double *tri, int32_t *cell_start, int32_t *tri_ids, int64_t entries);
int svr_mesh_contains(const void *points, int32_t points_f64, int64_t n, const double *tri, const int32_t *cell_start,
                      const int32_t *tri_ids, int32_t res, const double *scale_translate, uint8_t *contains,
                      uint8_t *holes, void *stream);

/* ---------------------------------------------------------------------------------------
 * Sample wire formats (replaces the Python loaders of dataset/implicit_dataset.py:24-56,
 * data_processing/volume_reader.py:36-45 and the np.load calls on process_sample.py:19-30's outputs).
 * Host side (plain C++ + zlib; `out` / `payload` are HOST buffers, ideally pinned):
 *   .df  = 3 x uint64 dims (X, Y, Z) + X*Y*Z float32, x fastest;
 *   .npz = zip of .npy members (stored by np.savez, deflated by np.savez_compressed; zip64 local headers).
 * Device side: x-fastest -> C-order transpose of a .df payload, casts to float32, and the random row subset of
 * implicit_dataset.py:40-43 as a gather with cast (idx: device int64; rows outside [0, n_rows) set *bad_flag).
 * ------------------------------------------------------------------------------------- */
#define SVR_DT_F32 0
#define SVR_DT_F64 1
#define SVR_DT_BOOL 2
#define SVR_DT_U8 3
#define SVR_DT_I32 4
#define SVR_DT_I64 5
int svr_df_dims(const char *path, int64_t *dims /*[3]*/);
int svr_df_read(const char *path, float *payload, int64_t n);
int svr_npz_member_info(const char *path, const char *member, int32_t *dtype, int32_t *ndim, int64_t *shape /*[8]*/,
                        int32_t *fortran_order);
int svr_npz_member_read(const char *path, const char *member, void *out, int64_t nbytes);
int svr_df_to_grid(const float *payload, float *out /*(X,Y,Z) C order*/, int32_t X, int32_t Y, int32_t Z, void *stream);
int svr_cast_to_f32(const void *in, int32_t dtype, float *out, int64_t n, void *stream);
int svr_subsample_rows(const void *rows, int32_t dtype, int64_t n_rows, int32_t cols, const int64_t *idx, int64_t n_idx,
                       float *out, int32_t *bad_flag, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SVR_HIP_H */
