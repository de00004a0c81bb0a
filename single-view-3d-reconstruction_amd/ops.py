"""Thin tensor-level wrappers over the C ABI: check device/dtype/contiguity, pass raw device
pointers and the current HIP stream.  torch is used for memory and streams only."""
import ctypes as C
import os

import torch

from . import _lib
from .arena import alloc as _alloc
from ._lib import EPI_BIAS, EPI_BIAS_RELU, EPI_MASK, EPI_NONE, GatherDesc, check  # noqa: F401


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    if t is None:
        return C.c_void_p(0)
    if not t.is_cuda:
        raise RuntimeError("svr_amd ops need GPU tensors (the HIP path has no CPU fallback)")
    if not t.is_contiguous():
        raise RuntimeError("svr_amd ops need contiguous tensors")
    return C.c_void_p(t.data_ptr())


def _f32(*ts):
    for t in ts:
        if t is not None and t.dtype != torch.float32:
            raise RuntimeError(f"expected float32, got {t.dtype}")


# ------------------------------------------------------------------------------------------
# feature-row layout
# ------------------------------------------------------------------------------------------
class FeatureLayout:
    """Column layout of a feature row: levels with C>=4 first (each 7*C wide, sample-major,
    channel contiguous, 16-float aligned), the C==1 level last; row stride padded to 32."""

    def __init__(self, channels):
        self.channels = list(channels)
        order = [l for l, c in enumerate(self.channels) if c >= 4] + [l for l, c in enumerate(self.channels) if c < 4]
        self.col = [0] * len(self.channels)
        off = 0
        for l in order:
            self.col[l] = off
            off += 7 * self.channels[l]
        self.width = off
        self.row_stride = (off + 31) // 32 * 32
        self._subsets = {}

    def subset(self, levels):
        """Compact layout of a SUBSET of the levels (same relative order, no gaps, row stride padded to 32): the
        kept-column matrix of the training step -- the levels whose backward is not projected keep their gathered
        columns in (B*N, 800) instead of a 2592-wide row (128-architecture)."""
        key = tuple(sorted(levels))
        k = self._subsets.get(key)
        if k is None:          # cached: the layout's device-side column index is uploaded once, not once per step
            k = self._subsets[key] = KeptLayout(self, levels)
        return k

    def reference_permutation(self):
        """perm[k_internal] = reference feature row k = c_global*7 + j (model/ifnet.py:43-45,197);
        -1 for padding columns."""
        perm = torch.full((self.row_stride,), -1, dtype=torch.long)
        cbase = 0
        for l, c in enumerate(self.channels):
            j = torch.arange(7).view(7, 1)
            ch = torch.arange(c).view(1, c)
            perm[self.col[l]: self.col[l] + 7 * c] = ((cbase + ch) * 7 + j).reshape(-1)
            cbase += c
        return perm


class KeptLayout:
    """Column layout of the kept-column matrix: `col[l]` for the kept levels (-1 for the others), `row_stride`, and
    `full_cols` (row_stride,) long: the column of the FULL feature row behind every kept column (a padding column of
    the full row -- zero weight, zero feature -- behind the kept matrix's own padding)."""

    def __init__(self, full, levels):
        self.full = full
        self.levels = tuple(sorted(levels, key=lambda l: full.col[l]))
        self.channels = list(full.channels)
        self.col = [-1] * len(full.channels)
        off, src = 0, []
        for l in self.levels:
            self.col[l] = off
            w = 7 * full.channels[l]
            src += list(range(full.col[l], full.col[l] + w))
            off += w
        self.width = off
        self.row_stride = (off + 31) // 32 * 32
        if self.row_stride > off and full.row_stride <= full.width:
            raise RuntimeError("KeptLayout: the full row has no padding column to stand behind the kept padding")
        src += [full.width] * (self.row_stride - off)            # a zero column of the full row
        self.full_cols = torch.tensor(src, dtype=torch.long)
        self._dev = {}

    def full_cols_on(self, device):
        t = self._dev.get(device)
        if t is None:
            t = self._dev[device] = self.full_cols.to(device)
        return t


def morton_order(points, want_sorted=False, arena=None):
    """(B*N,) int32 processing order: points of one sample sorted by a 64^3 Morton code
    (and, optionally, the points gathered into that order)."""
    _f32(points)
    B, N, _ = points.shape
    l = _lib.lib()
    order = _alloc(arena, "morton_order", (B * N,), torch.int32, points.device)
    spts = _alloc(arena, "morton_points", tuple(points.shape), torch.float32, points.device) if want_sorted else None
    ws = _alloc(arena, "morton_ws", (l.svr_points_morton_order_workspace(B, N),), torch.uint8, points.device)
    check(l.svr_points_morton_order(_p(points), _p(order), _p(spts), B, N, _p(ws), _stream()), "morton_order")
    return (order, spts) if want_sorted else order


def voxel_order(points, dims, align_corners=False):
    """(B*N,) int32: points ordered by the row-major index (x fastest) of their base voxel in a volume of `dims` (D,H,W)."""
    _f32(points)
    B, N, _ = points.shape
    l = _lib.lib()
    order = torch.empty(B * N, device=points.device, dtype=torch.int32)
    ws = torch.empty(l.svr_points_morton_order_workspace(B, N), device=points.device, dtype=torch.uint8)
    check(l.svr_points_voxel_order(_p(points), _p(order), B, N, dims[0], dims[1], dims[2], int(align_corners), _p(ws),
                                   _stream()), "voxel_order")
    return order


class PullPlan:
    """Plan of the atomic-free pull-form backward scatter of one level (svr_gather_pull_plan): keeps the device
    arrays alive and hands the C struct to the gather descriptor."""

    def __init__(self, keys, recs, heads, n_items, items=None, stats=None):
        self.keys, self.recs, self.heads, self.items, self.stats = keys, recs, heads, items, stats
        self.c = _lib.PullPlan(_p(keys), _p(recs), _p(heads), n_items)

    def record_stream(self, stream):
        if self.persistent:      # arena-owned buffers never return to the allocator
            return
        for t in (self.keys, self.recs, self.heads, self.items, self.stats):
            if t is not None:
                t.record_stream(stream)

    persistent = False


def pull_plan_supported(B, N, dims, C, row_stride):
    cells = B * (dims[0] + 1) * (dims[1] + 1) * (dims[2] + 1)
    return C in (16, 32, 64) and N > 0 and cells < 2 ** 31 - 1 and 7 * B * N < 2 ** 31 and B * N * row_stride < 2 ** 31


def pull_plan(points, dims, C, col, row_stride, displacement, align_corners=False, arena=None, tag=""):
    """Sort the 7*B*N (point, displacement) items of one level by base cell and build the records / cell heads.
    arena / tag: take the plan's arrays from a StepArena (names prefixed with `tag`) instead of the allocator."""
    _f32(points)
    B, N, _ = points.shape
    l = _lib.lib()
    T = 7 * B * N
    cells = B * (dims[0] + 1) * (dims[1] + 1) * (dims[2] + 1)
    dev = points.device
    keys = _alloc(arena, tag + "keys", (max(T, 1),), torch.int32, dev)
    recs = _alloc(arena, tag + "recs", (max(T, 1), 4), torch.int32, dev)
    heads = _alloc(arena, tag + "heads", (cells + 1,), torch.int32, dev)
    items = _alloc(arena, tag + "items", (max(T, 1),), torch.int32, dev)   # sorted item ids: also this level's item order
    stats = _alloc(arena, tag + "stats", (2,), torch.int32, dev).zero_()
    # (the sort workspace is shared by all plans of a step: they are built one after the other on one stream)
    ws = _alloc(arena, "plan_ws", (l.svr_gather_pull_plan_workspace(B, N) + l.svr_gather_pull_plan_workspace_cells(B, *dims),),
                torch.uint8, dev)
    check(l.svr_gather_pull_plan(_p(points), B, N, dims[0], dims[1], dims[2], C, col, row_stride, int(align_corners),
                                 displacement, _p(keys), _p(recs), _p(heads), _p(items), _p(stats), _p(ws), _stream()),
          "gather_pull_plan")
    plan = PullPlan(keys, recs, heads, T, items, stats)
    plan.persistent = arena is not None
    return plan


def item_order(points, dims, displacement, align_corners=False, with_j=False, arena=None, tag=""):
    """(7*B*N,) int32 item ids pn*7+j sorted by (sample, base cell of the displaced sample[, displacement j]) in a
    volume of `dims`."""
    _f32(points)
    B, N, _ = points.shape
    l = _lib.lib()
    items = _alloc(arena, tag + "items", (max(7 * B * N, 1),), torch.int32, points.device)
    ws = _alloc(arena, "plan_ws", (l.svr_gather_pull_plan_workspace(B, N),), torch.uint8, points.device)
    check(l.svr_gather_item_order(_p(points), B, N, dims[0], dims[1], dims[2], int(align_corners), displacement, int(with_j),
                                  _p(items), _p(ws), _stream()), "gather_item_order")
    return items


def project_bwd_supported(B, N, dims, lddh=256):
    cells = B * (dims[0] + 1) * (dims[1] + 1) * (dims[2] + 1)
    return N > 0 and 8 * cells < 2 ** 31 - 1 and 7 * B * N < 2 ** 31 and B * N * lddh < 2 ** 31 and max(dims) < 1022


_proj_partials = {}


class ProjPlan:
    """Plan of the two-pass (atomic-free) projected scatter of one level (svr_gather_project_plan): the items sorted by
    (sample, cell, displacement), their sorted keys, the run-start counts and the first slot of every key."""

    def __init__(self, items, keys, sidx, first_slot, slots):
        self.items, self.keys, self.sidx, self.first_slot, self.slots = items, keys, sidx, first_slot, slots

    def record_stream(self, stream):
        for t in (self.items, self.keys, self.sidx, self.first_slot):
            t.record_stream(stream)


def project_plan(points, dims, displacement, align_corners=False):
    _f32(points)
    B, N, _ = points.shape
    l = _lib.lib()
    T = 7 * B * N
    nkeys = B * (dims[0] + 1) * (dims[1] + 1) * (dims[2] + 1) * 8
    dev = points.device
    items = torch.empty(max(T, 1), device=dev, dtype=torch.int32)
    keys = torch.empty(max(T, 1), device=dev, dtype=torch.int32)          # uint32 bit patterns
    sidx = torch.empty(T + 1, device=dev, dtype=torch.int32)
    first_slot = torch.empty(nkeys + 1, device=dev, dtype=torch.int32)
    ws = torch.empty(l.svr_gather_project_plan_workspace(B, N), device=dev, dtype=torch.uint8)
    check(l.svr_gather_project_plan(_p(points), B, N, dims[0], dims[1], dims[2], int(align_corners), displacement, _p(items),
                                    _p(keys), _p(sidx), _p(first_slot), _p(ws), _stream()), "gather_project_plan")
    return ProjPlan(items, keys, sidx, first_slot, int(l.svr_gather_project_slots(B, N, dims[0], dims[1], dims[2])))


def gather_project_bwd(points, dh, dims, items, displacement, align_corners=False, out=None):
    """dP (B, D*H*W, 7, 256) = scatter of the 256-wide rows of dh (B*N, >= 256) with the trilinear weights of every
    (point, displacement) item; `items` = item_order(..., with_j=True) (float atomics into a zeroed dP) or a ProjPlan
    (two passes, no atomics, dP written once, bit-reproducible)."""
    _f32(points, dh)
    B, N, _ = points.shape
    assert dh.shape[0] == B * N and dh.shape[1] >= 256 and dh.stride(1) == 1
    if isinstance(items, ProjPlan):
        plan = items
        dP = torch.empty(B, dims[0] * dims[1] * dims[2], 7, 256, device=points.device, dtype=torch.float32)
        # the partial-sum buffer (2.7 GB at 16^3 x 8 samples: sized for the worst case, ~85 % used) is kept between calls:
        # as a per-step allocation it was split up by the caching allocator between two steps and re-allocated with
        # hipMalloc every time (22 instead of 18 ms/step).  One buffer per device and stream; calls on a stream serialise.
        key = (points.device.index, torch.cuda.current_stream(points.device).cuda_stream)
        partials = _proj_partials.get(key)
        if partials is None or partials.numel() < plan.slots * 8 * 256:
            partials = _proj_partials[key] = torch.empty(plan.slots * 8 * 256, device=points.device, dtype=torch.float32)
        check(_lib.lib().svr_gather_project_bwd2(_p(points), C.c_void_p(dh.data_ptr()), dh.stride(0), B, N, dims[0], dims[1],
                                                 dims[2], int(align_corners), displacement, _p(plan.items), _p(plan.sidx),
                                                 _p(plan.first_slot), _p(partials), _p(dP), _stream()), "gather_project_bwd2")
        return dP
    # out: a ZEROED (B, D*H*W, 7, 256) buffer of the caller (the training step zero-fills it on the side stream during the forward)
    dP = out if out is not None else torch.zeros(B, dims[0] * dims[1] * dims[2], 7, 256, device=points.device, dtype=torch.float32)
    assert tuple(dP.shape) == (B, dims[0] * dims[1] * dims[2], 7, 256) and dP.is_contiguous()
    check(_lib.lib().svr_gather_project_bwd(_p(points), C.c_void_p(dh.data_ptr()), dh.stride(0), B, N, dims[0], dims[1], dims[2],
                                            int(align_corners), displacement, _p(items), _p(dP), _stream()), "gather_project_bwd")
    return dP


def make_gather_desc(vols, gvols, layout, B, N, displacement, align_corners, order=None, level_orders=None, flags=0,
                     level_plans=None, skip_levels=()):
    """layout: a FeatureLayout, or a KeptLayout (compact kept-column matrix; the levels it does not hold must be in
    skip_levels).  skip_levels: these levels get NULL volume pointers (a backward skips them)."""
    d = GatherDesc()
    d.flags = int(flags)
    d.order = _p(order)
    d.n_levels = len(vols)
    d.B, d.N = B, N
    d.row_stride = layout.row_stride
    d.align_corners = int(align_corners)
    d.displacement = displacement
    for l, v in enumerate(vols):
        g = gvols[l] if gvols is not None else None
        ref = v if v is not None else g
        _f32(v, g)
        if ref.dim() != 5 or ref.shape[0] != B or ref.shape[4] != layout.channels[l]:
            raise RuntimeError(f"level {l}: expected (B,D,H,W,{layout.channels[l]}) channels-last, got {tuple(ref.shape)}")
        L = d.level[l]
        L.C, L.D, L.H, L.W = ref.shape[4], ref.shape[1], ref.shape[2], ref.shape[3]
        if l in skip_levels:
            L.vol, L.gvol, L.col = C.c_void_p(0), C.c_void_p(0), 0
            continue
        if layout.col[l] < 0:
            raise RuntimeError(f"level {l} has no columns in this layout (list it in skip_levels)")
        L.vol, L.gvol = _p(v), _p(g)
        L.col = layout.col[l]
        o = level_orders[l] if level_orders is not None else None
        # a (7*B*N) order is an ITEM order (svr_gather_item_order), a (B*N) one a point order (svr_points_voxel_order)
        if o is not None and B * N > 0 and o.numel() == 7 * B * N:
            L.item_order, L.order = _p(o), C.c_void_p(0)
        else:
            L.item_order, L.order = C.c_void_p(0), _p(o)
        if level_plans is not None and level_plans[l] is not None:
            L.plan = C.pointer(level_plans[l].c)
    return d


# Test switches (results do not change): GATHER_FLAGS is OR-ed into every gather descriptor --
# _lib.GATHER_WIDE_OFFSETS takes the 64-bit-offset forward body at any size, _lib.GATHER_DETERMINISTIC replaces the
# atomic backward scatter by the serial fixed-order one (bit-reproducible; equals ATen's CPU summation order).
GATHER_FLAGS = 0


def gather_fwd(vols, points, layout, displacement, align_corners, out=None, order=None, flags=0):
    B, N, _ = points.shape
    _f32(points)
    d = make_gather_desc(vols, None, layout, B, N, displacement, align_corners, order, flags=flags | GATHER_FLAGS)
    if out is None:   # the kernel writes every column (padding columns as zeros): no memset of the 4 GB buffer
        out = torch.empty(B * N, layout.row_stride, device=points.device, dtype=torch.float32)
    check(_lib.lib().svr_gather_trilinear_fwd(C.byref(d), _p(points), _p(out), _stream()), "gather_fwd")
    return out


def gather_fc0_supported(vols, points, layout, displacement, align_corners, n_out=256):
    B, N, _ = points.shape
    if n_out != 256 or not points.is_cuda:
        return False
    d = make_gather_desc(vols, None, layout, B, N, displacement, align_corners)
    return bool(_lib.lib().svr_gather_fc0_supported(C.byref(d)))


class Fc0Prepared:
    """fc_0's weights split into the fused kernel's f16 planes + the slab table of one pyramid (svr_gather_fc0_prepare):
    reusable for any number of gather_fc0_run calls on the same volumes / layout / weights."""

    def __init__(self, vols, layout, displacement, align_corners, n_out, ws, keep_levels, keep_layout, B=None, N=None):
        self.vols, self.layout, self.displacement, self.align_corners = list(vols), layout, displacement, align_corners
        # the point set the workspace was sized for (one FcBox record per 64-point tile and staged level): a larger query
        # would write past it on the device, so gather_fc0_run refuses it
        self.B, self.N = B, N
        self.n_out, self.ws, self.keep_levels, self.keep_layout = n_out, ws, tuple(keep_levels), keep_layout
        self.mask = 0
        for lv in self.keep_levels:
            self.mask |= 1 << lv
        self.kc = None
        self.stride = layout.row_stride
        if self.mask and keep_layout is not None:
            if set(keep_layout.levels) != set(self.keep_levels):
                raise RuntimeError("gather_fc0: keep_layout does not hold exactly keep_levels")
            self.stride = keep_layout.row_stride
            self.kc = (C.c_int32 * len(self.vols))(*[int(c) for c in keep_layout.col])


def gather_fc0_prepare(vols, layout, displacement, align_corners, w, B, N, keep_levels=(), keep_layout=None, arena=None):
    """Split w (n_out, >= layout.width) for the fused gather -> fc_0 kernel and store the slab table (three small
    launches); -> Fc0Prepared for gather_fc0_run.  B, N: the largest point set that will be queried."""
    _f32(w)
    d = make_gather_desc(vols, None, layout, B, N, displacement, align_corners)
    l = _lib.lib()
    n_out = w.shape[0]
    assert w.stride(1) == 1 and w.shape[1] >= layout.width
    ws_bytes = l.svr_gather_fc0_workspace(C.byref(d), n_out)
    if ws_bytes <= 0:
        raise RuntimeError("gather_fc0: unsupported level shapes (see svr_gather_fc0_supported)")
    ws = _alloc(arena, "fc0_ws", (ws_bytes,), torch.uint8, w.device)
    prep = Fc0Prepared(vols, layout, displacement, align_corners, n_out, ws, keep_levels, keep_layout, B, N)
    # (the kept matrix itself is only needed by run; prepare validates its geometry against a dummy aligned address)
    check(l.svr_gather_fc0_prepare(C.byref(d), C.c_void_p(w.data_ptr()), w.stride(0), n_out, C.c_void_p(256) if prep.mask else None,
                                   prep.stride if prep.mask else 0, prep.kc, prep.mask, _p(ws), _stream()), "gather_fc0_prepare")
    return prep


def gather_fc0_run(prep, points, bias, relu=True, rows_out=None):
    """The fused gather -> fc_0 kernel alone, on a prepared pyramid: -> (h0 (B*N, n_out), kept rows or None)."""
    B, N, _ = points.shape
    _f32(points, bias)
    if prep.B is not None and (B != prep.B or N > prep.N):
        raise RuntimeError(f"gather_fc0: points ({B}, {N}, 3) exceed the prepared capacity (B = {prep.B}, N <= {prep.N}): "
                           "prepare again for the larger point set")
    d = make_gather_desc(prep.vols, None, prep.layout, B, N, prep.displacement, prep.align_corners)
    out = torch.empty(B * N, prep.n_out, device=points.device, dtype=torch.float32)
    rows, ldf = None, 0
    if prep.mask:
        if rows_out is not None:
            if tuple(rows_out.shape) != (B * N, prep.stride) or not rows_out.is_contiguous():
                raise RuntimeError(f"gather_fc0: rows_out must be contiguous ({B * N}, {prep.stride})")
            _f32(rows_out)
            rows = rows_out
        else:
            rows = torch.empty(B * N, prep.stride, device=points.device, dtype=torch.float32)
        ldf = prep.stride
    epi = EPI_NONE if bias is None else (EPI_BIAS_RELU if relu else EPI_BIAS)
    check(_lib.lib().svr_gather_fc0_run(C.byref(d), _p(points), _p(bias), _p(out), out.stride(0), prep.n_out, _p(rows), ldf,
                                        prep.kc, prep.mask, epi, _p(prep.ws), _stream()), "gather_fc0_run")
    return out, rows


def gather_fc0_fwd(vols, points, layout, displacement, align_corners, w, bias, relu=True, keep_levels=(), keep_layout=None,
                   rows_out=None, arena=None):
    """h0 (B*N, 256) = [relu](feature_rows(vols, points) @ w.T + bias) without materialising the feature rows (gather_fc0.hip).
    keep_levels: levels whose gathered columns are also stored -> (h0, rows); () -> (h0, None).
      keep_layout None: rows (B*N, layout.row_stride) in the full row layout, valid ONLY in those levels' columns and the
        padding columns (the rest is uninitialised memory);
      keep_layout = layout.subset(keep_levels): rows (B*N, keep_layout.row_stride), the compact kept-column matrix (every
        column valid, padding zero).
    rows_out: write the kept rows there instead of allocating; arena: a StepArena for the per-call workspace.
    = gather_fc0_prepare + gather_fc0_run."""
    B, N, _ = points.shape
    prep = gather_fc0_prepare(vols, layout, displacement, align_corners, w, B, N, keep_levels, keep_layout, arena)
    return gather_fc0_run(prep, points, bias, relu, rows_out)


def gather_bwd(vols, gvols, points, gfeat, layout, displacement, align_corners, want_gpoints=False, order=None,
               level_orders=None, flags=0, level_plans=None, skip_levels=()):
    """level_plans[l] (PullPlan or None): that level is scattered atomic-free in pull form and its gvol OVERWRITTEN
    (it may be uninitialised); the other levels accumulate into their (zeroed) gvol with float atomics.
    layout may be a KeptLayout (gfeat = gradient of the compact kept-column matrix; skip_levels = the levels it lacks)."""
    B, N, _ = points.shape
    if gfeat.numel() != B * N * layout.row_stride:
        raise RuntimeError(f"gather_bwd: gfeat {tuple(gfeat.shape)} does not match the layout's ({B * N}, {layout.row_stride})")
    d = make_gather_desc(vols, gvols, layout, B, N, displacement, align_corners, order, level_orders,
                         flags=flags | GATHER_FLAGS, level_plans=level_plans, skip_levels=skip_levels)
    gp = torch.empty_like(points) if want_gpoints else None
    check(_lib.lib().svr_gather_trilinear_bwd(C.byref(d), _p(points), _p(gfeat), _p(gp), _stream()), "gather_bwd")
    return gp


def corner_indices(vols, points, layout, level, displacement, align_corners):
    B, N, _ = points.shape
    d = make_gather_desc(vols, None, layout, B, N, displacement, align_corners)
    out = torch.empty(B, 7, N, 3, device=points.device, dtype=torch.int32)
    check(_lib.lib().svr_gather_corner_indices(C.byref(d), level, _p(points), _p(out), _stream()), "corner_indices")
    return out


# ------------------------------------------------------------------------------------------
# bf16-storage throughput mode of the query path (bf16_path.hip): separate entry points, never the default
# ------------------------------------------------------------------------------------------
def cast_bf16(t):
    """f32 -> bf16 (round to nearest even), same shape."""
    _f32(t)
    out = torch.empty(t.shape, device=t.device, dtype=torch.bfloat16)
    check(_lib.lib().svr_cast_f32_to_bf16(_p(t), _p(out), t.numel(), _stream()), "cast_bf16")
    return out


def _bf16(*ts):
    for t in ts:
        if t is not None and t.dtype != torch.bfloat16:
            raise RuntimeError(f"expected bfloat16, got {t.dtype}")


def gather_fwd_bf16(vols, points, layout, displacement, align_corners, order=None):
    """bf16 volumes (B,D,H,W,C) -> bf16 feature rows (B*N, row_stride); geometry and accumulation in f32."""
    B, N, _ = points.shape
    _f32(points)
    _bf16(*vols)
    d = GatherDesc()
    d.order = _p(order)
    d.n_levels, d.B, d.N, d.row_stride = len(vols), B, N, layout.row_stride
    d.align_corners, d.displacement = int(align_corners), displacement
    for l, v in enumerate(vols):
        if v.dim() != 5 or v.shape[0] != B or v.shape[4] != layout.channels[l]:
            raise RuntimeError(f"level {l}: expected (B,D,H,W,{layout.channels[l]}) channels-last, got {tuple(v.shape)}")
        L = d.level[l]
        L.vol, L.C, L.D, L.H, L.W, L.col = _p(v), v.shape[4], v.shape[1], v.shape[2], v.shape[3], layout.col[l]
    out = torch.empty(B * N, layout.row_stride, device=points.device, dtype=torch.bfloat16)
    check(_lib.lib().svr_gather_trilinear_fwd_bf16(C.byref(d), _p(points), _p(out), _stream()), "gather_fwd_bf16")
    return out


def _bf16_desc(vols, layout, B, N, displacement, align_corners):
    _bf16(*vols)
    d = GatherDesc()
    d.order = C.c_void_p(0)
    d.n_levels, d.B, d.N, d.row_stride = len(vols), B, N, layout.row_stride
    d.align_corners, d.displacement = int(align_corners), displacement
    for l, v in enumerate(vols):
        if v.dim() != 5 or v.shape[0] != B or v.shape[4] != layout.channels[l]:
            raise RuntimeError(f"level {l}: expected (B,D,H,W,{layout.channels[l]}) channels-last, got {tuple(v.shape)}")
        L = d.level[l]
        L.vol, L.C, L.D, L.H, L.W, L.col = _p(v), v.shape[4], v.shape[1], v.shape[2], v.shape[3], layout.col[l]
    return d


def gather_fc0_bf16_supported(vols, layout, displacement, align_corners, n_out):
    """The fused bf16-storage kernel covers these (bf16) volumes and fc_0's width?"""
    d = _bf16_desc(vols, layout, vols[0].shape[0], 1, displacement, align_corners)
    return n_out == 256 and bool(_lib.lib().svr_gather_fc0_supported(C.byref(d)))


def gather_fc0_bf16_prepare(vols, layout, displacement, align_corners, w):
    """w (n_out, >= layout.width) f32 -> one bf16 plane in the fused kernel's fragment order + the slab table; -> a handle
    for gather_fc0_bf16_run.  vols: the bf16 pyramid (IFNet.encode(x, storage="bf16"))."""
    _f32(w)
    B = vols[0].shape[0]
    d = _bf16_desc(vols, layout, B, 1, displacement, align_corners)
    l = _lib.lib()
    n_out = w.shape[0]
    assert w.stride(1) == 1 and w.shape[1] >= layout.width
    ws_bytes = l.svr_gather_fc0_workspace(C.byref(d), n_out)
    if ws_bytes <= 0:
        raise RuntimeError("gather_fc0_bf16: unsupported level shapes (see svr_gather_fc0_supported)")
    ws = torch.empty(ws_bytes, device=w.device, dtype=torch.uint8)
    check(l.svr_gather_fc0_bf16_prepare(C.byref(d), C.c_void_p(w.data_ptr()), w.stride(0), n_out, _p(ws), _stream()),
          "gather_fc0_bf16_prepare")
    return {"vols": vols, "layout": layout, "disp": displacement, "ac": align_corners, "n_out": n_out, "ws": ws}


def gather_fc0_bf16_run(prep, points, bias, relu=True):
    """-> h0 (B*N, n_out) bf16 = [relu](bf16 features @ bf16(w).T + bias), the features never leave the chip."""
    B, N, _ = points.shape
    _f32(points, bias)
    d = _bf16_desc(prep["vols"], prep["layout"], B, N, prep["disp"], prep["ac"])
    out = torch.empty(B * N, prep["n_out"], device=points.device, dtype=torch.bfloat16)
    epi = EPI_NONE if bias is None else (EPI_BIAS_RELU if relu else EPI_BIAS)
    check(_lib.lib().svr_gather_fc0_bf16_run(C.byref(d), _p(points), _p(bias), _p(out), out.stride(0), prep["n_out"], epi,
                                             _p(prep["ws"]), _stream()), "gather_fc0_bf16_run")
    return out


def linear_fwd_bf16(x, w, bias, relu=True):
    """y (bf16) = [relu](x (bf16) @ w (bf16).T + bias (f32)), f32 accumulation."""
    _bf16(x, w)
    _f32(bias)
    M, K = x.shape
    N = w.shape[0]
    assert w.shape[1] == K and x.stride(1) == 1 and w.stride(1) == 1
    out = torch.empty(M, N, device=x.device, dtype=torch.bfloat16)
    epi = EPI_NONE if bias is None else (EPI_BIAS_RELU if relu else EPI_BIAS)
    check(_lib.lib().svr_linear_fwd_bf16(C.c_void_p(x.data_ptr()), x.stride(0), C.c_void_p(w.data_ptr()), w.stride(0), _p(bias),
                                         _p(out), out.stride(0), M, N, K, epi, _stream()), "linear_fwd_bf16")
    return out


def fc_out_fwd_bf16(h, w, b, row_map=None):
    _bf16(h)
    _f32(w, b)
    M, K = h.shape
    out = torch.empty(M, device=h.device, dtype=torch.float32)
    check(_lib.lib().svr_fc_out_fwd_bf16(_p(h), h.stride(0), _p(w), _p(b), _p(out), _p(row_map), M, K, _stream()),
          "fc_out_fwd_bf16")
    return out


# ------------------------------------------------------------------------------------------
# point MLP
# ------------------------------------------------------------------------------------------
# Arithmetic of the forward GEMMs of the point MLP: "f16x3" (3-product f16 split with power-of-two scaling, as
# accurate as an f32 GEMM for |x| < 65504, gemm_f16x3.hip), "bf16x6" (6-product bf16 split, any f32 range,
# gemm_bf16x6.hip) or "f32" (exact-f32 MFMA).
FORWARD_GEMM = "f16x3"


class PreparedWeights:
    """Workspaces of the split-precision kernels, prepared AHEAD of the layer calls (include/svr_hip.h, PREPARE / RUN): the
    launches that only depend on the parameters (amax, split planes: 2-4 launch-bound kernels per layer and pass) run once
    per training step on a side stream instead of in front of every layer on the main stream.  Keyed by the parameter's
    storage address; valid from finish() until invalidate() -- i.e. for one forward + backward; every lookup outside that
    window, for a tensor that was not prepared, or for one that was modified in place since (its version counter moved), misses
    and the op prepares its own workspace as before."""

    def __init__(self):
        self._ws = {}
        self._valid = {}          # key -> (the parameter's version counter when its planes were made, index of their event)
        self._events = []
        self.ready = None
        self._waited = set()

    def _buf(self, key, nbytes, device):
        t = self._ws.get(key)
        if t is None or t.numel() < nbytes or t.device != device:
            t = self._ws[key] = torch.empty(nbytes, device=device, dtype=torch.uint8)
        return t

    def begin(self):
        self._valid.clear()
        self._events = []
        self._waited = set()      # (event index, stream): waits already enqueued

    def add_conv(self, w, which="both"):
        """w (Co,Ci,3,3,3): forward (f16x3) and backward-data (bf16x3 / f16x3s) planes, where those modes apply.  which: "fwd",
        "bwd" or "both" -- the caller prepares every layer's forward planes first and marks them (mark()), so that the forward
        pass waits for those only."""
        _f32(w)
        Co, Ci = w.shape[0], w.shape[1]
        l = _lib.lib()
        z = C.c_void_p(0)
        fwd, bwd = which in ("fwd", "both"), which in ("bwd", "both")
        if fwd and FORWARD_CONV == "f16x3" and Ci % 16 == 0:
            ws = self._buf(("cf", w.data_ptr()), l.svr_conv3d_fwd_f16x3_workspace(Ci, Co), w.device)
            check(l.svr_conv3d_k3_fwd_f16x3(z, _p(w), z, z, 0, 0, 0, 0, Ci, Co, EPI_NONE, _p(ws), _stream()), "conv3d_fwd_f16x3 prepare")
            self._valid[("cf", w.data_ptr())] = (w._version, len(self._events))
        if bwd and BACKWARD_CONV == "bf16x3" and Ci % 2 == 0 and Co % 16 == 0:
            ws = self._buf(("cb", w.data_ptr()), l.svr_conv3d_bwd_data_bf16x3_workspace(Ci, Co), w.device)
            check(l.svr_conv3d_k3_bwd_data_bf16x3(z, _p(w), z, 0, 0, 0, 0, Ci, Co, EPI_NONE, z, _p(ws), _stream()),
                  "conv3d_bwd_data_bf16x3 prepare")
            self._valid[("cb", w.data_ptr())] = (w._version, len(self._events))
        if bwd and BACKWARD_CONV == "f16x3s" and Ci % 2 == 0 and Co % 16 == 0:
            ws = self._buf(("cbh", w.data_ptr()), l.svr_conv3d_bwd_data_f16x3_workspace(Ci, Co), w.device)
            check(l.svr_conv3d_k3_bwd_data_f16x3(z, _p(w), z, 0, 0, 0, 0, Ci, Co, EPI_NONE, z, z, z, _p(ws), _stream()),
                  "conv3d_bwd_data_f16x3 prepare")
            self._valid[("cbh", w.data_ptr())] = (w._version, len(self._events))

    def add_linear(self, w, which="both"):
        """w (N,K) row-major: forward (f16x3) and backward-data (bf16x3 / f16x3s) planes, where those modes apply."""
        _f32(w)
        N, K = w.shape
        assert w.stride(1) == 1
        l = _lib.lib()
        z = C.c_void_p(0)
        fwd, bwd = which in ("fwd", "both"), which in ("bwd", "both")
        if fwd and FORWARD_GEMM == "f16x3" and K % 16 == 0:
            ws = self._buf(("lf", w.data_ptr()), l.svr_linear_fwd_f16x3_workspace(N, K), w.device)
            check(l.svr_linear_fwd_f16x3(z, 0, C.c_void_p(w.data_ptr()), w.stride(0), z, z, 0, 0, N, K, EPI_NONE, _p(ws), _stream()),
                  "linear_fwd_f16x3 prepare")
            self._valid[("lf", w.data_ptr())] = (w._version, len(self._events))
        if bwd and BACKWARD_GEMM == "bf16x3" and N % 32 == 0 and K % 4 == 0:
            ws = self._buf(("lb", w.data_ptr()), l.svr_linear_bwd_data_bf16x3_workspace(N, K), w.device)
            check(l.svr_linear_bwd_data_bf16x3(z, 0, C.c_void_p(w.data_ptr()), w.stride(0), z, 0, 0, N, K, EPI_NONE, z, 0, _p(ws),
                                               _stream()), "linear_bwd_data_bf16x3 prepare")
            self._valid[("lb", w.data_ptr())] = (w._version, len(self._events))
        if bwd and BACKWARD_GEMM == "f16x3s" and N % 16 == 0:
            ws = self._buf(("lbh", w.data_ptr()), l.svr_linear_bwd_data_f16x3_workspace(N, K), w.device)
            check(l.svr_linear_bwd_data_f16x3(z, 0, C.c_void_p(w.data_ptr()), w.stride(0), z, 0, 0, N, K, EPI_NONE, z, 0, z, z,
                                              _p(ws), _stream()), "linear_bwd_data_f16x3 prepare")
            self._valid[("lbh", w.data_ptr())] = (w._version, len(self._events))

    def mark(self, stream):
        """An event behind the planes added so far: a lookup waits for the event of ITS planes only (the forward pass does not
        wait for the backward planes -- 54 launch-bound kernels per step, 0.5 ms of side-stream time)."""
        ev = torch.cuda.Event()
        ev.record(stream)
        self._events.append(ev)

    def finish(self, stream):
        self.mark(stream)
        self.ready = self._events[-1]

    def invalidate(self):
        self._valid.clear()

    def lookup(self, kind, w):
        """The prepared workspace of `w` for `kind` ("cf" / "cb" / "lf" / "lb"), or None.  The first hit of a step makes
        the calling stream wait for the preparation (the step's later work on other streams is ordered behind it)."""
        key = (kind, w.data_ptr())
        ver, idx = self._valid.get(key, (-1, 0))
        if ver != w._version or idx >= len(self._events):      # not prepared, or modified in place since (an optimizer step)
            return None
        cur = torch.cuda.current_stream()
        if (idx, cur.cuda_stream) not in self._waited:     # once per event and stream
            cur.wait_event(self._events[idx])
            self._waited.add((idx, cur.cuda_stream))
        return self._ws[key]


_prepared = None     # the PreparedWeights of the training step in flight (model/ifnet.py), or None


def set_prepared(p):
    global _prepared
    _prepared = p


def _lookup(kind, w, mode, default):
    if _prepared is None or (mode is not None and mode != default):
        return None
    return _prepared.lookup(kind, w)


def linear_fwd(x, w, bias, relu=True, out=None, mode=None):
    """y = [relu](x @ w.T + bias); x (M,K) row stride may exceed K (padded feature rows)."""
    _f32(x, w, bias)
    M, K = x.shape
    N = w.shape[0]
    assert w.shape[1] == K and x.stride(1) == 1 and w.stride(1) == 1
    if out is None:
        out = torch.empty(M, N, device=x.device, dtype=torch.float32)
    epi = EPI_NONE if bias is None else (EPI_BIAS_RELU if relu else EPI_BIAS)
    if (mode or FORWARD_GEMM) == "f16x3" and K % 16 == 0:
        l = _lib.lib()
        ws = _lookup("lf", w, mode, "f16x3") if w.stride(0) == K else None
        wptr = C.c_void_p(0) if ws is not None else C.c_void_p(w.data_ptr())      # W NULL: the workspace is prepared
        if ws is None:
            ws = torch.empty(l.svr_linear_fwd_f16x3_workspace(N, K), device=x.device, dtype=torch.uint8)
        check(l.svr_linear_fwd_f16x3(C.c_void_p(x.data_ptr()), x.stride(0), wptr, w.stride(0), _p(bias),
                                     _p(out), out.stride(0), M, N, K, epi, _p(ws), _stream()), "linear_fwd_f16x3")
        return out
    if (mode or FORWARD_GEMM) == "bf16x6" and K % 16 == 0:
        l = _lib.lib()
        ws = torch.empty(l.svr_linear_fwd_bf16x6_workspace(N, K), device=x.device, dtype=torch.uint8)
        check(l.svr_linear_fwd_bf16x6(C.c_void_p(x.data_ptr()), x.stride(0), C.c_void_p(w.data_ptr()), w.stride(0), _p(bias),
                                      _p(out), out.stride(0), M, N, K, epi, _p(ws), _stream()), "linear_fwd_bf16x6")
        return out
    check(_lib.lib().svr_linear_fwd(C.c_void_p(x.data_ptr()), x.stride(0), C.c_void_p(w.data_ptr()), w.stride(0),
                                    _p(bias), _p(out), out.stride(0), M, N, K, epi, C.c_void_p(0), 0, _stream()),
          "linear_fwd")
    return out


# Arithmetic of the two backward GEMMs of the point MLP: "f16x3s" (DEFAULT since round 4: the SCALED 3-product f16 split --
# gradient operands brought into f16's range by an exact power of two taken from their |max|, 22 mantissa bits per operand =
# f32 level like the reference's fp32 backward, the same three matrix instructions per block; gemm_f16x3.hip /
# gemm_bf16x3.hip), "bf16x3" (3-term bf16 split, ~1.5e-5 relative per product: 4 % faster per step, the default of rounds
# 1-3; SVR_BACKWARD=bf16x3) or "f32" (exact-f32 MFMA).  The forward never uses them.
BACKWARD_GEMM = os.environ.get("SVR_BACKWARD", "f16x3s")
BACKWARD_MODES = ("bf16x3", "f16x3s", "f32")


class _AmaxSlots:
    """Zero-initialised int32 words for the atomic-maximum outputs of the "f16x3s" kernels, without a memset launch and an
    allocator call per word: one ring of 512 words per (device, stream), its halves zeroed alternately by ONE fill each
    time the ring enters them -- a half is reused ~250 words (several training steps) after its words were handed out, by
    when the tensors that referred to them are gone.  A word is produced and zeroed on the stream that owns the ring; a
    consumer on another stream is ordered behind the producer by the same event that orders it behind the tensor itself."""
    N = 512

    def __init__(self):
        self._rings = {}

    def take(self, device):
        if torch.cuda.is_current_stream_capturing():      # a replayed graph must zero its words itself
            return torch.zeros(1, device=device, dtype=torch.int32)
        key = (device, torch.cuda.current_stream(device).cuda_stream)
        r = self._rings.get(key)
        if r is None:
            r = self._rings[key] = {"buf": torch.zeros(self.N, device=device, dtype=torch.int32), "i": 0}
        i = r["i"]
        if i % (self.N // 2) == 0 and (i or r.get("lap")):    # entering a half whose words were used a lap ago
            r["buf"][i:i + self.N // 2].zero_()
        r["i"] = (i + 1) % self.N
        if r["i"] == 0:
            r["lap"] = True
        r["serial"] = r.get("serial", 0) + 1
        w = r["buf"][i:i + 1]
        w._svr_ring, w._svr_serial = r, r["serial"]       # (how old the word is: fresh() below)
        return w

    @staticmethod
    def fresh(w):
        """Is this word still the one that was handed out?  (A gradient tensor kept across many steps and fed to a later
        backward would otherwise be scaled by whatever the recycled word holds by then.)"""
        r = getattr(w, "_svr_ring", None)
        return r is None or r["serial"] - w._svr_serial < _AmaxSlots.N // 2 - 8


_amax_slots = _AmaxSlots()


def amax_slot(device):
    """A zeroed (1,) int32 word for an amax output."""
    return _amax_slots.take(device)


def amax_of(t):
    """(1,) int32 tensor = bit pattern of max |t| on the device (the power-of-two scale of a gradient operand of the
    "f16x3s" kernels).  Kernels that produce a gradient leave it on the tensor they return (attribute `_svr_amax`: no extra
    pass); otherwise one pass over t on the current stream, remembered on the tensor object."""
    a = getattr(t, "_svr_amax", None)
    if a is not None and _AmaxSlots.fresh(a):
        return a
    _f32(t)
    a = amax_slot(t.device)
    if t.dim() == 2 and t.stride(1) == 1 and t.shape[1] % 4 == 0 and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0:
        M, N, ld, src = t.shape[0], t.shape[1], t.stride(0), t
    else:
        src = t if t.is_contiguous() else t.contiguous()
        n = src.numel()
        if n % 4 or src.data_ptr() % 16:
            raise RuntimeError("amax_of: tensor must have a multiple of 4 elements and be 16-byte aligned")
        M, N, ld = n // 4, 4, 4
    check(_lib.lib().svr_amax_f32(C.c_void_p(src.data_ptr()), ld, M, N, _p(a), _stream()), "amax_f32")
    t._svr_amax = a
    return a


def linear_bwd_data(dy, w, mask=None, out=None, mode=None):
    """dx = (dy @ w) [* (mask > 0)]; dy (M,N), w (N,K)."""
    _f32(dy, w, mask)
    M, N = dy.shape
    K = w.shape[1]
    if out is None:
        out = torch.empty(M, K, device=dy.device, dtype=torch.float32)
    epi = EPI_MASK if mask is not None else EPI_NONE
    if ((mode or BACKWARD_GEMM) == "f16x3s" and N % 16 == 0 and dy.stride(1) == 1 and dy.stride(0) % 4 == 0
            and dy.data_ptr() % 16 == 0 and out.stride(1) == 1 and out.stride(0) % 4 == 0 and out.data_ptr() % 16 == 0
            and (mask is None or (mask.stride(1) == 1 and mask.stride(0) % 4 == 0 and mask.data_ptr() % 16 == 0))):
        l = _lib.lib()
        ws = _lookup("lbh", w, mode, "f16x3s") if (w.stride(0) == K and w.stride(1) == 1) else None
        wptr = C.c_void_p(0) if ws is not None else C.c_void_p(w.data_ptr())      # W NULL: the workspace is prepared
        if ws is None:
            assert w.stride(1) == 1
            ws = torch.empty(l.svr_linear_bwd_data_f16x3_workspace(N, K), device=dy.device, dtype=torch.uint8)
        amax_dy = amax_of(dy)
        amax_dx = amax_slot(dy.device)
        check(l.svr_linear_bwd_data_f16x3(C.c_void_p(dy.data_ptr()), dy.stride(0), wptr, w.stride(0),
                                          C.c_void_p(out.data_ptr()), out.stride(0), M, N, K, epi,
                                          C.c_void_p(mask.data_ptr()) if mask is not None else C.c_void_p(0),
                                          mask.stride(0) if mask is not None else 0, _p(amax_dy), _p(amax_dx), _p(ws), _stream()),
              "linear_bwd_data_f16x3")
        out._svr_amax = amax_dx        # |max| of what was stored: the next layer's scale, no extra pass
        return out
    if (mode or BACKWARD_GEMM) == "f16x3s":
        mode = "f32"                   # (shapes the f16 kernel does not take: exact f32, never a narrower split)
    if (mode or BACKWARD_GEMM) == "bf16x3" and N % 32 == 0:
        l = _lib.lib()
        ws = _lookup("lb", w, mode, "bf16x3") if (w.stride(0) == K and w.stride(1) == 1) else None
        wptr = C.c_void_p(0) if ws is not None else C.c_void_p(w.data_ptr())      # W NULL: the workspace is prepared
        if ws is None:
            ws = torch.empty(l.svr_linear_bwd_data_bf16x3_workspace(N, K), device=dy.device, dtype=torch.uint8)
        check(l.svr_linear_bwd_data_bf16x3(C.c_void_p(dy.data_ptr()), dy.stride(0), wptr, w.stride(0),
                                           C.c_void_p(out.data_ptr()), out.stride(0), M, N, K, epi,
                                           C.c_void_p(mask.data_ptr()) if mask is not None else C.c_void_p(0),
                                           mask.stride(0) if mask is not None else 0, _p(ws), _stream()),
              "linear_bwd_data_bf16x3")
        return out
    check(_lib.lib().svr_linear_bwd_data(C.c_void_p(dy.data_ptr()), dy.stride(0), C.c_void_p(w.data_ptr()), w.stride(0),
                                         C.c_void_p(out.data_ptr()), out.stride(0), M, N, K, epi,
                                         C.c_void_p(mask.data_ptr()) if mask is not None else C.c_void_p(0),
                                         mask.stride(0) if mask is not None else 0, _stream()), "linear_bwd_data")
    return out


def linear_bwd_weight(dy, x, want_bias=True, mode=None):
    """dW (N,K) = dy.T @ x, db (N) = dy.sum(0)."""
    _f32(dy, x)
    M, N = dy.shape
    K = x.shape[1]
    l = _lib.lib()
    dw = torch.empty(N, K, device=dy.device, dtype=torch.float32)
    db = torch.empty(N, device=dy.device, dtype=torch.float32) if want_bias else None
    if ((mode or BACKWARD_GEMM) == "f16x3s" and N % 4 == 0 and K % 4 == 0 and N >= 4 and K >= 4 and M > 0 and dy.stride(1) == 1
            and x.stride(1) == 1 and dy.stride(0) % 4 == 0 and x.stride(0) % 4 == 0 and dy.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0):
        ws = torch.empty(l.svr_linear_bwd_weight_f16x3_workspace(M, N, K), device=dy.device, dtype=torch.uint8)
        check(l.svr_linear_bwd_weight_f16x3(C.c_void_p(dy.data_ptr()), dy.stride(0), C.c_void_p(x.data_ptr()), x.stride(0),
                                            _p(dw), dw.stride(0), _p(db), M, N, K, _p(amax_of(dy)), _p(ws), _stream()),
              "linear_bwd_weight_f16x3")
        return dw, db
    if (mode or BACKWARD_GEMM) == "f16x3s":
        mode = "f32"
    if (mode or BACKWARD_GEMM) == "bf16x3" and N % 4 == 0:
        ws = torch.empty(l.svr_linear_bwd_weight_bf16x3_workspace(M, N, K), device=dy.device, dtype=torch.uint8)
        check(l.svr_linear_bwd_weight_bf16x3(C.c_void_p(dy.data_ptr()), dy.stride(0), C.c_void_p(x.data_ptr()), x.stride(0),
                                             _p(dw), dw.stride(0), _p(db), M, N, K, _p(ws), _stream()),
              "linear_bwd_weight_bf16x3")
        return dw, db
    ws = torch.empty(l.svr_linear_bwd_weight_workspace(M, N, K), device=dy.device, dtype=torch.uint8)
    check(l.svr_linear_bwd_weight(C.c_void_p(dy.data_ptr()), dy.stride(0), C.c_void_p(x.data_ptr()), x.stride(0),
                                  _p(dw), dw.stride(0), _p(db), M, N, K, _p(ws), _stream()), "linear_bwd_weight")
    return dw, db


def fc_out_fwd(h, w, b, row_map=None):
    """logits[row_map[m]] (or [m]) = h[m] . w + b"""
    _f32(h, w, b)
    M, K = h.shape
    out = torch.empty(M, device=h.device, dtype=torch.float32)
    check(_lib.lib().svr_fc_out_fwd(_p(h), h.stride(0), _p(w), _p(b), _p(out), _p(row_map), M, K, _stream()), "fc_out_fwd")
    return out


def fc_out_bwd(h, w, dlogits, row_map=None):
    _f32(h, w, dlogits)
    M, K = h.shape
    l = _lib.lib()
    ws = torch.empty(l.svr_fc_out_bwd_workspace(M, K), device=h.device, dtype=torch.uint8)
    dh = torch.empty(M, K, device=h.device, dtype=torch.float32)
    dw = torch.empty(K, device=h.device, dtype=torch.float32)
    db = torch.empty(1, device=h.device, dtype=torch.float32)
    amax = amax_slot(h.device) if BACKWARD_GEMM == "f16x3s" else None       # |max| of dh for the next layer's scaled split
    check(l.svr_fc_out_bwd(_p(h), h.stride(0), _p(w), _p(dlogits), _p(row_map), _p(dh), dh.stride(0), _p(dw), _p(db), M, K,
                           _p(amax), _p(ws), _stream()), "fc_out_bwd")
    if amax is not None:
        dh._svr_amax = amax
    return dh, dw, db


def bce_logits_sum_mean(logits, targets, want_grad=True, gscale=1.0):
    _f32(logits, targets)
    B, N = logits.shape
    loss = torch.empty(1, device=logits.device, dtype=torch.float32)
    dz = torch.empty_like(logits) if want_grad else None
    ws = torch.empty(B, device=logits.device, dtype=torch.float64)
    check(_lib.lib().svr_bce_logits_sum_mean(_p(logits), _p(targets), _p(loss), _p(dz), B, N, gscale, _p(ws), _stream()),
          "bce")
    return loss, dz


# ------------------------------------------------------------------------------------------
# encoder: conv3d / batch-norm / max-pool (channels-last volumes (B,D,H,W,C))
# ------------------------------------------------------------------------------------------
def conv3d_pack_weight(w, want_bwd=True):
    """(Co,Ci,3,3,3) -> Wp_fwd [27][Ci][Co], Wp_bwd [27][Co][Ci] (taps flipped)."""
    _f32(w)
    Co, Ci = w.shape[0], w.shape[1]
    wf = torch.empty(27, Ci, Co, device=w.device, dtype=torch.float32)
    wb = torch.empty(27, Co, Ci, device=w.device, dtype=torch.float32) if want_bwd else None
    check(_lib.lib().svr_conv3d_pack_weight(_p(w), _p(wf), _p(wb), Ci, Co, _stream()), "conv3d_pack_weight")
    return wf, wb


def conv3d_unpack_wgrad(dwp, Ci, Co):
    dw = torch.empty(Co, Ci, 3, 3, 3, device=dwp.device, dtype=torch.float32)
    check(_lib.lib().svr_conv3d_unpack_wgrad(_p(dwp), _p(dw), Ci, Co, _stream()), "conv3d_unpack_wgrad")
    return dw


def conv3d_k3(x, wp, bias=None, relu=False, mask=None):
    """x (B,D,H,W,Ci), wp [27][Ci][Co] -> (B,D,H,W,Co); epilogue: bias[+relu] or mask or none."""
    _f32(x, wp, bias, mask)
    B, D, H, W, Ci = x.shape
    Co = wp.shape[2]
    assert wp.shape[1] == Ci
    out = torch.empty(B, D, H, W, Co, device=x.device, dtype=torch.float32)
    if mask is not None:
        epi = EPI_MASK
    elif bias is not None:
        epi = EPI_BIAS_RELU if relu else EPI_BIAS
    else:
        epi = EPI_NONE
    check(_lib.lib().svr_conv3d_k3(_p(x), _p(wp), _p(bias), _p(out), B, D, H, W, Ci, Co, epi, _p(mask), _stream()), "conv3d_k3")
    return out


# Arithmetic of the encoder's backward-data convolutions: "bf16x3" (conv3d_bf16.hip) or "f32".
BACKWARD_CONV = os.environ.get("SVR_BACKWARD", "f16x3s")     # "bf16x3", "f16x3s" (scaled f16 split: f32 level) or "f32"
# ... and of its forward convolutions: "f16x3" (3-product f16 split, f32-level accuracy for |x| < 65504), "bf16x6"
# (6-product bf16 split, any f32 range) or "f32" (exact-f32 MFMA)
FORWARD_CONV = "f16x3"


class StatParts:
    """Per-workgroup partial sums (sum, sum of squares; float64 [blocks][2][C]) of a tensor a kernel has just written: the
    BatchNorm that follows finalizes from them (bn_forward(stats=...)) instead of reading the tensor again."""

    def __init__(self, part, blocks):
        self.part, self.blocks = part, blocks


def conv3d_k3_fwd(x, w, bias, relu=True, mode=None, want_stats=False):
    """relu(conv(x (B,D,H,W,Ci), w (Co,Ci,3,3,3)) + bias) -> (B,D,H,W,Co).  want_stats: -> (out, StatParts or None): the
    BatchNorm statistics of the output come out of the kernel's epilogue where the f16x3 kernel runs."""
    _f32(x, w, bias)
    B, D, H, W, Ci = x.shape
    Co = w.shape[0]
    if (mode or FORWARD_CONV) == "f16x3" and Ci % 16 == 0:
        l = _lib.lib()
        out = torch.empty(B, D, H, W, Co, device=x.device, dtype=torch.float32)
        ws = _lookup("cf", w, mode, "f16x3")
        wptr = C.c_void_p(0) if ws is not None else _p(w)                          # W NULL: the workspace is prepared
        if ws is None:
            ws = torch.empty(l.svr_conv3d_fwd_f16x3_workspace(Ci, Co), device=x.device, dtype=torch.uint8)
        epi = EPI_BIAS_RELU if relu else EPI_BIAS
        if want_stats:
            blocks = l.svr_conv3d_fwd_f16x3_stats_blocks(B, D, H, W, Ci, Co)
            part = torch.empty(blocks, 2, Co, device=x.device, dtype=torch.float64)
            check(l.svr_conv3d_k3_fwd_f16x3_stats(_p(x), wptr, _p(bias), _p(out), _p(part), B, D, H, W, Ci, Co, epi, _p(ws),
                                                  _stream()), "conv3d_fwd_f16x3_stats")
            return out, StatParts(part, blocks)
        check(l.svr_conv3d_k3_fwd_f16x3(_p(x), wptr, _p(bias), _p(out), B, D, H, W, Ci, Co, epi, _p(ws), _stream()),
              "conv3d_fwd_f16x3")
        return out
    if want_stats:
        return conv3d_k3_fwd(x, w, bias, relu, mode), None
    if (mode or FORWARD_CONV) == "bf16x6" and Ci % 16 == 0:
        l = _lib.lib()
        out = torch.empty(B, D, H, W, Co, device=x.device, dtype=torch.float32)
        ws = torch.empty(l.svr_conv3d_fwd_bf16x6_workspace(Ci, Co), device=x.device, dtype=torch.uint8)
        check(l.svr_conv3d_k3_fwd_bf16x6(_p(x), _p(w), _p(bias), _p(out), B, D, H, W, Ci, Co,
                                         EPI_BIAS_RELU if relu else EPI_BIAS, _p(ws), _stream()), "conv3d_fwd_bf16x6")
        return out
    wf, _ = conv3d_pack_weight(w, want_bwd=False)
    return conv3d_k3(x, wf, bias, relu=relu)


def conv3d_c1_fwd_stats(x, w, bias, relu=True):
    """conv_in (Ci == 1) forward + the BatchNorm statistics of its output: (out, stats float64 [mean | biased var])."""
    _f32(x, w, bias)
    B, D, H, W, Ci = x.shape
    Co = w.shape[0]
    assert Ci == 1 and Co in (16, 32)
    wf, _ = conv3d_pack_weight(w, want_bwd=False)
    l = _lib.lib()
    out = torch.empty(B, D, H, W, Co, device=x.device, dtype=torch.float32)
    stats = torch.empty(2 * Co, device=x.device, dtype=torch.float64)
    ws = torch.empty(l.svr_conv3d_c1_fwd_stats_workspace(B, D, H, W, Co), device=x.device, dtype=torch.uint8)
    check(l.svr_conv3d_c1_fwd_stats(_p(x), _p(wf), _p(bias), _p(out), _p(stats), B, D, H, W, Co,
                                    EPI_BIAS_RELU if relu else EPI_BIAS, _p(ws), _stream()), "conv3d_c1_fwd_stats")
    return out, stats


def conv3d_k3_bwd_data(dout, w, mask=None, mode=None):
    """din (B,D,H,W,Ci) = conv^T(dout (B,D,H,W,Co), w (Co,Ci,3,3,3)) [* (mask > 0)]."""
    _f32(dout, w, mask)
    B, D, H, W, Co = dout.shape
    Ci = w.shape[1]
    assert w.shape[0] == Co
    if (mode or BACKWARD_CONV) == "f16x3s" and Ci % 2 == 0 and Co % 16 == 0 and dout.is_contiguous() and dout.numel() % 4 == 0:
        l = _lib.lib()
        din = torch.empty(B, D, H, W, Ci, device=dout.device, dtype=torch.float32)
        ws = _lookup("cbh", w, mode, "f16x3s")
        wptr = C.c_void_p(0) if ws is not None else _p(w)                          # W NULL: the workspace is prepared
        if ws is None:
            ws = torch.empty(l.svr_conv3d_bwd_data_f16x3_workspace(Ci, Co), device=dout.device, dtype=torch.uint8)
        amax_din = amax_slot(dout.device)
        check(l.svr_conv3d_k3_bwd_data_f16x3(_p(dout), wptr, _p(din), B, D, H, W, Ci, Co,
                                             EPI_MASK if mask is not None else EPI_NONE, _p(mask), _p(amax_of(dout)), _p(amax_din),
                                             _p(ws), _stream()), "conv3d_bwd_data_f16x3")
        din._svr_amax = amax_din
        return din
    if (mode or BACKWARD_CONV) == "f16x3s":
        mode = "f32"                   # (shapes the f16 kernel does not take: exact f32, never a narrower split)
    if (mode or BACKWARD_CONV) == "bf16x3" and Ci % 2 == 0 and Co % 16 == 0:
        l = _lib.lib()
        din = torch.empty(B, D, H, W, Ci, device=dout.device, dtype=torch.float32)
        ws = _lookup("cb", w, mode, "bf16x3")
        wptr = C.c_void_p(0) if ws is not None else _p(w)                          # W NULL: the workspace is prepared
        if ws is None:
            ws = torch.empty(l.svr_conv3d_bwd_data_bf16x3_workspace(Ci, Co), device=dout.device, dtype=torch.uint8)
        check(l.svr_conv3d_k3_bwd_data_bf16x3(_p(dout), wptr, _p(din), B, D, H, W, Ci, Co,
                                              EPI_MASK if mask is not None else EPI_NONE, _p(mask), _p(ws), _stream()),
              "conv3d_bwd_data_bf16x3")
        return din
    _, wb = conv3d_pack_weight(w, want_bwd=True)
    return conv3d_k3(dout, wb, mask=mask)


# Arithmetic of the encoder's weight gradients: "bf16x3" (conv3d_bwdw_bf16.hip), "f16x3s" (its scaled f16 form: f32 level)
# or "f32" (exact-f32 MFMA)
BACKWARD_CONV_WEIGHT = os.environ.get("SVR_BACKWARD", "f16x3s")


def conv3d_k3_bwd_weight(x, dout, want_bias=True, mode=None, param_layout=False):
    """dWp [27][Ci][Co], db (Co).  param_layout: the weight gradient in the parameter's layout (Co,Ci,3,3,3) instead (the
    bf16x3 kernel writes it directly; the other paths unpack)."""
    _f32(x, dout)
    B, D, H, W, Ci = x.shape
    Co = dout.shape[4]
    l = _lib.lib()
    # the split-precision kernel's 32-bit offsets inside one sample (conv3d_bwdw_bf16.hip); larger volumes take the f32 kernel
    fits = D * H * W <= (1 << 24) and D * H * W * max(Ci, Co) < (1 << 30)
    if not fits and (mode or BACKWARD_CONV_WEIGHT) in ("bf16x3", "f16x3s"):
        mode = "f32"
    if (mode or BACKWARD_CONV_WEIGHT) == "f16x3s" and Ci % 4 == 0 and Co % 4 == 0 and dout.is_contiguous() and dout.numel() % 4 == 0:
        ws = torch.empty(l.svr_conv3d_k3_bwd_weight_bf16x3_workspace(B, D, H, W, Ci, Co), device=x.device, dtype=torch.uint8)
        dw = torch.empty((Co, Ci, 3, 3, 3) if param_layout else (27, Ci, Co), device=x.device, dtype=torch.float32)
        db = torch.empty(Co, device=x.device, dtype=torch.float32) if want_bias else None
        check(l.svr_conv3d_k3_bwd_weight_f16x3(_p(x), _p(dout), _p(dw), _p(db), B, D, H, W, Ci, Co, 1 if param_layout else 0,
                                               _p(amax_of(dout)), _p(ws), _stream()), "conv3d_k3_bwd_weight_f16x3")
        return dw, db
    if (mode or BACKWARD_CONV_WEIGHT) == "f16x3s":
        mode = "f32"
    if (mode or BACKWARD_CONV_WEIGHT) == "bf16x3" and Ci % 4 == 0 and Co % 4 == 0 and param_layout:
        ws = torch.empty(l.svr_conv3d_k3_bwd_weight_bf16x3_workspace(B, D, H, W, Ci, Co), device=x.device, dtype=torch.uint8)
        dw = torch.empty(Co, Ci, 3, 3, 3, device=x.device, dtype=torch.float32)
        db = torch.empty(Co, device=x.device, dtype=torch.float32) if want_bias else None
        check(l.svr_conv3d_k3_bwd_weight_bf16x3_param(_p(x), _p(dout), _p(dw), _p(db), B, D, H, W, Ci, Co, _p(ws), _stream()),
              "conv3d_k3_bwd_weight_bf16x3_param")
        return dw, db
    if param_layout:
        dwp, db = conv3d_k3_bwd_weight(x, dout, want_bias, mode)
        return conv3d_unpack_wgrad(dwp, Ci, Co), db
    if (mode or BACKWARD_CONV_WEIGHT) == "bf16x3" and Ci % 4 == 0 and Co % 4 == 0:
        ws = torch.empty(l.svr_conv3d_k3_bwd_weight_bf16x3_workspace(B, D, H, W, Ci, Co), device=x.device, dtype=torch.uint8)
        dwp = torch.empty(27, Ci, Co, device=x.device, dtype=torch.float32)
        db = torch.empty(Co, device=x.device, dtype=torch.float32) if want_bias else None
        check(l.svr_conv3d_k3_bwd_weight_bf16x3(_p(x), _p(dout), _p(dwp), _p(db), B, D, H, W, Ci, Co, _p(ws), _stream()),
              "conv3d_k3_bwd_weight_bf16x3")
        return dwp, db
    ws = torch.empty(l.svr_conv3d_k3_bwd_weight_workspace(B, D, H, W, Ci, Co), device=x.device, dtype=torch.uint8)
    dwp = torch.empty(27, Ci, Co, device=x.device, dtype=torch.float32)
    db = torch.empty(Co, device=x.device, dtype=torch.float32) if want_bias else None
    check(l.svr_conv3d_k3_bwd_weight(_p(x), _p(dout), _p(dwp), _p(db), B, D, H, W, Ci, Co, _p(ws), _stream()),
          "conv3d_k3_bwd_weight")
    return dwp, db


def bn_forward(x, gamma, beta, running_mean, running_var, training, eps=1e-5, momentum=0.1, want_pool=True, stats=None):
    """x (B,D,H,W,C) -> y, pooled, argmax, scale_shift(3C), mean(C).  `stats` (float64 [mean | biased var]): statistics
    of x already computed by the producer (conv3d_c1_fwd_stats); the statistics pass over x is then skipped."""
    _f32(x, gamma, beta, running_mean, running_var)
    B, D, H, W, Cc = x.shape
    rows = B * D * H * W
    if training and rows <= 1:
        # same error as torch.nn.functional.batch_norm (the reference raises here too)
        raise ValueError(f"Expected more than 1 value per channel when training, got input size "
                         f"torch.Size([{B}, {Cc}, {D}, {H}, {W}])")
    l = _lib.lib()
    dev = x.device
    ss = torch.empty(3 * Cc, device=dev, dtype=torch.float32)
    mean = torch.empty(Cc, device=dev, dtype=torch.float32)
    if training and isinstance(stats, StatParts):      # partial sums from the producing kernel's epilogue: one launch
        check(l.svr_bn_finalize_parts(_p(stats.part), stats.blocks, C.c_void_p(0), _p(gamma), _p(beta), _p(running_mean),
                                      _p(running_var), _p(ss), _p(mean), rows, Cc, eps, momentum, _stream()), "bn_finalize_parts")
    elif training and stats is None:      # statistics + finalize: two launches
        ws = torch.empty(l.svr_bn_stats_workspace(rows, Cc), device=dev, dtype=torch.uint8)
        check(l.svr_bn_stats_finalize(_p(x), C.c_void_p(0), _p(gamma), _p(beta), _p(running_mean), _p(running_var), _p(ss), _p(mean),
                                      rows, Cc, eps, momentum, _p(ws), _stream()), "bn_stats_finalize")
    else:
        check(l.svr_bn_finalize(_p(stats if training else None), _p(gamma), _p(beta), _p(running_mean), _p(running_var), _p(ss),
                                _p(mean), rows, Cc, eps, momentum, int(training), _stream()), "bn_finalize")
    y = torch.empty_like(x)
    pooled = argmax = None
    if want_pool:
        pooled = torch.empty(B, D // 2, H // 2, W // 2, Cc, device=dev, dtype=torch.float32)
        argmax = torch.empty(B, D // 2, H // 2, W // 2, Cc, device=dev, dtype=torch.uint8)
    check(l.svr_bn_apply_pool(_p(x), _p(ss), _p(y), _p(pooled), _p(argmax), B, D, H, W, Cc, _stream()), "bn_apply_pool")
    return y, pooled, argmax, ss, mean


def bn_backward(x, dy, dpooled, argmax, mean, ss, relu_mask=True, training=True):
    """-> dx (grad wrt the conv pre-activation when relu_mask), dgamma, dbeta.  training=False: the forward normalised
    with the running statistics (eval mode), so dx has no batch-mean terms."""
    _f32(x, dy, dpooled, mean, ss)
    B, D, H, W, Cc = x.shape
    l = _lib.lib()
    dev = x.device
    sums = torch.empty(2 * Cc, device=dev, dtype=torch.float64)
    ws = torch.empty(l.svr_bn_stats_workspace(B * D * H * W, Cc), device=dev, dtype=torch.uint8)
    check(l.svr_bn_bwd_reduce(_p(x), _p(dy), _p(dpooled), _p(argmax), _p(mean), _p(ss), _p(sums), B, D, H, W, Cc, _p(ws),
                              _stream()), "bn_bwd_reduce")
    dx = torch.empty_like(x)
    dgamma = torch.empty(Cc, device=dev, dtype=torch.float32)
    dbeta = torch.empty(Cc, device=dev, dtype=torch.float32)
    amax = amax_slot(dev) if "f16x3s" in (BACKWARD_CONV, BACKWARD_CONV_WEIGHT, BACKWARD_GEMM) else None
    check(l.svr_bn_bwd_apply(_p(x), _p(dy), _p(dpooled), _p(argmax), _p(mean), _p(ss), C.c_void_p(0), _p(sums), _p(dx),
                             _p(dgamma), _p(dbeta), B, D, H, W, Cc, int(relu_mask) | (0 if training else 2), _p(amax), _stream()),
          "bn_bwd_apply")
    if amax is not None:
        dx._svr_amax = amax         # |max| of dx: the scale of the convolution gradients' scaled f16 split, no extra pass
    return dx, dgamma, dbeta


def stage1_supported(x, Co):
    """The recomputed first stage (stage1.hip) covers this input grid?  x (B,D,H,W,1)."""
    B, D, H, W, Ci = x.shape
    return Ci == 1 and bool(_lib.lib().svr_stage1_supported(B, D, H, W, Co))


def stage1_arith(mode=None):
    """1 if the first stage's recomputed convolution runs as the f16 split (FORWARD_CONV == "f16x3", the default), 0 = exact
    f32.  The backward recomputes the forward's activation, so it must get the forward's value (stage1_fwd stores it on wp)."""
    return 1 if (mode or FORWARD_CONV) == "f16x3" else 0


def stage1_fwd(x, w, bias, gamma, beta, running_mean, running_var, training, eps=1e-5, momentum=0.1, want_pool=True, mode=None):
    """x (B,D,H,W,1) -> y = BN(relu(conv_in(x))), pooled, argmax, scale_shift (3*16), mean (16), wp: conv_in's activation
    is recomputed wherever it is needed and never stored (stage1.hip).  mode: "f16x3" / "f32" (default: FORWARD_CONV)."""
    _f32(x, w, bias, gamma, beta, running_mean, running_var)
    B, D, H, W, _ = x.shape
    Co = w.shape[0]
    if training and B * D * H * W <= 1:
        raise ValueError(f"Expected more than 1 value per channel when training, got input size torch.Size([{B}, {Co}, {D}, {H}, {W}])")
    l = _lib.lib()
    dev = x.device
    wp, _ = conv3d_pack_weight(w, want_bwd=False)
    y = torch.empty(B, D, H, W, Co, device=dev, dtype=torch.float32)
    pooled = argmax = None
    if want_pool:
        pooled = torch.empty(B, D // 2, H // 2, W // 2, Co, device=dev, dtype=torch.float32)
        argmax = torch.empty(B, D // 2, H // 2, W // 2, Co, device=dev, dtype=torch.uint8)
    ss = torch.empty(3 * Co, device=dev, dtype=torch.float32)
    mean = torch.empty(Co, device=dev, dtype=torch.float32)
    stats = torch.empty(2 * Co, device=dev, dtype=torch.float64) if training else None
    ws = torch.empty(l.svr_stage1_workspace(B, D, H, W), device=dev, dtype=torch.uint8)
    check(l.svr_stage1_fwd(_p(x), _p(wp), _p(bias), _p(gamma), _p(beta), _p(running_mean), _p(running_var), _p(y), _p(pooled),
                           _p(argmax), _p(ss), _p(mean), _p(stats), B, D, H, W, Co, eps, momentum, int(training),
                           stage1_arith(mode), _p(ws), _stream()), "stage1_fwd")
    wp.svr_stage1_arith = stage1_arith(mode)     # the arithmetic the backward has to recompute with
    return y, pooled, argmax, ss, mean, wp


def stage1_bwd(x, wp, bias, dy, dpooled, argmax, mean, ss, relu_mask=True, training=True, want_dout=False):
    """Backward of stage1_fwd -> dgamma, dbeta, dW (16,1,3,3,3), db, dout (None unless want_dout).  wp: the tensor
    stage1_fwd returned (it carries the forward's arithmetic)."""
    _f32(x, wp, bias, dy, dpooled, mean, ss)
    B, D, H, W, _ = x.shape
    Co = wp.shape[2]
    l = _lib.lib()
    dev = x.device
    sums = torch.empty(2 * Co, device=dev, dtype=torch.float64)
    dgamma = torch.empty(Co, device=dev, dtype=torch.float32)
    dbeta = torch.empty(Co, device=dev, dtype=torch.float32)
    dwp = torch.empty(Co, 1, 3, 3, 3, device=dev, dtype=torch.float32)      # the parameter's layout
    db = torch.empty(Co, device=dev, dtype=torch.float32)
    dout = torch.empty(B, D, H, W, Co, device=dev, dtype=torch.float32) if want_dout else None
    ws = torch.empty(l.svr_stage1_workspace(B, D, H, W), device=dev, dtype=torch.uint8)
    check(l.svr_stage1_bwd(_p(x), _p(wp), _p(bias), _p(dy), _p(dpooled), _p(argmax), _p(mean), _p(ss), _p(sums), _p(dgamma),
                           _p(dbeta), _p(dwp), _p(db), _p(dout), B, D, H, W, Co, int(relu_mask) | (0 if training else 2),
                           int(getattr(wp, "svr_stage1_arith", stage1_arith())), _p(ws), _stream()), "stage1_bwd")
    return dgamma, dbeta, dwp, db, dout


# ------------------------------------------------------------------------------------------
# projection: unproject / splat / clamp / blur
# ------------------------------------------------------------------------------------------
def _consts(c):
    arr = (C.c_float * 12)(*[float(v) for v in c])
    return arr


def unproject(depth, consts, normalize):
    _f32(depth)
    B, Hi, Wi = depth.shape
    pc = torch.empty(B, Hi * Wi, 3, device=depth.device, dtype=torch.float32)
    check(_lib.lib().svr_unproject_fwd(_p(depth), _p(pc), B, Hi, Wi, _consts(consts), int(normalize), _stream()), "unproject_fwd")
    return pc


def unproject_bwd(depth, gpc, consts, normalize):
    B, Hi, Wi = depth.shape
    gd = torch.empty_like(depth)
    check(_lib.lib().svr_unproject_bwd(_p(depth), _p(gpc), _p(gd), B, Hi, Wi, _consts(consts), int(normalize), _stream()),
          "unproject_bwd")
    return gd


def splat_fwd(pts, dims, want_indices=False):
    _f32(pts)
    B, N, _ = pts.shape
    acc = torch.zeros(B, *dims, device=pts.device, dtype=torch.float32)
    base = torch.empty(B, N, 3, device=pts.device, dtype=torch.int32) if want_indices else None
    valid = torch.empty(B, N, device=pts.device, dtype=torch.uint8) if want_indices else None
    check(_lib.lib().svr_voxelize_splat_fwd(_p(pts), _p(acc), _p(base), _p(valid), B, N, dims[0], dims[1], dims[2], _stream()),
          "splat_fwd")
    return acc, base, valid


def splat_bwd(pts, gacc, dims):
    B, N, _ = pts.shape
    gp = torch.empty_like(pts)
    check(_lib.lib().svr_voxelize_splat_bwd(_p(pts), _p(gacc), _p(gp), B, N, dims[0], dims[1], dims[2], _stream()), "splat_bwd")
    return gp


def scale_clamp01(x, scale):
    out = torch.empty_like(x)
    check(_lib.lib().svr_scale_clamp01_fwd(_p(x), _p(out), x.numel(), scale, _stream()), "scale_clamp01_fwd")
    return out


def scale_clamp01_bwd(x, gout, scale):
    gin = torch.empty_like(x)
    check(_lib.lib().svr_scale_clamp01_bwd(_p(x), _p(gout), _p(gin), x.numel(), scale, _stream()), "scale_clamp01_bwd")
    return gin


def blur_axis(x, taps, axis):
    _f32(x, taps)
    B, D0, D1, D2 = x.shape
    out = torch.empty_like(x)
    check(_lib.lib().svr_blur_axis_fwd(_p(x), _p(taps), _p(out), B, D0, D1, D2, axis, taps.numel(), _stream()), "blur_fwd")
    return out


def blur_axis_bwd(x, taps, gout, axis, want_gin=True, want_gtaps=True):
    B, D0, D1, D2 = x.shape
    gin = torch.empty_like(x) if want_gin else None
    gt = torch.zeros(taps.numel(), device=x.device, dtype=torch.float64) if want_gtaps else None
    check(_lib.lib().svr_blur_axis_bwd(_p(x), _p(taps), _p(gout), _p(gin), _p(gt), B, D0, D1, D2, axis, taps.numel(),
                                       _stream()), "blur_bwd")
    return gin, gt


# ------------------------------------------------------------------------------------------
# UNet 2-D convolution blocks (conv2d.hip): act -> [x2 upsample] -> conv on cat(src0, src1), channels-last
# ------------------------------------------------------------------------------------------
ACT_NONE, ACT_LEAKY, ACT_RELU = 0, 1, 2


def _conv2d_desc(src0, src1, k, stride, act, up):
    _f32(src0, src1)
    B, H, W, C0 = src0.shape
    d = _lib.Conv2dDesc(_p(src0), _p(src1), B, H, W, C0, src1.shape[3] if src1 is not None else 0, k, stride, act, int(up))
    Hv, Wv = (2 * H, 2 * W) if up else (H, W)
    pad = 0 if k == 1 else 1
    Ho, Wo = (Hv + 2 * pad - k) // stride + 1, (Wv + 2 * pad - k) // stride + 1
    return d, (Hv, Wv, Ho, Wo)


def conv2d_im2col(src0, src1, k, stride, act, up):
    """-> col (B*Ho*Wo, k*k*C) float32 and (Ho, Wo)."""
    d, (Hv, Wv, Ho, Wo) = _conv2d_desc(src0, src1, k, stride, act, up)
    C_ = d.C0 + d.C1
    col = torch.empty(d.B * Ho * Wo, k * k * C_, device=src0.device, dtype=torch.float32)
    check(_lib.lib().svr_conv2d_im2col(C.byref(d), _p(col), _stream()), "conv2d_im2col")
    return col, (Ho, Wo)


def conv2d_col2im(src0, src1, k, stride, act, up, dcol, need0=True, need1=True):
    """dcol (B*Ho*Wo, k*k*C) -> gradients wrt src0 / src1 (None where not needed)."""
    d, (Hv, Wv, Ho, Wo) = _conv2d_desc(src0, src1, k, stride, act, up)
    C_ = d.C0 + d.C1
    dvirt = torch.empty(d.B, Hv, Wv, C_, device=src0.device, dtype=torch.float32)
    d0 = torch.empty_like(src0) if need0 else None
    d1 = torch.empty_like(src1) if (src1 is not None and need1) else None
    if d0 is None and d1 is None:
        return None, None
    check(_lib.lib().svr_conv2d_col2im(C.byref(d), _p(dcol), _p(dvirt), _p(d0), _p(d1), _stream()), "conv2d_col2im")
    return d0, d1


# ---- the same blocks as implicit GEMMs (conv2d_igemm.hip; the default of model/unet.py) -------------------------------------
UNET_IGEMM = os.environ.get("SVR_UNET_IGEMM", "1") != "0"      # "0": the explicit patch-matrix path above (A/B)


class Conv2dPlanes:
    """Split f16 planes of one nn.Conv2d weight (forward and backward-data products) + the word holding max|W|.  Layers with
    1..4 output channels keep the f32 weight itself (`small`: plain-FMA kernels, conv2d.hip)."""

    def __init__(self, weight, stride, want_bwd=True):
        _f32(weight)
        w = weight.detach().contiguous()
        self.Cout, self.C, self.k = int(w.shape[0]), int(w.shape[1]), int(w.shape[2])
        self.stride, self.has_bwd = stride, bool(want_bwd)
        self.small = bool(_lib.lib().svr_conv2d_small_supported(self.Cout, self.C, self.k))
        if self.small:
            self.w, self.has_bwd = w, True
            return
        nbytes = 256 + _lib.lib().svr_conv2d_planes_bytes(self.Cout, self.C, self.k)
        self.buf = torch.empty(nbytes, device=w.device, dtype=torch.uint8)
        self._amax, self._planes = self.buf.data_ptr(), self.buf.data_ptr() + 256
        check(_lib.lib().svr_conv2d_prepare(_p(w), self.Cout, self.C, self.k, stride, int(self.has_bwd), self.amax_ptr(),
                                            self.planes_ptr(), _stream()), "conv2d_prepare")

    def amax_ptr(self):
        return self._amax

    def planes_ptr(self):
        return self._planes


def conv2d_prepare_many(items):
    """[(weight, stride, want_bwd), ...] -> [Conv2dPlanes, ...]: the planes of all layers of a network in three launches per
    16 layers (svr_conv2d_prepare_many) instead of four per layer; bit-identical to preparing them one by one."""
    l = _lib.lib()
    out, todo = [None] * len(items), []
    for i, (weight, stride, want_bwd) in enumerate(items):
        _f32(weight)
        w = weight.detach().contiguous()
        pl = Conv2dPlanes.__new__(Conv2dPlanes)
        pl.Cout, pl.C, pl.k = int(w.shape[0]), int(w.shape[1]), int(w.shape[2])
        pl.stride, pl.has_bwd = stride, bool(want_bwd)
        pl.small = bool(l.svr_conv2d_small_supported(pl.Cout, pl.C, pl.k))
        out[i] = pl
        if pl.small:
            pl.w, pl.has_bwd = w, True
        else:
            todo.append((pl, w))
    for g0 in range(0, len(todo), 16):
        grp = todo[g0:g0 + 16]
        n = len(grp)
        sizes = [(int(l.svr_conv2d_planes_bytes(pl.Cout, pl.C, pl.k)) + 255) // 256 * 256 for pl, _ in grp]
        buf = torch.empty(256 * n + sum(sizes), device=grp[0][1].device, dtype=torch.uint8)
        base, off = buf.data_ptr(), 256 * n
        ptrs = []
        for i, (pl, _) in enumerate(grp):
            pl.buf, pl._amax, pl._planes = buf, base + 256 * i, base + off
            ptrs.append(base + off)
            off += sizes[i]
        I32A, PA = C.c_int32 * n, C.c_void_p * n
        check(l.svr_conv2d_prepare_many(n, PA(*[w.data_ptr() for _, w in grp]), I32A(*[pl.Cout for pl, _ in grp]),
                                        I32A(*[pl.C for pl, _ in grp]), I32A(*[pl.k for pl, _ in grp]),
                                        I32A(*[pl.stride for pl, _ in grp]), I32A(*[int(pl.has_bwd) for pl, _ in grp]),
                                        C.c_void_p(base), PA(*ptrs), _stream()), "conv2d_prepare_many")
    return out


def _amax_any(t):
    """amax word of a gradient tensor of any element count (amax_of wants multiples of 4 and 16-byte alignment)."""
    if t.numel() % 4 == 0 and t.data_ptr() % 16 == 0:
        return amax_of(t)
    return t.detach().abs().max().reshape(1).view(torch.int32)


def _conv2d_ws(nbytes, device):
    return torch.empty(max(int(nbytes), 256), device=device, dtype=torch.uint8)


def conv2d_virtual(src0, src1, act, up):
    """V (B,Hv,Wv,C0+C1) = [x2 upsample](act(cat(src0, src1))): a decoder block's input, written once."""
    d, (Hv, Wv, _, _) = _conv2d_desc(src0, src1, 3, 1, act, up)
    V = torch.empty(d.B, Hv, Wv, d.C0 + d.C1, device=src0.device, dtype=torch.float32)
    check(_lib.lib().svr_conv2d_virtual(C.byref(d), _p(V), _stream()), "conv2d_virtual")
    return V


def conv2d_fwd(src0, src1, k, stride, act, planes, bias, amax_x=None, want_amax=False):
    """y (B,Ho,Wo,Cout) = conv_k(act(cat(src0, src1))) + bias as one implicit GEMM (no upsample here: conv2d_virtual first).
    amax_x: the input is a gradient operand (scaled f16 split); want_amax: leave max|y| on the result (`_svr_amax`)."""
    d, (_, _, Ho, Wo) = _conv2d_desc(src0, src1, k, stride, act, False)
    if planes.C != d.C0 + d.C1 or planes.k != k or planes.stride != stride:
        raise ValueError("conv2d_fwd: the planes were prepared for another layer")
    _f32(bias)
    y = torch.empty(d.B, Ho, Wo, planes.Cout, device=src0.device, dtype=torch.float32)
    if planes.small:
        check(_lib.lib().svr_conv2d_small_fwd(C.byref(d), _p(planes.w), _p(bias), _p(y), planes.Cout, _stream()), "conv2d_small_fwd")
        return y
    ws = _conv2d_ws(_lib.lib().svr_conv2d_workspace_bytes(C.byref(d), planes.Cout), src0.device)
    amax_y = amax_slot(src0.device) if want_amax else None
    check(_lib.lib().svr_conv2d_fwd(C.byref(d), planes.planes_ptr(), planes.amax_ptr(), _p(bias), _p(y), planes.Cout, _p(amax_x),
                                    _p(amax_y), _p(ws), _stream()), "conv2d_fwd")
    if want_amax:
        y._svr_amax = amax_y
    return y


def linear_bwd_data_splitk(dy, wt):
    """dx (M,K) = dy (M,N) @ wt.T for wt (K,N) contiguous, through the implicit-GEMM kernel's k = 1 mode: the reduction over N is
    split over workgroups and the partial tiles are summed in a fixed order (few output tiles, long reduction).  f16x3s
    arithmetic.  Measured on IF-Net's projected levels' voxel GEMMs (reduction 1 792, 128 outputs): 22 + 26 us against 96 at
    4 096 rows, 77 + 41 against 104 at 32 768 rows -- and no change of the step's median (the GEMMs sit in the shadow of the
    projected scatter), so the step keeps linear_bwd_data; kept as an op for callers with that shape (tools/exp/dbg_splitk.py)."""
    _f32(dy, wt)
    M, N = dy.shape
    K = wt.shape[0]
    planes = Conv2dPlanes(wt.view(K, N, 1, 1), 1, want_bwd=False)
    y = conv2d_fwd(dy.view(1, 1, M, N), None, 1, 1, ACT_NONE, planes, None, amax_x=amax_of(dy), want_amax=True)
    out = y.view(M, K)
    out._svr_amax = y._svr_amax
    return out


def conv2d_bwd_data(src0, src1, k, stride, planes, dy):
    """dIn (B,H,W,C0+C1): gradient of the convolution's activated, concatenated input (scaled f16 split on dy)."""
    d, _ = _conv2d_desc(src0, src1, k, stride, 0, False)
    if not planes.has_bwd:
        raise ValueError("conv2d_bwd_data: the planes were prepared without the backward product")
    _f32(dy)
    dy = dy.contiguous()
    din = torch.empty(d.B, d.H, d.W, d.C0 + d.C1, device=dy.device, dtype=torch.float32)
    if planes.small:
        check(_lib.lib().svr_conv2d_small_bwd_data(C.byref(d), _p(planes.w), _p(dy), planes.Cout, _p(din), _stream()), "conv2d_small_bwd_data")
        return din
    ws = _conv2d_ws(_lib.lib().svr_conv2d_workspace_bytes(C.byref(d), planes.Cout), dy.device)
    check(_lib.lib().svr_conv2d_bwd_data(C.byref(d), planes.planes_ptr(), planes.amax_ptr(), _p(dy), _amax_any(dy).data_ptr(),
                                         planes.Cout, _p(din), _p(ws), _stream()), "conv2d_bwd_data")
    return din


def conv2d_finish_bwd(src0, src1, act, up, dvirt, need0=True, need1=True):
    """dvirt (B,Hv,Wv,C) -> gradients of src0 / src1 (upsample adjoint, activation derivative, channel split)."""
    d, _ = _conv2d_desc(src0, src1, 3, 1, act, up)
    d0 = torch.empty_like(src0) if need0 else None
    d1 = torch.empty_like(src1) if (src1 is not None and need1) else None
    if d0 is None and d1 is None:
        return None, None
    check(_lib.lib().svr_conv2d_finish_bwd(C.byref(d), _p(dvirt), _p(d0), _p(d1), _stream()), "conv2d_finish_bwd")
    return d0, d1


def conv2d_bwd_weight(src0, src1, k, stride, act, dy, Cout, want_bias=True):
    """dW (Cout, C0+C1, k, k) in nn.Conv2d's layout and db from dy (B,Ho,Wo,Cout) and the gathered input."""
    d, _ = _conv2d_desc(src0, src1, k, stride, act, False)
    _f32(dy)
    dy = dy.contiguous()
    dw = torch.empty(Cout, d.C0 + d.C1, k, k, device=dy.device, dtype=torch.float32)
    db = torch.empty(Cout, device=dy.device, dtype=torch.float32) if want_bias else None
    if _lib.lib().svr_conv2d_small_supported(Cout, d.C0 + d.C1, k):
        ws = _conv2d_ws(_lib.lib().svr_conv2d_small_bwd_weight_workspace(C.byref(d), Cout), dy.device)
        check(_lib.lib().svr_conv2d_small_bwd_weight(C.byref(d), _p(dy), Cout, _p(dw), _p(db), _p(ws), _stream()), "conv2d_small_bwd_weight")
        return dw, db
    ws = _conv2d_ws(_lib.lib().svr_conv2d_bwd_weight_workspace(C.byref(d), Cout), dy.device)
    check(_lib.lib().svr_conv2d_bwd_weight(C.byref(d), _p(dy), _amax_any(dy).data_ptr(), Cout, _p(dw), _p(db), _p(ws), _stream()),
          "conv2d_bwd_weight")
    return dw, db
