"""IF-Net on MI355X: host-side mirror of the reference's model/ifnet.py.

Same public surface as the reference (model/ifnet.py:10-61,64-199,202-229): ``IFNet(hidden_dim)``
with ``forward(x (B,1,D,H,W), points (B,N,3)) -> logits (B,N)``, submodule / parameter names of
SURVEY.md App. A.1 (so Lightning checkpoints `ifnet.*` load), feature extractors
``IFNetFeatureExtractor128`` / ``IFNetFeatureExtractor`` and the inference helpers
``make_3d_grid`` / ``evaluate_network_on_grid``.  The one deliberate difference: the architecture
is a constructor argument (``net_res=128``) instead of a value parsed from sys.argv at import
(model/ifnet.py:8).

Every arithmetic step runs in the HIP kernels of libsvr_hip.so (include/svr_hip.h).  The
nn.Conv3d / nn.BatchNorm3d / nn.Conv1d submodules only HOLD the parameters and buffers; their
torch forward is never called.  There is no CPU fallback.
"""
import contextlib
import os

import torch
import torch.nn as nn

from .. import ops
from ..arena import StepArena

_ARCH = {
    # stages: ((conv attr names), bn attr name); channels per stage; gather displacement; align_corners
    128: dict(stages=[(("conv_in",), "conv_in_bn"), (("conv_0", "conv_0_1"), "conv0_1_bn"),
                      (("conv_1", "conv_1_1"), "conv1_1_bn"), (("conv_2", "conv_2_1"), "conv2_1_bn"),
                      (("conv_3", "conv_3_1"), "conv3_1_bn")],
              disp=0.0722, align_corners=False),
    32: dict(stages=[(("conv_1", "conv_1_1"), "conv1_1_bn"), (("conv_2", "conv_2_1"), "conv2_1_bn"),
                     (("conv_3", "conv_3_1"), "conv3_1_bn")],
             disp=0.035, align_corners=True),
}


def _displacements(d):
    rows = [[0.0, 0.0, 0.0]]
    for axis in range(3):
        for sign in (-1, 1):
            r = [0.0, 0.0, 0.0]
            r[axis] = sign * d
            rows.append(r)
    return torch.tensor(rows)


_side_streams = {}


def _get_side_stream(device, which=0):
    """Side stream `which` of a device (0: plans / orders in the forward and the kept-column branch of the backward;
    1: the kept columns' weight gradient)."""
    if os.environ.get("SVR_NO_SIDE_STREAM"):      # measurement switch: everything on the caller's stream
        return torch.cuda.current_stream(device)
    key = (torch.device(device).index, which)
    if key not in _side_streams:
        _side_streams[key] = torch.cuda.Stream(device=device)
    return _side_streams[key]


# Backward scatter form per level (gather.hip):
#   "auto"   (default) levels with C <= 64: atomic-free PULL form when the points are spread out (uniform-like: it is bound
#            by its longest serial walk), float atomics over the joint ITEM order when they are clustered (surface samples:
#            walks of thousands of items).  The statistic is the plan's `longest walk` of the PREVIOUS step, read without
#            synchronisation (pinned copy + event); the first step of a shape uses the item order.  C = 128: item order.
#   "pull"   force the pull form for C <= 64;  "items": item orders everywhere;  "atomic": the round-1 configuration
#            (per-displacement point orders for levels with >= ~0.15 points per voxel).
# Measured at config 3 (ms, levels 1/2/3): uniform points pull 0.56/0.38/0.57, items 1.07/0.99/0.45; points clustered on
# planes pull 1.72/1.81/3.53, items 0.65/0.56/0.49 -- longest walk 13/24/94 against 1409/1669/3282.
SCATTER_FORM = os.environ.get("SVR_SCATTER_FORM", "auto")
# Bit-reproducible training step (SVR_DETERMINISTIC=1, or set ifnet.DETERMINISTIC before the step): every scatter of the
# backward takes an atomic-free form -- pull plans for the levels with C <= 64 whatever the point distribution, the two-pass
# form for the projected 128-channel levels -- so no float atomic is left in the step and two runs give the same bits
# (every other reduction of the step already has a fixed order).  Measured on one box, configs[2], uniform points: 17.0 ms
# against 16.2 ms per step (the atomic forms win inside the backward's fork, and level 3's pull form is bound by its longest
# walk); with surface-clustered points the pull form is several times slower, so it is a mode, not
# the default.  A level that has no atomic-free form (C = 128 without the projection: the 32-architecture) raises.
DETERMINISTIC = os.environ.get("SVR_DETERMINISTIC") is not None
PULL_MAX_WALK = int(os.environ.get("SVR_PULL_MAX_WALK", "64"))
_pull_hint = {}       # (device, level, dims, C, B, N) -> {"use": last decision, "slots": [(pinned int32[2], event), ...]}


PULL_DECISION_LAG = 3   # steps between a plan's statistic and the decision that uses it


def _pull_decision(key, plan, side):
    """Use the pull form?  Decided from the plan statistic of the step PULL_DECISION_LAG steps back -- a FIXED lag, so the
    form a level takes in step s is a function of the data alone and not of how far the host happens to run ahead of the
    GPU (round 2 took "the most recent statistic that has arrived": two runs of the same workload could take different
    forms in the same step).  The statistic travels through a ring of pinned slots (queued behind the plan on the side
    stream); DataParallelTrainer keeps at most 2 steps in flight, so the slot read here has normally arrived long ago --
    if it has not (an unthrottled caller), the host waits for it.  The first steps of a shape use the item order."""
    h = _pull_hint.setdefault(key, {"use": False, "slots": [None] * (PULL_DECISION_LAG + 1), "seq": 0})
    if torch.cuda.is_current_stream_capturing():
        return h["use"]            # inside a HIP-graph capture: no event waits / pinned allocations; the decision of the
        #                            eager warm-up steps is baked into the graph
    seq, ring = h["seq"], h["slots"]
    old = ring[(seq - PULL_DECISION_LAG) % len(ring)] if seq >= PULL_DECISION_LAG else None
    if old is not None:
        old["event"].synchronize()
        h["use"] = 0 < int(old["host"][0]) <= PULL_MAX_WALK
    slot = ring[seq % len(ring)]
    if slot is None:
        slot = ring[seq % len(ring)] = {"host": torch.empty(2, dtype=torch.int32, pin_memory=True), "event": torch.cuda.Event()}
    slot["host"].copy_(plan.stats, non_blocking=True)
    slot["event"].record(side)
    h["seq"] = seq + 1
    return h["use"]


def _level_orders_async(pts, D, H, W, n_levels, align, layout=None, disp=None, proj_levels=(), arena=None):
    """Backward-scatter preparation that depends only on the points, computed on a side stream beside the encoder;
    returns (orders per level, pull plans per level, ready event).  Level l has the pyramid's resolution
    (D, H, W) >> (l - 1) for l >= 1 (level 0 is the input grid, one channel: neither).  See SCATTER_FORM.
    layout: the layout of the gradient rows the scatter will read (the full FeatureLayout, or the KeptLayout of the fused
    step).  arena: the step's StepArena -- the plans' arrays then live there (no allocator traffic on the side stream)."""
    B, N = pts.shape[0], pts.shape[1]
    orders = [None] * n_levels
    plans = [None] * n_levels
    main = torch.cuda.current_stream()
    side = _get_side_stream(pts.device)
    side.wait_stream(main)          # pts may have just been produced on the main stream
    if arena is None or not arena.owns(pts):
        # ... and must not return to the main stream's pool while the side stream reads it.  (Skipped only for points that
        # LIVE in the arena -- the Morton-sorted copy; with spatial_sort off or N == 1 pts is the caller's tensor even
        # under a lease, and a graph dropped without a backward would hand its block out again on main.)
        pts.record_stream(side)
    launched = False
    form = "pull" if DETERMINISTIC else SCATTER_FORM
    fits32 = layout is not None and N > 0 and 7 * B * N < 2 ** 31 and B * N * layout.row_stride < 2 ** 31
    with torch.cuda.stream(side):
        for l in range(1, n_levels):
            dhw = (max(D >> (l - 1), 1), max(H >> (l - 1), 1), max(W >> (l - 1), 1))
            C = layout.channels[l] if layout is not None else 0
            tag = f"L{l}."
            if l in proj_levels:                     # backward-only projection: items by (cell, displacement)
                if DETERMINISTIC or (PROJ_TWO_PASS and min(dhw) >= PROJ_TWO_PASS_MIN_DIM):
                    orders[l] = ops.project_plan(pts, dhw, disp, align)      # two-pass form: no float atomics
                    orders[l].record_stream(main)
                else:
                    orders[l] = ops.item_order(pts, dhw, disp, align, with_j=True, arena=arena, tag=tag)
                    if arena is None:
                        orders[l].record_stream(main)
                launched = True
            elif form in ("auto", "pull") and fits32 and ops.pull_plan_supported(B, N, dhw, C, layout.row_stride):
                plan = ops.pull_plan(pts, dhw, C, layout.col[l], layout.row_stride, disp, align, arena=arena, tag=tag)
                plan.record_stream(main)
                if form == "pull" or _pull_decision((pts.device.index, l, dhw, C, B, N), plan, side):
                    plans[l] = plan
                else:
                    orders[l] = plan.items            # the plan's sorted item ids are this level's item order
                launched = True
            elif DETERMINISTIC and N > 0:
                raise RuntimeError(f"IF-Net HIP path, DETERMINISTIC: level {l} ({C} channels, {dhw}) has no atomic-free scatter "
                                   "(pull plans cover 16 / 32 / 64 channels, the projection the 128-channel levels of the "
                                   "128-architecture)")
            elif form in ("auto", "pull", "items") and fits32:
                orders[l] = ops.item_order(pts, dhw, disp, align, arena=arena, tag=tag)
                if arena is None:
                    orders[l].record_stream(main)
                launched = True
            elif N >= 0.15 * dhw[0] * dhw[1] * dhw[2] and N > 64:
                orders[l] = ops.voxel_order(pts, dhw, align)
                orders[l].record_stream(main)
                launched = True
        ready = None
        if launched:
            ready = torch.cuda.Event()
            ready.record(side)
    return orders, plans, ready


# Backward-only projection of the 128-channel levels (gather.hip, gather_bwd_proj_kernel): their 2 x 896 feature columns
# are 69 % of fc_0's K, and the scatter commutes with fc_0's product -- dh0 rows (256 wide) are scattered into
# dP_l[b][voxel][j][256], and two GEMMs over VOXELS (32 768 rows at level 4) give the level's gradient volume and its slice
# of dW0.  dX0 / dW0 of the point MLP then only cover the remaining 800 columns.  The forward pass is unchanged.
PROJECT_WIDE_LEVELS = os.environ.get("SVR_NO_PROJECTION") is None
# Optional two-pass form of the projected scatter for levels of 16^3 voxels and more (run sums stored, then one gather-form
# pass per dP row: no float atomics, dP written once, bit-reproducible).  At 16^3 x 8 samples there are ~270 000 runs of ~10
# items: 2.2 GB of run-end atomics against 2.2 GB stored + read plainly, 1.53 -> 1.20 ms alone -- but only 18.25 -> 18.18 ms
# per step inside the backward's fork (it is not on the critical stream), a 2.7 GB buffer and a longer allocator warm-up, so
# the atomic form stays the default.  SVR_PROJ_TWO_PASS=1 selects it (at 8^3 the atomic form wins anyway: 0.49 vs 0.55 ms).
PROJ_TWO_PASS = os.environ.get("SVR_PROJ_TWO_PASS") is not None
PROJ_TWO_PASS_MIN_DIM = int(os.environ.get("SVR_PROJ_TWO_PASS_MIN_DIM", "16"))
# Fused gather -> fc_0 forward (gather_fc0.hip): the feature rows are never written to HBM; only the columns a backward
# still needs (the levels that are not projected) are kept.  SVR_NO_FUSED_FC0=1 restores the two separate kernels.
FUSE_FC0 = os.environ.get("SVR_NO_FUSED_FC0") is None
# ... and its bf16-storage variant for the query path's throughput mode (SVR_NO_FUSED_FC0_BF16=1: gather + fc_0 as two kernels)
FUSE_FC0_BF16 = os.environ.get("SVR_NO_FUSED_FC0_BF16") is None
# Fused + projected backward: the kept-column branch on the side stream beside the projected branch (SVR_NO_BWD_OVERLAP=1: serial)
PREPARE_FORWARD_FIRST = os.environ.get("SVR_PREP_FWD_FIRST", "1") != "0"   # "0": one event behind all weight planes (A/B)
OVERLAP_BACKWARD = os.environ.get("SVR_NO_BWD_OVERLAP") is None
# First stage of the 128-architecture (conv_in -> ReLU -> BatchNorm -> pool) with conv_in's activation recomputed instead
# of stored (stage1.hip): 3 of 5 forward and 5 of 7 backward passes over 1 GB tensors less.  SVR_NO_STAGE1=1: the separate
# conv / BatchNorm / weight-gradient kernels.
STAGE1_RECOMPUTE = os.environ.get("SVR_NO_STAGE1") is None
# Training step: the Morton sort of the points runs on the side stream beside the encoder (SVR_SORT_ON_MAIN=1: in front of it)
SORT_ON_SIDE_STREAM = os.environ.get("SVR_SORT_ON_MAIN") is None
# ... and so do the parameter-only preparations of the split-precision layers (SVR_NO_WEIGHT_PREP=1: in front of every layer)
PREPARE_WEIGHTS_AHEAD = os.environ.get("SVR_NO_WEIGHT_PREP") is None
# Level of the kept branch whose scatter runs on the third stream of the backward's fork (SVR_FORK_SPLIT_LEVEL=0: none)
FORK_SPLIT_LEVEL = int(os.environ.get("SVR_FORK_SPLIT_LEVEL", "3"))
# Per-level events instead of one join in front of the encoder's backward (SVR_NO_FORK_PIPELINE=1: the single join)
FORK_PIPELINED = os.environ.get("SVR_NO_FORK_PIPELINE") is None
# BatchNorm statistics of stages 2..5 from the epilogue of the stage's last convolution (SVR_NO_CONV_STATS=1: a separate pass)
STATS_IN_CONV_EPILOGUE = os.environ.get("SVR_NO_CONV_STATS") is None
# The step's large cross-stream buffers live in a per-module StepArena (arena.py); SVR_NO_ARENA=1: ordinary allocations
USE_ARENA = os.environ.get("SVR_NO_ARENA") is None


def _fc0_fusable(channels, B, dims, n_out):
    """Shapes gather_fc0.hip is built for (svr_gather_fc0_supported): 256 outputs, channel counts 1 (once), 16, 32 or
    multiples of 64, every level's volume below 2^30 elements (32-bit byte offsets)."""
    if n_out != 256 or sum(1 for c in channels if c == 1) > 1:
        return False
    D, H, W = dims
    for l, c in enumerate(channels):
        if not (c in (1, 16, 32) or c % 64 == 0):
            return False
        s = max(l - 1, 0)
        if B * max(D >> s, 1) * max(H >> s, 1) * max(W >> s, 1) * c >= 2 ** 30:
            return False
    return True


def _stage1_applies(si, convs, inp):
    """First stage, one convolution from one input channel to 16 (the 128-architecture's conv_in), shapes stage1.hip covers."""
    return (STAGE1_RECOMPUTE and si == 0 and len(convs) == 1 and inp.is_cuda and tuple(convs[0].weight.shape[:2]) == (16, 1)
            and min(inp.shape[1:4]) >= 1 and ops.stage1_supported(inp, 16))


class _ProjLink:
    """Hand-over between the point MLP's backward (which runs first and produces dh0) and the encoder's."""

    def __init__(self, levels, layout):
        self.levels = tuple(levels)
        self.layout = layout
        self.dh0 = None
        self.need_level0 = False
        cols = sorted((layout.col[l], layout.col[l] + 7 * layout.channels[l]) for l in levels)
        keep, start = [], 0
        for a, b in cols:                              # complement of the projected column ranges inside the row
            if a > start:
                keep.append((start, a))
            start = b
        if start < layout.row_stride:
            keep.append((start, layout.row_stride))
        # the raw-grid level (C == 1) gets a segment of its own, so that "its gradient is only needed for d(loss)/d(input)"
        # can never drop the columns of another kept level that happens to share a segment with it
        raw = [(layout.col[l], layout.col[l] + 7) for l, c in enumerate(layout.channels) if c == 1 and l not in levels]
        split = []
        for a, b in keep:
            cuts = sorted({a, b} | {c for r in raw for c in r if a < c < b})
            split += list(zip(cuts[:-1], cuts[1:]))
        self.keep = split                               # column segments dX0 / dW0 still have to cover
        self.raw = set(raw)                             # ... those of them that hold nothing but the raw-grid level


class _EncoderGatherFn(torch.autograd.Function):
    """x, points, encoder parameters -> feature rows (B*N, FS) in the internal column layout.

    forward  = reference model/ifnet.py:155-198 (conv/ReLU/BN/pool pyramid + 6 grid_samples + cat)
    backward = autograd of the same, hand-scheduled (gather scatter -> BN -> ReLU -> conv per stage).
    """

    @staticmethod
    def forward(ctx, ext, grad_mode, link, lease, w0p, b0, x, points, *params):
        """b0 given (with link and w0p): FUSED form -- the output is h0 = relu(fc_0(rows)) straight from the gather
        (gather_fc0.hip), the rows of the levels that are not projected are kept for the backward, and backward()
        takes the gradient wrt fc_0's PRE-activation (what _PointMLPFn's headless form returns for its input)."""
        B = x.shape[0]
        D, H, W = x.shape[2:]
        training = ext.training
        x_cl = x.contiguous().view(B, D, H, W, 1)
        pts = points.contiguous()
        levels = [x_cl]
        saved = []
        inp = x_cl
        nst = len(ext._stages)
        # The per-level visiting orders of the backward scatter depend only on the points: their radix sorts (dozens
        # of ~6 us launches) run on a side stream beside the encoder instead of in front of the scatter.
        ctx.level_orders, ctx.level_plans, ctx.orders_ready = None, None, None
        # only when a backward can follow: grad mode of the CALLER (it is always off inside Function.forward, and
        # needs_input_grad ignores no_grad) and something that requires grad
        will_backward = bool(grad_mode) and any(ctx.needs_input_grad)
        ctx.link = link if (will_backward and link is not None) else None
        if ctx.link is not None:
            ctx.link.need_level0 = bool(ctx.needs_input_grad[6])
        # the step arena (large cross-stream buffers, allocated once): held from here to the end of backward()
        if lease is not None and not (will_backward and x.is_cuda):
            lease.release()
            lease = None
        ctx.lease = lease
        arena = lease.arena if lease is not None else None
        ctx.fused = b0 is not None
        nlev = nst + 1
        keep = []
        if ctx.fused and will_backward:
            keep = [l for l in range(nlev) if ctx.link is None or l not in ctx.link.levels]
        # fused step: the backward reads the COMPACT kept-column matrix (800 columns at the 128-architecture)
        ctx.klay = ext._layout.subset(keep) if keep else None
        if will_backward and x.is_cuda:
            ctx.level_orders, ctx.level_plans, ctx.orders_ready = _level_orders_async(
                pts, D, H, W, nlev, ext._align, ctx.klay if ctx.klay is not None else ext._layout, ext._disp,
                proj_levels=ctx.link.levels if ctx.link is not None else (), arena=arena)
        for si, (convs, bn) in enumerate(ext._stages):
            if _stage1_applies(si, convs, inp):
                conv = convs[0]
                y, pooled, argmax, ss, mean, wp = ops.stage1_fwd(
                    inp, conv.weight.detach(), conv.bias.detach(), bn.weight.detach(), bn.bias.detach(), bn.running_mean,
                    bn.running_var, training, eps=bn.eps, momentum=bn.momentum, want_pool=(si + 1 < nst))
                levels.append(y)
                saved.append((inp, ("stage1", wp), argmax, ss, mean))      # conv_in's activation is never stored
                inp = pooled
                continue
            acts = []
            cur = inp
            stats = None
            for conv in convs:
                if training and len(convs) == 1 and conv.weight.shape[1] == 1 and conv.weight.shape[0] in (16, 32):
                    # conv_in: the statistics of the BatchNorm that follows come out of the conv kernel's epilogue
                    cur, stats = ops.conv3d_c1_fwd_stats(cur, conv.weight.detach(), conv.bias.detach(), relu=True)
                elif training and conv is convs[-1] and STATS_IN_CONV_EPILOGUE:
                    # the stage's last convolution: its epilogue also leaves the BatchNorm's partial sums (no statistics pass)
                    cur, stats = ops.conv3d_k3_fwd(cur, conv.weight.detach(), conv.bias.detach(), relu=True, want_stats=True)
                else:
                    cur = ops.conv3d_k3_fwd(cur, conv.weight.detach(), conv.bias.detach(), relu=True)
                acts.append(cur)
            y, pooled, argmax, ss, mean = ops.bn_forward(
                cur, bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, training,
                eps=bn.eps, momentum=bn.momentum, want_pool=(si + 1 < nst), stats=stats)
            levels.append(y)
            saved.append((inp, acts, argmax, ss, mean))
            inp = pooled
        if training:      # one multi-tensor launch for the counters of all stages
            torch._foreach_add_([bn.num_batches_tracked for _, bn in ext._stages], 1)
        ready, ext._points_ready = getattr(ext, "_points_ready", None), None
        if ready is not None:
            torch.cuda.current_stream().wait_event(ready)      # the sorted points (IFNet.forward sorts them on the side stream)
        if ctx.fused:
            rows_out = None
            if keep and arena is not None:
                rows_out = arena.get("kept_rows", (B * pts.shape[1], ctx.klay.row_stride), torch.float32, x.device)
            out, feat = ops.gather_fc0_fwd(levels, pts, ext._layout, ext._disp, ext._align, w0p.detach(), b0.detach(), relu=True,
                                           keep_levels=keep, keep_layout=ctx.klay, rows_out=rows_out, arena=arena)
            ctx.feat = feat
        else:
            out = feat = ops.gather_fwd(levels, pts, ext._layout, ext._disp, ext._align)
        # the zero-initialised gradient volumes of the backward scatter (1.45 GB of memset at config 3) are prepared on
        # the side stream too, beside the point MLP, instead of in front of the scatter
        ctx.gvols = None
        if will_backward and x.is_cuda:
            main, side = torch.cuda.current_stream(), _get_side_stream(x.device)
            with torch.cuda.stream(side):
                # (levels with a pull plan are written by plain stores: no zero fill)
                proj = ctx.link.levels if ctx.link is not None else ()
                ctx.gvols = [None] + [None if l in proj else _gvol(arena, l, v, zero=ctx.level_plans[l] is None)
                                      for l, v in enumerate(levels) if l >= 1]
                # ... and the zeroed (voxel, displacement, 256) slabs the projected levels' atomic scatter accumulates into
                ctx.dP = {}
                if arena is not None:
                    for l in proj:
                        if not isinstance(ctx.level_orders[l], ops.ProjPlan):      # (the two-pass form overwrites its own)
                            v = levels[l]
                            ctx.dP[l] = arena.get(f"dP{l}", (B, v.shape[1] * v.shape[2] * v.shape[3], 7, 256), torch.float32,
                                                  x.device).zero_()
                if arena is None:
                    for g in ctx.gvols[1:]:
                        if g is not None:
                            g.record_stream(main)
                ctx.orders_ready = torch.cuda.Event()
                ctx.orders_ready.record(side)
        ctx.ext, ctx.saved, ctx.levels, ctx.pts = ext, saved, levels, pts
        ctx.w0p = w0p.detach() if (ctx.link is not None or ctx.fused) else None
        ctx.x_shape = x.shape
        ctx.training = training
        return out

    @staticmethod
    def backward(ctx, gfeat):
        if ctx.saved is None:
            raise RuntimeError("IF-Net HIP path: second backward through the same forward pass (the saved pyramid and "
                               "activations are released by the first one; retain_graph is not supported)")
        ext, saved, levels, pts = ctx.ext, ctx.saved, ctx.levels, ctx.pts
        # plain attributes are not freed with the graph's saved tensors: a caller that keeps the loss tensor (and with it
        # this node) alive into the next step would otherwise hold ~9 GB of activations twice
        ctx.saved = ctx.levels = ctx.pts = None
        need_x, need_pts = ctx.needs_input_grad[6], ctx.needs_input_grad[7]
        gfeat = gfeat.contiguous()
        link = ctx.link
        proj = link.levels if link is not None else ()
        lease, ctx.lease = ctx.lease, None
        arena = lease.arena if lease is not None else None
        klay = ctx.klay if ctx.fused else None
        scatter_layout = klay if klay is not None else ext._layout
        dw0_keep = db0 = None
        level_orders, level_plans = ctx.level_orders, ctx.level_plans
        if level_orders is None:
            level_orders, level_plans, ready = _level_orders_async(pts, *levels[0].shape[1:4], len(levels), ext._align,
                                                                   scatter_layout, ext._disp, proj_levels=proj, arena=arena)
        else:
            ready = ctx.orders_ready
        if ops.GATHER_FLAGS & ops._lib.GATHER_DETERMINISTIC:      # the serial test scatter accumulates into zeros
            level_plans = [None] * len(levels)
            ctx.gvols = None
        if ctx.gvols is not None:
            gvols = [torch.zeros_like(levels[0]) if need_x else None] + ctx.gvols[1:]
            ctx.gvols = None
        else:
            gvols = [torch.zeros_like(levels[0]) if need_x else None] + \
                    [None if l in proj else _gvol(arena, l, v, zero=level_plans[l] is None)
                     for l, v in enumerate(levels) if l >= 1]
        main = torch.cuda.current_stream() if gfeat.is_cuda else None
        # Fused forward + projection: the backward of the KEPT columns (dW0 / dX0 over 800 columns, then the scatter of
        # levels 1-3) and the backward of the PROJECTED levels (row scatter + voxel GEMMs) only share dh0 as an input;
        # neither saturates the chip (GEMMs at 2-3x their memory floor, scatters latency / atomics bound), so the kept
        # branch runs on the side stream beside the projected one and the two join in front of the encoder's backward.
        fork = (ctx.fused and link is not None and main is not None and OVERLAP_BACKWARD
                and not torch.cuda.is_current_stream_capturing())      # (a captured step keeps the serial order)
        keep_stream = _get_side_stream(gfeat.device) if fork else main
        skip = ()
        if ctx.fused:
            # fused forward: fc_0's backward lives here.  gfeat is dz0 (B*N, 256); dW0 / dX0 only over the kept columns,
            # ONE product each over the compact kept-column matrix
            dh0, feat, w0p = gfeat, ctx.feat, ctx.w0p
            ctx.feat = None
            cols = klay.full_cols_on(w0p.device)
            w0k = w0p[:, cols]                           # (256, 800): fc_0's weights of the kept columns (zeros behind the padding)
            dw0_keep = torch.zeros_like(w0p)
            if fork:
                # (no record_stream on what the side streams read: dh0, the kept rows, the gradient volumes ... stay
                # referenced until main has joined both streams -- or live in the step arena -- so the allocator cannot
                # hand them out earlier)
                keep_stream.wait_stream(main)
            # dW0 over the kept columns is a leaf (needed at the return only): third stream
            w_stream = _get_side_stream(gfeat.device, 1) if fork else main
            if fork:
                w_stream.wait_stream(main)
            with torch.cuda.stream(w_stream) if fork else contextlib.nullcontext():
                dwk, db0 = ops.linear_bwd_weight(dh0, feat, want_bias=True)
                dw0_keep[:, cols[:klay.width]] = dwk[:, :klay.width]
                if fork:
                    db0.record_stream(main)
            with torch.cuda.stream(keep_stream) if fork else contextlib.nullcontext():
                if arena is not None:
                    gfeat = arena.get("kept_grad", tuple(feat.shape), torch.float32, feat.device)
                else:
                    gfeat = torch.empty_like(feat)
                ops.linear_bwd_data(dh0, w0k, out=gfeat)
                if fork:
                    dx0_done = torch.cuda.Event()
                    dx0_done.record(keep_stream)
            skip = tuple(l for l in range(len(levels)) if klay.col[l] < 0)
            if link is not None:
                link.dh0 = dh0
        if link is not None and link.dh0 is None:
            raise RuntimeError("IF-Net HIP path: the projected backward needs dh0 from the point MLP's backward")
        # the kept branch is the longest chain of the fork (dX0 -> scatter of levels 2, 1, 3 one after the other): its last
        # level moves to the third stream, behind dW0, as soon as dX0 is there (FORK_SPLIT_LEVEL; arena buffers only)
        split = FORK_SPLIT_LEVEL if (fork and arena is not None and ctx.fused and not need_pts and FORK_SPLIT_LEVEL not in skip
                                     and FORK_SPLIT_LEVEL not in proj and 0 < FORK_SPLIT_LEVEL < len(levels)) else None
        lo = [None if l in proj else o for l, o in enumerate(level_orders)]
        # PIPELINED join (FORK_PIPELINED): instead of joining every branch in front of the encoder's backward, every level's
        # gradient volume gets its own event and stage s only waits for level s + 1 -- the deep stages' backward (1.3 ms of
        # small, low-occupancy kernels) runs while the side streams still scatter the fine levels; the kept levels are
        # scattered one call per level, coarse to fine (the order the stages need them), and the coarsest projected level
        # gets a stream of its own beside the other one's scatter on the main stream.
        pipelined = FORK_PIPELINED and split is not None
        level_done = {}

        def scatter_level(l):
            ops.gather_bwd(levels, gvols, pts, gfeat, scatter_layout, ext._disp, ext._align, want_gpoints=False,
                           level_orders=lo, level_plans=level_plans, skip_levels=tuple(k for k in range(len(levels)) if k != l))
            level_done[l] = torch.cuda.Event()
            level_done[l].record(torch.cuda.current_stream())

        with torch.cuda.stream(keep_stream) if fork else contextlib.nullcontext():
            if ready is not None:
                torch.cuda.current_stream().wait_event(ready)
            if pipelined:
                gpts = None
                for l in range(len(levels) - 1, -1, -1):      # (level 0, the raw grid, only when d(loss)/d(input) is wanted)
                    if l != split and l not in skip and l not in proj and gvols[l] is not None:
                        scatter_level(l)
            else:
                gpts = ops.gather_bwd(levels, gvols, pts, gfeat, scatter_layout, ext._disp, ext._align, want_gpoints=need_pts,
                                      level_orders=lo, level_plans=level_plans,
                                      skip_levels=skip if split is None else tuple(skip) + (split,))
            if fork:
                keep_done = torch.cuda.Event()
                keep_done.record(keep_stream)
        if split is not None:
            with torch.cuda.stream(w_stream):
                w_stream.wait_event(dx0_done)
                if ready is not None:
                    w_stream.wait_event(ready)
                scatter_level(split)
        dw0p = None
        p_stream = None
        dPs, ctx.dP = (getattr(ctx, "dP", None) or {}), None
        if proj:
            # projected levels: dP = scatter of the dh0 rows, then two GEMMs over voxels (see gather_bwd_proj_kernel)
            lay, w0p, dh0 = ext._layout, ctx.w0p, link.dh0
            link.dh0 = None
            if fork and ready is not None:
                main.wait_event(ready)
            dw0p = dw0_keep if dw0_keep is not None else torch.zeros_like(w0p)

            def project_level(l):
                v = levels[l]
                B_, Dl, Hl, Wl, Cl = v.shape
                c0 = lay.col[l]
                dP = ops.gather_project_bwd(pts, dh0, (Dl, Hl, Wl), level_orders[l], ext._disp, ext._align, out=dPs.get(l))
                dP2 = dP.view(B_ * Dl * Hl * Wl, 7 * 256)
                wl = w0p[:, c0:c0 + 7 * Cl].reshape(256, 7, Cl).permute(1, 0, 2).reshape(7 * 256, Cl).contiguous()   # rows (j, n)
                gvols[l] = ops.linear_bwd_data(dP2, wl).view(v.shape)
                dwl, _ = ops.linear_bwd_weight(dP2, v.view(-1, Cl), want_bias=False)                                # (7*256, Cl)
                dw0p[:, c0:c0 + 7 * Cl] = dwl.view(7, 256, Cl).permute(1, 0, 2).reshape(256, 7 * Cl)

            own = max(proj) if (pipelined and len(proj) > 1) else None     # the coarsest projected level: its own stream
            if own is not None:
                p_stream = _get_side_stream(gfeat.device, 2)
                p_stream.wait_stream(main)
                with torch.cuda.stream(p_stream):
                    if ready is not None:
                        p_stream.wait_event(ready)
                    project_level(own)
                    gvols[own].record_stream(main)       # allocated in this stream's pool, read by the stage's backward on main
                    level_done[own] = torch.cuda.Event()
                    level_done[own].record(p_stream)
            for l in proj:
                if l != own:
                    project_level(l)
        if fork and not pipelined:
            main.wait_event(keep_done)
            main.wait_stream(w_stream)
            feat = None      # (only now: see the note at the fork)
        grads = {}
        dpooled = None
        gx = None
        for si in range(len(ext._stages) - 1, -1, -1):
            convs, bn = ext._stages[si]
            inp, acts, argmax, ss, mean = saved[si]
            if pipelined and (si + 1) in level_done:
                main.wait_event(level_done[si + 1])        # this stage's gradient volume, from whichever stream scattered it
            if isinstance(acts, tuple):       # ("stage1", wp): BatchNorm backward + conv_in's weight gradient in two passes
                conv = convs[0]
                dgamma, dbeta, dwp, db, dout = ops.stage1_bwd(
                    inp, acts[1], conv.bias.detach(), gvols[si + 1], dpooled, argmax if dpooled is not None else None, mean, ss,
                    relu_mask=True, training=ctx.training, want_dout=need_x)
                grads[bn.weight], grads[bn.bias] = dgamma, dbeta
                grads[conv.weight], grads[conv.bias] = dwp, db
                if need_x:
                    gx = ops.conv3d_k3_bwd_data(dout, conv.weight.detach())
                continue
            dout, dgamma, dbeta = ops.bn_backward(acts[-1], gvols[si + 1], dpooled, argmax if dpooled is not None else None,
                                                  mean, ss, relu_mask=True, training=ctx.training)
            grads[bn.weight], grads[bn.bias] = dgamma, dbeta
            for k in range(len(convs) - 1, -1, -1):
                conv = convs[k]
                cin = acts[k - 1] if k > 0 else inp
                Co, Ci = conv.weight.shape[0], conv.weight.shape[1]
                grads[conv.weight], grads[conv.bias] = ops.conv3d_k3_bwd_weight(cin, dout, param_layout=True)
                if k > 0:
                    dout = ops.conv3d_k3_bwd_data(dout, conv.weight.detach(), mask=acts[k - 1])
                elif si > 0:
                    dpooled = ops.conv3d_k3_bwd_data(dout, conv.weight.detach())
                elif need_x:
                    gx = ops.conv3d_k3_bwd_data(dout, conv.weight.detach())
        if need_x:
            if pipelined and 0 in level_done:
                main.wait_event(level_done[0])
            gx = (gx + gvols[0]).view(ctx.x_shape)
        if dw0p is None:
            dw0p = dw0_keep
        if pipelined:      # the leaves of the side streams (dW0's slices, db0) and everything that reads the arena
            main.wait_event(keep_done)
            main.wait_stream(w_stream)
            if p_stream is not None:
                main.wait_stream(p_stream)
            feat = None
        if lease is not None:
            lease.release()     # every kernel that touches the arena is enqueued; the next step orders itself behind them
            #                     (also expires the prepared weight planes: the optimizer is about to change the parameters)
        out = [None, None, None, None, dw0p, db0, gx, gpts]
        for p in ext._param_list:
            out.append(grads.get(p))
        return tuple(out)


def _gvol(arena, level, like, zero):
    """Gradient volume of a level for the backward scatter: from the step arena when there is one."""
    if arena is None:
        return torch.zeros_like(like) if zero else torch.empty_like(like)
    g = arena.get(f"gvol{level}", tuple(like.shape), like.dtype, like.device)
    return g.zero_() if zero else g


class _PointMLPFn(torch.autograd.Function):
    """feature rows (B*N, FS) -> logits (B*N): fc_0, fc_1, fc_2 with ReLU, fc_out
    (reference model/ifnet.py:55-59) on the f32 matrix cores."""

    @staticmethod
    def forward(ctx, feat, row_map, w0p, b0, w1, b1, w2, b2, wo, bo, link=None):
        """row_map (int32, or None): feature row m belongs to caller point row_map[m]; the logits are
        scattered back to the caller's order by the fc_out kernel.
        w0p None: HEADLESS form -- `feat` is already h0 = relu(fc_0(rows)) (the fused gather produced it) and the
        gradient returned for it is the one wrt fc_0's pre-activation (h0's ReLU mask applied), the contract of
        _EncoderGatherFn's fused backward."""
        ctx.headless = w0p is None
        if ctx.headless:
            h0 = feat
            h1 = ops.linear_fwd(h0, w1, b1, relu=True)
            h2 = ops.linear_fwd(h1, w2, b2, relu=True)
            logits = ops.fc_out_fwd(h2, wo, bo, row_map)
            ctx.save_for_backward(w1, w2, wo, h0, h1, h2)
            ctx.row_map = row_map
            return logits
        h0 = ops.linear_fwd(feat, w0p, b0, relu=True)
        h1 = ops.linear_fwd(h0, w1, b1, relu=True)
        h2 = ops.linear_fwd(h1, w2, b2, relu=True)
        logits = ops.fc_out_fwd(h2, wo, bo, row_map)
        ctx.save_for_backward(feat, w0p, w1, w2, wo, h0, h1, h2)
        ctx.row_map = row_map
        ctx.link = link
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        if ctx.headless:
            w1, w2, wo, h0, h1, h2 = ctx.saved_tensors
            dh2, dwo, dbo = ops.fc_out_bwd(h2, wo, dlogits.contiguous(), ctx.row_map)
            dw2, db2 = ops.linear_bwd_weight(dh2, h1)
            dh1 = ops.linear_bwd_data(dh2, w2, mask=h1)
            dw1, db1 = ops.linear_bwd_weight(dh1, h0)
            dz0 = ops.linear_bwd_data(dh1, w1, mask=h0)
            return dz0, None, None, None, dw1, db1, dw2, db2, dwo, dbo, None
        feat, w0p, w1, w2, wo, h0, h1, h2 = ctx.saved_tensors
        dh2, dwo, dbo = ops.fc_out_bwd(h2, wo, dlogits.contiguous(), ctx.row_map)   # dh2 already masked by h2 > 0
        dw2, db2 = ops.linear_bwd_weight(dh2, h1)
        dh1 = ops.linear_bwd_data(dh2, w2, mask=h1)
        dw1, db1 = ops.linear_bwd_weight(dh1, h0)
        dh0 = ops.linear_bwd_data(dh1, w1, mask=h0)
        link = ctx.link
        if link is None:
            dw0, db0 = ops.linear_bwd_weight(dh0, feat)
            dfeat = ops.linear_bwd_data(dh0, w0p) if ctx.needs_input_grad[0] else None
            return dfeat, None, dw0, db0, dw1, db1, dw2, db2, dwo, dbo, None
        # projected wide levels: dX0 / dW0 only over the columns that stay (K 2592 -> 800); the encoder's backward gets dh0
        link.dh0 = dh0
        dw0 = torch.zeros_like(w0p)
        dfeat = torch.empty_like(feat)            # the projected levels' columns are never read
        db0 = None
        for a, b in link.keep:
            dws, dbs = ops.linear_bwd_weight(dh0, feat[:, a:b], want_bias=db0 is None)
            dw0[:, a:b] = dws
            db0 = dbs if db0 is None else db0
            if (a, b) in link.raw and not link.need_level0:
                continue                          # raw-grid columns: their gradient is only needed for d(loss)/d(input)
            ops.linear_bwd_data(dh0, w0p[:, a:b], out=dfeat[:, a:b])
        return dfeat, None, dw0, db0, dw1, db1, dw2, db2, dwo, dbo, None


class _PermuteColumnsFn(torch.autograd.Function):
    """w (H, F) -> w[:, src] with zeros where mask (H, FS): a column permutation with zero padding columns.  Backward is the inverse
    permutation (one index_select) -- autograd's own backward of advanced indexing is an index_put with a sort (a dozen
    launches at the end of every step)."""

    @staticmethod
    def forward(ctx, w, src, mask, inv):
        ctx.save_for_backward(inv)
        return w.index_select(1, src).masked_fill_(mask, 0.0)

    @staticmethod
    def backward(ctx, g):
        (inv,) = ctx.saved_tensors
        return g.index_select(1, inv), None, None, None


class _ExtractorBase(nn.Module):
    """Shared host logic of the two extractor variants."""

    # Per-process scratch state that is neither parameter nor buffer: the step arena (multi-GB device buffers, allocated by
    # the first training steps and kept for the module's lifetime -- release_step_buffers() frees them) and the prepared
    # weight planes (hold a torch.cuda.Event, which can be neither pickled nor deep-copied).  copy.deepcopy(model),
    # pickle / torch.save(model) get fresh, empty ones, like the reference nn.Module that has neither.
    _SCRATCH = ("_arena", "_prepared", "_points_ready")

    def __getstate__(self):
        st = dict(self.__dict__)
        for k in self._SCRATCH:
            st.pop(k, None)
        return st

    def __setstate__(self, st):
        self.__dict__.update(st)
        self._arena = StepArena()
        self._prepared = ops.PreparedWeights()
        self._points_ready = None

    def __deepcopy__(self, memo):
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        new.__setstate__(copy.deepcopy(self.__getstate__(), memo))
        # (_stages / _param_list hold references to submodules / parameters: copied through the same memo, so they point at
        # the copy's own conv / bn modules)
        return new

    def release_step_buffers(self):
        """Hand the step arena's device memory (4.4 GB at configs[2]) and the prepared weight planes back: call between
        training and a memory-hungry inference phase.  The next training step allocates them again (a few hipMalloc)."""
        self._arena.release()
        self._prepared = ops.PreparedWeights()

    def _finish_init(self, net_res):
        a = _ARCH[net_res]
        self._stages = [([getattr(self, c) for c in convs], getattr(self, bn)) for convs, bn in a["stages"]]
        self._disp = float(torch.tensor(a["disp"], dtype=torch.float32))
        self._align = a["align_corners"]
        self.displacments = _displacements(a["disp"])            # plain attribute, like the reference
        chans = [1] + [convs[-1].out_channels for convs, _ in self._stages]
        self._layout = ops.FeatureLayout(chans)
        self._arena = StepArena()      # the step's large cross-stream buffers, allocated once (arena.py)
        self._prepared = ops.PreparedWeights()   # split-precision weight planes of a step, prepared on the side stream
        self._param_list = []
        for convs, bn in self._stages:
            for c in convs:
                self._param_list += [c.weight, c.bias]
            self._param_list += [bn.weight, bn.bias]

    @torch.no_grad()
    def encode_levels(self, x):
        """Inference only: the sampled pyramid [x, bn_1, ...] (channels-last), computed ONCE so dense-grid
        evaluation does not redo the encoder per chunk (the reference recomputes it for every chunk,
        model/ifnet.py:220-226).  Uses the module's current mode for BatchNorm (eval -> running stats)."""
        if not x.is_cuda:
            raise RuntimeError("IF-Net HIP path needs GPU tensors (no CPU fallback)")
        B = x.shape[0]
        D, H, W = x.shape[2:]
        inp = x.float().contiguous().view(B, D, H, W, 1)
        levels = [inp]
        nst = len(self._stages)
        for si, (convs, bn) in enumerate(self._stages):
            if _stage1_applies(si, convs, inp):
                conv = convs[0]
                y, pooled = ops.stage1_fwd(inp, conv.weight.detach(), conv.bias.detach(), bn.weight.detach(), bn.bias.detach(),
                                           bn.running_mean, bn.running_var, self.training, eps=bn.eps, momentum=bn.momentum,
                                           want_pool=(si + 1 < nst))[:2]
                levels.append(y)
                inp = pooled
                continue
            cur = inp
            for conv in convs:
                cur = ops.conv3d_k3_fwd(cur, conv.weight.detach(), conv.bias.detach(), relu=True)
            y, pooled, _, _, _ = ops.bn_forward(cur, bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var,
                                                self.training, eps=bn.eps, momentum=bn.momentum, want_pool=(si + 1 < nst))
            levels.append(y)
            inp = pooled
        return levels

    @torch.no_grad()
    def feature_rows_from_levels(self, levels, points, order=None):
        return ops.gather_fwd(levels, points.float().contiguous(), self._layout, self._disp, self._align, order=order)

    def feature_rows(self, x, points, link=None, w0p=None, b0=None, lease=None):
        """(B*N, FS) rows in the internal column layout (what the point MLP consumes).  link / w0p: the backward-only
        projection of the wide levels (IFNet.forward wires it; the encoder's backward then also returns its share of
        fc_0's weight gradient)."""
        if not x.is_cuda:
            raise RuntimeError("IF-Net HIP path needs GPU tensors (no CPU fallback)")
        return _EncoderGatherFn.apply(self, torch.is_grad_enabled(), link, lease, w0p, b0, x.float(), points.float(), *self._param_list)

    def forward(self, x, points):
        """Reference layout (B, sumC, 1, 7, N) -- model/ifnet.py:197; used by API-compat callers."""
        B, N = points.shape[0], points.shape[1]
        rows = self.feature_rows(x, points)
        perm = self._layout.reference_permutation().to(rows.device)
        inv = torch.empty(self._layout.width, dtype=torch.long, device=rows.device)
        valid = perm >= 0
        inv[perm[valid]] = torch.nonzero(valid).squeeze(1)
        f = rows[:, inv].view(B, N, -1, 7).permute(0, 2, 3, 1)
        return f.unsqueeze(2)


class IFNetFeatureExtractor128(_ExtractorBase):
    def __init__(self):
        super().__init__()
        self.conv_in = nn.Conv3d(1, 16, 3, padding=1)
        self.conv_0 = nn.Conv3d(16, 32, 3, padding=1)
        self.conv_0_1 = nn.Conv3d(32, 32, 3, padding=1)
        self.conv_1 = nn.Conv3d(32, 64, 3, padding=1)
        self.conv_1_1 = nn.Conv3d(64, 64, 3, padding=1)
        self.conv_2 = nn.Conv3d(64, 128, 3, padding=1)
        self.conv_2_1 = nn.Conv3d(128, 128, 3, padding=1)
        self.conv_3 = nn.Conv3d(128, 128, 3, padding=1)
        self.conv_3_1 = nn.Conv3d(128, 128, 3, padding=1)
        self.actvn = nn.ReLU()
        self.maxpool = nn.MaxPool3d(2)
        self.conv_in_bn = nn.BatchNorm3d(16)
        self.conv0_1_bn = nn.BatchNorm3d(32)
        self.conv1_1_bn = nn.BatchNorm3d(64)
        self.conv2_1_bn = nn.BatchNorm3d(128)
        self.conv3_1_bn = nn.BatchNorm3d(128)
        self._finish_init(128)


class IFNetFeatureExtractor(_ExtractorBase):
    def __init__(self, f1, f2, f3, f4):
        super().__init__()
        self.conv_1 = nn.Conv3d(1, f1, 3, padding=1)
        self.conv_1_1 = nn.Conv3d(f1, f2, 3, padding=1)
        self.conv_2 = nn.Conv3d(f2, f3, 3, padding=1)
        self.conv_2_1 = nn.Conv3d(f3, f4, 3, padding=1)
        self.conv_3 = nn.Conv3d(f4, f4, 3, padding=1)
        self.conv_3_1 = nn.Conv3d(f4, f4, 3, padding=1)
        self.actvn = nn.ReLU()
        self.maxpool = nn.MaxPool3d(2)
        self.conv1_1_bn = nn.BatchNorm3d(f2)
        self.conv2_1_bn = nn.BatchNorm3d(f4)
        self.conv3_1_bn = nn.BatchNorm3d(f4)
        self._finish_init(32)


class IFNet(nn.Module):
    def __init__(self, hidden_dim=256, net_res=128):
        super().__init__()
        self.net_res = net_res
        if net_res == 128:
            self.ifnet_feature_extractor = IFNetFeatureExtractor128()
            feature_size = (1 + 16 + 32 + 64 + 128 + 128) * 7
            self.fc_0 = nn.Conv1d(feature_size, hidden_dim, 1)
            self.fc_1 = nn.Conv1d(hidden_dim, hidden_dim, 1)
            self.fc_2 = nn.Conv1d(hidden_dim, hidden_dim, 1)
        elif net_res == 32:
            self.ifnet_feature_extractor = IFNetFeatureExtractor(32, 64, 128, 128)
            feature_size = (1 + 64 + 128 + 128) * 7
            self.fc_0 = nn.Conv1d(feature_size, hidden_dim * 2, 1)
            self.fc_1 = nn.Conv1d(hidden_dim * 2, hidden_dim, 1)
            self.fc_2 = nn.Conv1d(hidden_dim, hidden_dim, 1)
        else:
            # the reference *returns* NotImplementedError here (model/ifnet.py:31-32); raise instead
            raise NotImplementedError(f"net_res={net_res}")
        self.fc_out = nn.Conv1d(hidden_dim, 1, 1)
        self.actvn = nn.ReLU()
        perm = self.ifnet_feature_extractor._layout.reference_permutation()
        self.register_buffer("_fc0_perm", perm, persistent=False)
        # gather index / 0-1 mask of the internal columns, and the inverse (reference feature k -> internal column)
        self.register_buffer("_fc0_src", perm.clamp(min=0), persistent=False)
        self.register_buffer("_fc0_mask", perm < 0, persistent=False)          # True on the padding columns
        inv = torch.empty(int((perm >= 0).sum()), dtype=torch.long)
        inv[perm[perm >= 0]] = torch.nonzero(perm >= 0).squeeze(1)
        self.register_buffer("_fc0_inv", inv, persistent=False)

    def _fc0_internal(self):
        """fc_0.weight (H, F, 1) with reference column order k = c*7+j -> (H, FS) in the internal
        column order (zero weight on padding columns).  The checkpoint layout never changes."""
        return _PermuteColumnsFn.apply(self.fc_0.weight.squeeze(2), self._fc0_src, self._fc0_mask, self._fc0_inv)

    @torch.no_grad()
    def _prepare_weights_async(self, lease):
        """Training step: the weight planes of every split-precision layer (encoder convolutions behind conv_in, fc_1,
        fc_2; forward and backward-data forms) are made on the side stream, in front of the Morton sort, instead of in
        2-4 launch-bound kernels in front of each layer call on the main stream (~40 launches per step).  Valid until
        the encoder's backward has been enqueued -- or the forward's graph is dropped without one (they expire with the
        arena lease) -- ops find them by the parameter's address + version (ops.PreparedWeights)."""
        if not PREPARE_WEIGHTS_AHEAD:
            return
        ext = self.ifnet_feature_extractor
        prep = ext._prepared
        main, side = torch.cuda.current_stream(), _get_side_stream(self.fc_out.weight.device)
        side.wait_stream(main)          # the optimizer step that produced these parameters
        with torch.cuda.stream(side):
            prep.begin()
            # forward planes first, with their own event: the forward pass waits for those only, the backward planes (as many
            # launches again) are made while it runs
            for which in (("fwd", "bwd") if PREPARE_FORWARD_FIRST else ("both",)):
                for convs, _ in ext._stages:
                    for conv in convs:
                        if conv.weight.shape[1] > 1:
                            prep.add_conv(conv.weight.detach(), which)
                prep.add_linear(self.fc_1.weight.detach().squeeze(2), which)
                prep.add_linear(self.fc_2.weight.detach().squeeze(2), which)
                if which == "fwd":
                    prep.mark(side)
            prep.finish(side)
        ops.set_prepared(prep)

        def expire():       # with the lease: at the end of the backward, or when the graph is dropped without one
            prep.invalidate()
            if ops._prepared is prep:
                ops.set_prepared(None)
        lease.on_release.append(expire)

    @torch.no_grad()
    def encode(self, x, storage="f32"):
        """Cache the feature pyramid of a grid for repeated queries (dense-grid inference).  storage="bf16": the
        throughput mode -- the pyramid (computed in f32) is stored in bf16 and query() runs the bf16-storage gather +
        point MLP (f32 accumulation; bf16_path.hip).  Never the default: bf16 logits are not held to the fp32 gate."""
        levels = self.ifnet_feature_extractor.encode_levels(x)
        if storage == "bf16":
            return [ops.cast_bf16(v) for v in levels]
        if storage != "f32":
            raise ValueError(f"storage={storage!r}")
        return levels

    @torch.no_grad()
    def _query_bf16(self, levels, points, row_map, prepared=None):
        B, N = points.shape[0], points.shape[1]
        ext = self.ifnet_feature_extractor
        if prepared is None and FUSE_FC0_BF16 and ops.gather_fc0_bf16_supported(levels, ext._layout, ext._disp, ext._align,
                                                                                self.fc_0.out_channels):
            prepared = ops.gather_fc0_bf16_prepare(levels, ext._layout, ext._disp, ext._align, self._fc0_internal())
        if prepared is not None:      # fused gather -> fc_0 on bf16 storage: the bf16 feature rows never reach HBM
            if any(a.data_ptr() != b.data_ptr() or a.shape != b.shape for a, b in zip(prepared["vols"], levels)):
                raise RuntimeError("IFNet.query: `prepared` was made for another pyramid than `levels` (prepare_query again)")
            h = ops.gather_fc0_bf16_run(prepared, points, self.fc_0.bias, relu=True)
        else:
            rows = ops.gather_fwd_bf16(levels, points, ext._layout, ext._disp, ext._align)
            h = ops.linear_fwd_bf16(rows, ops.cast_bf16(self._fc0_internal()), self.fc_0.bias, relu=True)
        h = ops.linear_fwd_bf16(h, ops.cast_bf16(self.fc_1.weight.squeeze(2).contiguous()), self.fc_1.bias, relu=True)
        h = ops.linear_fwd_bf16(h, ops.cast_bf16(self.fc_2.weight.squeeze(2).contiguous()), self.fc_2.bias, relu=True)
        return ops.fc_out_fwd_bf16(h, self.fc_out.weight.reshape(-1).contiguous(), self.fc_out.bias, row_map).view(B, N)

    @torch.no_grad()
    def prepare_query(self, levels, max_points):
        """Per-pyramid preparation of the f32 query path (fc_0's weights split for the fused kernel, slab table): pass the
        result to query(..., prepared=) when one pyramid is queried chunk by chunk (dense-grid inference); None when the
        fused kernel does not cover the shapes."""
        ext = self.ifnet_feature_extractor
        if levels[0].dtype == torch.bfloat16:
            if FUSE_FC0_BF16 and ops.gather_fc0_bf16_supported(levels, ext._layout, ext._disp, ext._align, self.fc_0.out_channels):
                return ops.gather_fc0_bf16_prepare(levels, ext._layout, ext._disp, ext._align, self._fc0_internal())
            return None
        if levels[0].dtype != torch.float32 or not FUSE_FC0 or self.fc_0.out_channels != 256:
            return None
        B = levels[0].shape[0]
        probe = torch.empty(B, 1, 3, device=levels[0].device)
        if not ops.gather_fc0_supported(levels, probe, ext._layout, ext._disp, ext._align, self.fc_0.out_channels):
            return None
        return ops.gather_fc0_prepare(levels, ext._layout, ext._disp, ext._align, self._fc0_internal(), B, int(max_points))

    @torch.no_grad()
    def query(self, levels, points, spatial_sort=False, prepared=None):
        """Logits (B,N) for `points` against a pyramid from encode() (f32, or bf16 storage).  spatial_sort: visit the
        points of every sample in Morton order (pays off for scattered points: the gather's corner reads become cache
        hits; a dense lattice is ordered already); results do not depend on it.  prepared: prepare_query(levels, ...)."""
        B, N = points.shape[0], points.shape[1]
        points = points.float().contiguous()
        row_map = None
        if spatial_sort and N > 1:
            row_map, points = ops.morton_order(points, want_sorted=True)
        if levels[0].dtype == torch.bfloat16:
            return self._query_bf16(levels, points, row_map, prepared)
        ext = self.ifnet_feature_extractor
        if prepared is not None:
            # the prepared descriptor holds ITS pyramid (slab table, volume pointers): a different `levels` would be ignored
            if len(prepared.vols) != len(levels) or any(a.data_ptr() != b.data_ptr() or a.shape != b.shape
                                                        for a, b in zip(prepared.vols, levels)):
                raise RuntimeError("IFNet.query: `prepared` was made for another pyramid than `levels` (prepare_query again)")
            h, _ = ops.gather_fc0_run(prepared, points, self.fc_0.bias)
        elif FUSE_FC0 and ops.gather_fc0_supported(levels, points, ext._layout, ext._disp, ext._align, self.fc_0.out_channels):
            h, _ = ops.gather_fc0_fwd(levels, points, ext._layout, ext._disp, ext._align, self._fc0_internal(), self.fc_0.bias)
        else:
            rows = ext.feature_rows_from_levels(levels, points)
            h = ops.linear_fwd(rows, self._fc0_internal(), self.fc_0.bias, relu=True)
        h = ops.linear_fwd(h, self.fc_1.weight.squeeze(2), self.fc_1.bias, relu=True)
        h = ops.linear_fwd(h, self.fc_2.weight.squeeze(2), self.fc_2.bias, relu=True)
        return ops.fc_out_fwd(h, self.fc_out.weight.reshape(-1).contiguous(), self.fc_out.bias, row_map).view(B, N)

    def forward(self, x, points, spatial_sort=True):
        """logits (B,N).  With spatial_sort the points of every sample are visited in Morton order: the whole
        gather -> MLP -> scatter chain runs on the permuted point set (rows are independent), and only the
        (B,N) logits are permuted back.  That makes the gather's reads L2-local and lets the backward scatter
        combine runs of samples that share corners (gather.hip); results do not depend on the order."""
        B, N = points.shape[0], points.shape[1]
        if not x.is_cuda:
            raise RuntimeError("IF-Net HIP path needs GPU tensors (no CPU fallback)")
        row_map = None
        ext = self.ifnet_feature_extractor
        # training step: the large buffers that cross streams come from the extractor's step arena (held until the
        # encoder's backward has been enqueued; a second forward in between gets ordinary allocations)
        lease = ext._arena.lease() if (USE_ARENA and torch.is_grad_enabled() and not points.requires_grad
                                       and not torch.cuda.is_current_stream_capturing()) else None
        arena = lease.arena if lease is not None else None
        ext._points_ready = None
        if spatial_sort and N > 1:
            pts = points.detach().float().contiguous()
            if arena is not None and SORT_ON_SIDE_STREAM and not os.environ.get("SVR_NO_SIDE_STREAM"):
                self._prepare_weights_async(lease)
                # the Morton sort (a chain of ~25 launch-bound radix-sort kernels, 0.2 ms) is only needed by the gather and
                # by the scatter plans: it runs on the side stream, in front of the plans, while the encoder starts on the
                # main stream at once; the main stream waits for it in front of the gather (_EncoderGatherFn.forward).
                # (Arena buffers only: they never return to an allocator pool, so no record_stream bookkeeping.)
                main, side = torch.cuda.current_stream(), _get_side_stream(x.device)
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    row_map, sorted_pts = ops.morton_order(pts, want_sorted=True, arena=arena)
                    ext._points_ready = torch.cuda.Event()
                    ext._points_ready.record(side)
            else:
                row_map, sorted_pts = ops.morton_order(pts, want_sorted=True, arena=arena)     # samples stay contiguous
            if points.requires_grad:       # rare (subsample_points > 0): keep the permutation differentiable
                points = points.reshape(B * N, 3)[row_map.long()].view(B, N, 3)
            else:
                points = sorted_pts
        w0p = self._fc0_internal()
        link = None
        if (PROJECT_WIDE_LEVELS and torch.is_grad_enabled() and w0p.requires_grad and not points.requires_grad
                and self.fc_0.out_channels == 256 and SCATTER_FORM != "atomic"
                and not (ops.GATHER_FLAGS & ops._lib.GATHER_DETERMINISTIC)):
            D, H, W = x.shape[2:]
            wide = [l for l, c in enumerate(ext._layout.channels) if c == 128 and
                    ops.project_bwd_supported(B, N, (max(D >> (l - 1), 1), max(H >> (l - 1), 1), max(W >> (l - 1), 1)))]
            if wide and B * N * ext._layout.row_stride < 2 ** 31:
                link = _ProjLink(wide, ext._layout)
        if (FUSE_FC0 and not points.requires_grad and B * N * ext._layout.row_stride < 2 ** 31
                and _fc0_fusable(ext._layout.channels, B, x.shape[2:], self.fc_0.out_channels)):
            h0 = ext.feature_rows(x, points, link, w0p, self.fc_0.bias, lease=lease)
            logits = _PointMLPFn.apply(h0, row_map, None, None,
                                       self.fc_1.weight.squeeze(2), self.fc_1.bias,
                                       self.fc_2.weight.squeeze(2), self.fc_2.bias,
                                       self.fc_out.weight.reshape(-1), self.fc_out.bias, None)
            return logits.view(B, N)
        rows = ext.feature_rows(x, points, link, w0p if link is not None else None, lease=lease)
        logits = _PointMLPFn.apply(rows, row_map, w0p, self.fc_0.bias,
                                   self.fc_1.weight.squeeze(2), self.fc_1.bias,
                                   self.fc_2.weight.squeeze(2), self.fc_2.bias,
                                   self.fc_out.weight.reshape(-1), self.fc_out.bias, link)
        return logits.view(B, N)


def make_3d_grid(bb_min, bb_max, shape, res_increase=1):
    """Lattice of query points, C-order with the last axis fastest (model/ifnet.py:202-212)."""
    shape = [int(s) for s in shape]
    size = shape[0] * shape[1] * shape[2] * res_increase ** 3
    full = [s * res_increase for s in shape]
    pxs = torch.linspace(bb_min[0], bb_max[0], full[0]).view(-1, 1, 1).expand(*full).contiguous().view(size)
    pys = torch.linspace(bb_min[1], bb_max[1], full[1]).view(1, -1, 1).expand(*full).contiguous().view(size)
    pzs = torch.linspace(bb_min[2], bb_max[2], full[2]).view(1, 1, -1).expand(*full).contiguous().view(size)
    return torch.stack([pxs, pys, pzs], dim=1)


def evaluate_network_on_grid(network, x, resolution, res_increase=1, points_batch_size=2048 * 16, storage="f32"):
    """Occupancy probabilities on the dense lattice (model/ifnet.py:215-229).

    Same result as the reference loop, but the encoder pyramid is computed once (network.encode) instead of
    once per chunk, the lattice is built on the device, and the values stay on the device until the single
    D2H copy at the end (the reference does one per chunk, :226).  `network` in eval() mode reproduces the
    reference's validation call; any module without encode()/query() falls back to network(x, chunk)."""
    pointsf = make_3d_grid((-0.5,) * 3, (0.5,) * 3, resolution, res_increase).to(x.device)
    values = []
    with torch.no_grad():
        levels = (network.encode(x, storage) if storage != "f32" else network.encode(x)) if hasattr(network, "encode") else None
        prep = network.prepare_query(levels, points_batch_size) if (levels is not None and hasattr(network, "prepare_query")) else None
        for pi in torch.split(pointsf, points_batch_size):
            pi = pi.unsqueeze(0)
            z = (network.query(levels, pi, prepared=prep) if prep is not None else network.query(levels, pi)) \
                if levels is not None else network(x, pi)
            values.append(torch.sigmoid(z).squeeze(0))
    value = torch.cat(values, dim=0).cpu().numpy()
    r = [int(s) * res_increase for s in resolution]
    return value.reshape(r[0], r[1], r[2])
