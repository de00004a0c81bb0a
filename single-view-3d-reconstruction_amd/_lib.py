"""ctypes binding of libsvr_hip.so (the C ABI declared in include/svr_hip.h).

There is NO fallback: if the library is missing this raises, and every op raises on a
non-GPU tensor.  (The CPU oracle lives under oracle/ and is test infrastructure only.)
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libsvr_hip.so")

SVR_MAX_LEVELS = 6
EPI_NONE, EPI_BIAS, EPI_BIAS_RELU, EPI_MASK = 0, 1, 2, 3

P = C.c_void_p
I32, I64, F32 = C.c_int32, C.c_int64, C.c_float


class Conv2dDesc(C.Structure):
    _fields_ = [("src0", P), ("src1", P), ("B", I32), ("H", I32), ("W", I32), ("C0", I32), ("C1", I32), ("k", I32),
                ("stride", I32), ("act", I32), ("upsample", I32)]


class PullPlan(C.Structure):
    _fields_ = [("keys", P), ("recs", P), ("heads", P), ("n_items", I64)]


class Level(C.Structure):
    _fields_ = [("vol", P), ("gvol", P), ("C", I32), ("D", I32), ("H", I32), ("W", I32), ("col", I32), ("order", P),
                ("item_order", P), ("plan", C.POINTER(PullPlan))]


class GatherDesc(C.Structure):
    _fields_ = [("n_levels", I32), ("B", I32), ("N", I32), ("row_stride", I32), ("align_corners", I32),
                ("displacement", F32), ("order", P), ("level", Level * SVR_MAX_LEVELS), ("flags", I32)]


GATHER_WIDE_OFFSETS, GATHER_DETERMINISTIC = 1, 2


# name -> (restype, argtypes): exactly the declarations of include/svr_hip.h
SIGNATURES = {
    "svr_version": (C.c_int, []),
    "svr_last_error": (C.c_char_p, []),
    "svr_sizeof_level": (I64, []),
    "svr_sizeof_gather_desc": (I64, []),
    "svr_points_morton_order_workspace": (I64, [I32, I32]),
    "svr_points_morton_order": (C.c_int, [P, P, P, I32, I32, P, P]),
    "svr_points_voxel_order": (C.c_int, [P, P, I32, I32, I32, I32, I32, I32, P, P]),
    "svr_gather_item_order": (C.c_int, [P, I32, I32, I32, I32, I32, I32, F32, I32, P, P, P]),
    "svr_gather_project_bwd": (C.c_int, [P, P, I64, I32, I32, I32, I32, I32, I32, F32, P, P, P]),
    "svr_gather_project_plan_workspace": (C.c_int64, [I32, I32]),
    "svr_gather_project_slots": (C.c_int64, [I32, I32, I32, I32, I32]),
    "svr_gather_project_plan": (C.c_int, [P, I32, I32, I32, I32, I32, I32, F32, P, P, P, P, P, P]),
    "svr_gather_project_bwd2": (C.c_int, [P, P, I64, I32, I32, I32, I32, I32, I32, F32, P, P, P, P, P, P]),
    "svr_gather_pull_plan_workspace": (I64, [I32, I32]),
    "svr_gather_pull_plan_workspace_cells": (I64, [I32, I32, I32, I32]),
    "svr_gather_pull_plan": (C.c_int, [P, I32, I32, I32, I32, I32, I32, I32, I32, I32, F32, P, P, P, P, P, P, P]),
    "svr_gather_trilinear_fwd": (C.c_int, [C.POINTER(GatherDesc), P, P, P]),
    "svr_gather_fc0_supported": (C.c_int32, [C.POINTER(GatherDesc)]),
    "svr_gather_fc0_workspace": (C.c_int64, [C.POINTER(GatherDesc), I32]),
    "svr_gather_fc0_fwd": (C.c_int, [C.POINTER(GatherDesc), P, P, I64, P, P, I64, I32, P, I64, P, C.c_uint32, I32, P, P]),
    "svr_gather_fc0_prepare": (C.c_int, [C.POINTER(GatherDesc), P, I64, I32, P, I64, P, C.c_uint32, P, P]),
    "svr_gather_fc0_run": (C.c_int, [C.POINTER(GatherDesc), P, P, P, I64, I32, P, I64, P, C.c_uint32, I32, P, P]),
    "svr_gather_fc0_bf16_prepare": (C.c_int, [C.POINTER(GatherDesc), P, I64, I32, P, P]),
    "svr_gather_fc0_bf16_run": (C.c_int, [C.POINTER(GatherDesc), P, P, P, I64, I32, I32, P, P]),
    "svr_gather_trilinear_bwd": (C.c_int, [C.POINTER(GatherDesc), P, P, P, P]),
    "svr_gather_corner_indices": (C.c_int, [C.POINTER(GatherDesc), I32, P, P, P]),
    "svr_cast_f32_to_bf16": (C.c_int, [P, P, I64, P]),
    "svr_gather_trilinear_fwd_bf16": (C.c_int, [C.POINTER(GatherDesc), P, P, P]),
    "svr_linear_fwd_bf16": (C.c_int, [P, I64, P, I64, P, P, I64, I64, I64, I64, C.c_int, P]),
    "svr_fc_out_fwd_bf16": (C.c_int, [P, I64, P, P, P, P, I64, I64, P]),
    "svr_linear_fwd": (C.c_int, [P, I64, P, I64, P, P, I64, I64, I64, I64, C.c_int, P, I64, P]),
    "svr_linear_bwd_data": (C.c_int, [P, I64, P, I64, P, I64, I64, I64, I64, C.c_int, P, I64, P]),
    "svr_linear_bwd_weight_workspace": (I64, [I64, I64, I64]),
    "svr_linear_bwd_weight": (C.c_int, [P, I64, P, I64, P, I64, P, I64, I64, I64, P, P]),
    "svr_linear_fwd_bf16x6_workspace": (I64, [I64, I64]),
    "svr_linear_fwd_bf16x6": (C.c_int, [P, I64, P, I64, P, P, I64, I64, I64, I64, C.c_int, P, P]),
    "svr_linear_fwd_f16x3_workspace": (I64, [I64, I64]),
    "svr_linear_fwd_f16x3": (C.c_int, [P, I64, P, I64, P, P, I64, I64, I64, I64, C.c_int, P, P]),
    "svr_linear_bwd_data_bf16x3_workspace": (I64, [I64, I64]),
    "svr_linear_bwd_data_bf16x3": (C.c_int, [P, I64, P, I64, P, I64, I64, I64, I64, C.c_int, P, I64, P, P]),
    "svr_linear_bwd_weight_bf16x3_workspace": (I64, [I64, I64, I64]),
    "svr_linear_bwd_weight_bf16x3": (C.c_int, [P, I64, P, I64, P, I64, P, I64, I64, I64, P, P]),
    "svr_amax_f32": (C.c_int, [P, I64, I64, I64, P, P]),
    "svr_linear_bwd_data_f16x3_workspace": (I64, [I64, I64]),
    "svr_linear_bwd_data_f16x3": (C.c_int, [P, I64, P, I64, P, I64, I64, I64, I64, C.c_int, P, I64, P, P, P, P]),
    "svr_linear_bwd_weight_f16x3_workspace": (I64, [I64, I64, I64]),
    "svr_linear_bwd_weight_f16x3": (C.c_int, [P, I64, P, I64, P, I64, P, I64, I64, I64, P, P, P]),
    "svr_fc_out_fwd": (C.c_int, [P, I64, P, P, P, P, I64, I64, P]),
    "svr_fc_out_bwd_workspace": (I64, [I64, I64]),
    "svr_fc_out_bwd": (C.c_int, [P, I64, P, P, P, P, I64, P, P, I64, I64, P, P, P]),
    "svr_bce_logits_sum_mean": (C.c_int, [P, P, P, P, I64, I64, F32, P, P]),
    "svr_conv3d_pack_weight": (C.c_int, [P, P, P, I32, I32, P]),
    "svr_conv3d_unpack_wgrad": (C.c_int, [P, P, I32, I32, P]),
    "svr_conv3d_k3": (C.c_int, [P, P, P, P, I32, I32, I32, I32, I32, I32, C.c_int, P, P]),
    "svr_conv3d_fwd_bf16x6_workspace": (I64, [I32, I32]),
    "svr_conv3d_k3_fwd_bf16x6": (C.c_int, [P, P, P, P, I32, I32, I32, I32, I32, I32, C.c_int, P, P]),
    "svr_conv3d_fwd_f16x3_workspace": (I64, [I32, I32]),
    "svr_conv3d_k3_fwd_f16x3": (C.c_int, [P, P, P, P, I32, I32, I32, I32, I32, I32, C.c_int, P, P]),
    "svr_conv3d_fwd_f16x3_stats_blocks": (I32, [I32, I32, I32, I32, I32, I32]),
    "svr_conv3d_k3_fwd_f16x3_stats": (C.c_int, [P, P, P, P, P, I32, I32, I32, I32, I32, I32, C.c_int, P, P]),
    "svr_conv3d_bwd_data_bf16x3_workspace": (I64, [I32, I32]),
    "svr_conv3d_k3_bwd_data_bf16x3": (C.c_int, [P, P, P, I32, I32, I32, I32, I32, I32, C.c_int, P, P, P]),
    "svr_conv3d_c1_fwd_stats_workspace": (I64, [I32, I32, I32, I32, I32]),
    "svr_conv3d_c1_fwd_stats": (C.c_int, [P, P, P, P, P, I32, I32, I32, I32, I32, C.c_int, P, P]),
    "svr_conv3d_k3_bwd_weight_workspace": (I64, [I32, I32, I32, I32, I32, I32]),
    "svr_conv3d_k3_bwd_weight": (C.c_int, [P, P, P, P, I32, I32, I32, I32, I32, I32, P, P]),
    "svr_conv3d_k3_bwd_weight_bf16x3_workspace": (I64, [I32, I32, I32, I32, I32, I32]),
    "svr_conv3d_k3_bwd_weight_bf16x3": (C.c_int, [P, P, P, P, I32, I32, I32, I32, I32, I32, P, P]),
    "svr_conv3d_k3_bwd_weight_bf16x3_param": (C.c_int, [P, P, P, P, I32, I32, I32, I32, I32, I32, P, P]),
    "svr_conv3d_bwd_data_f16x3_workspace": (I64, [I32, I32]),
    "svr_conv3d_k3_bwd_data_f16x3": (C.c_int, [P, P, P, I32, I32, I32, I32, I32, I32, C.c_int, P, P, P, P, P]),
    "svr_conv3d_k3_bwd_weight_f16x3": (C.c_int, [P, P, P, P, I32, I32, I32, I32, I32, I32, I32, P, P, P]),
    "svr_bn_stats_workspace": (I64, [I64, I32]),
    "svr_bn_stats": (C.c_int, [P, P, I64, I32, P, P]),
    "svr_bn_stats_finalize": (C.c_int, [P, P, P, P, P, P, P, P, I64, I32, F32, F32, P, P]),
    "svr_bn_finalize_parts": (C.c_int, [P, I32, P, P, P, P, P, P, P, I64, I32, F32, F32, P]),
    "svr_bn_finalize": (C.c_int, [P, P, P, P, P, P, P, I64, I32, F32, F32, C.c_int, P]),
    "svr_bn_apply_pool": (C.c_int, [P, P, P, P, P, I32, I32, I32, I32, I32, P]),
    "svr_bn_bwd_reduce": (C.c_int, [P, P, P, P, P, P, P, I32, I32, I32, I32, I32, P, P]),
    "svr_bn_bwd_apply": (C.c_int, [P, P, P, P, P, P, P, P, P, P, P, I32, I32, I32, I32, I32, C.c_int, P, P]),
    "svr_stage1_supported": (I32, [I32, I32, I32, I32, I32]),
    "svr_stage1_workspace": (I64, [I32, I32, I32, I32]),
    "svr_stage1_fwd": (C.c_int, [P, P, P, P, P, P, P, P, P, P, P, P, P, I32, I32, I32, I32, I32, F32, F32, C.c_int, C.c_int, P, P]),
    "svr_stage1_bwd": (C.c_int, [P, P, P, P, P, P, P, P, P, P, P, P, P, P, I32, I32, I32, I32, I32, C.c_int, C.c_int, P, P]),
    "svr_conv2d_im2col": (C.c_int, [C.POINTER(Conv2dDesc), P, P]),
    "svr_conv2d_col2im": (C.c_int, [C.POINTER(Conv2dDesc), P, P, P, P, P]),
    "svr_conv2d_small_supported": (C.c_int, [I32, I32, I32]),
    "svr_conv2d_small_fwd": (C.c_int, [C.POINTER(Conv2dDesc), P, P, P, I32, P]),
    "svr_conv2d_small_bwd_data": (C.c_int, [C.POINTER(Conv2dDesc), P, P, I32, P, P]),
    "svr_conv2d_small_bwd_weight_workspace": (I64, [C.POINTER(Conv2dDesc), I32]),
    "svr_conv2d_small_bwd_weight": (C.c_int, [C.POINTER(Conv2dDesc), P, I32, P, P, P, P]),
    "svr_conv2d_planes_bytes": (I64, [I32, I32, I32]),
    "svr_conv2d_prepare": (C.c_int, [P, I32, I32, I32, I32, I32, P, P, P]),
    "svr_conv2d_prepare_many": (C.c_int, [I32, P, P, P, P, P, P, P, P, P]),
    "svr_conv2d_workspace_bytes": (I64, [C.POINTER(Conv2dDesc), I32]),
    "svr_conv2d_virtual": (C.c_int, [C.POINTER(Conv2dDesc), P, P]),
    "svr_conv2d_fwd": (C.c_int, [C.POINTER(Conv2dDesc), P, P, P, P, I32, P, P, P, P]),
    "svr_conv2d_bwd_data": (C.c_int, [C.POINTER(Conv2dDesc), P, P, P, P, I32, P, P, P]),
    "svr_conv2d_finish_bwd": (C.c_int, [C.POINTER(Conv2dDesc), P, P, P, P]),
    "svr_conv2d_bwd_weight_workspace": (I64, [C.POINTER(Conv2dDesc), I32]),
    "svr_conv2d_bwd_weight": (C.c_int, [C.POINTER(Conv2dDesc), P, P, I32, P, P, P, P]),
    "svr_mesh_hash_entries": (I64, [P, I64, P, I64, I32, P]),
    "svr_mesh_hash_build": (C.c_int, [P, I64, P, I64, I32, P, P, P, I64]),
    "svr_mesh_contains": (C.c_int, [P, I32, I64, P, P, P, I32, P, P, P, P]),
    "svr_df_dims": (C.c_int, [C.c_char_p, P]),
    "svr_df_read": (C.c_int, [C.c_char_p, P, I64]),
    "svr_npz_member_info": (C.c_int, [C.c_char_p, C.c_char_p, P, P, P, P]),
    "svr_npz_member_read": (C.c_int, [C.c_char_p, C.c_char_p, P, I64]),
    "svr_df_to_grid": (C.c_int, [P, P, I32, I32, I32, P]),
    "svr_cast_to_f32": (C.c_int, [P, I32, P, I64, P]),
    "svr_subsample_rows": (C.c_int, [P, I32, I64, I32, P, I64, P, P, P]),
    "svr_unproject_fwd": (C.c_int, [P, P, I32, I32, I32, C.POINTER(F32), C.c_int, P]),
    "svr_unproject_bwd": (C.c_int, [P, P, P, I32, I32, I32, C.POINTER(F32), C.c_int, P]),
    "svr_voxelize_splat_fwd": (C.c_int, [P, P, P, P, I32, I32, I32, I32, I32, P]),
    "svr_voxelize_splat_bwd": (C.c_int, [P, P, P, I32, I32, I32, I32, I32, P]),
    "svr_scale_clamp01_fwd": (C.c_int, [P, P, I64, F32, P]),
    "svr_scale_clamp01_bwd": (C.c_int, [P, P, P, I64, F32, P]),
    "svr_blur_axis_fwd": (C.c_int, [P, P, P, I32, I32, I32, I32, I32, I32, P]),
    "svr_blur_axis_bwd": (C.c_int, [P, P, P, P, P, I32, I32, I32, I32, I32, I32, P]),
}

_lib = None


def lib():
    """Load (once) and return the bound library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the HIP path.")
        # torch must load ITS HIP runtime first: libsvr_hip.so only names libamdhip64 by SONAME, and if the
        # system copy gets mapped before torch's, the process ends up with two runtimes (kernels launched by one
        # on memory owned by the other: "no ROCm-capable device" from rocPRIM).
        import torch  # noqa: F401
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)          # AttributeError here = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        # layout pin: the ctypes mirrors above against the structs the library was compiled with
        if l.svr_sizeof_level() != C.sizeof(Level) or l.svr_sizeof_gather_desc() != C.sizeof(GatherDesc):
            raise RuntimeError(f"svr_level / svr_gather_desc layout mismatch: library {l.svr_sizeof_level()} / "
                               f"{l.svr_sizeof_gather_desc()} bytes, binding {C.sizeof(Level)} / {C.sizeof(GatherDesc)}")
        _lib = l
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().svr_last_error()
        raise RuntimeError(f"libsvr_hip {what} failed (rc={rc}): {msg.decode(errors='replace') if msg else ''}")
