"""Native readers for the sample wire formats: thin numpy-level wrappers over the C ABI's host
functions (svr_df_*, svr_npz_member_*: C++ + zlib inside libsvr_hip.so; no GPU involved) and the device-side helpers
(transpose / cast / row subset).  `out=` lets the caller pass a pinned buffer so the H2D copy can be asynchronous."""
import ctypes as C
import os

import numpy as np
import torch

from .. import _lib
from .._lib import check

_NP = {0: np.float32, 1: np.float64, 2: np.bool_, 3: np.uint8, 4: np.int32, 5: np.int64}
_CODE = {torch.float32: 0, torch.float64: 1, torch.bool: 2, torch.uint8: 3, torch.int32: 4, torch.int64: 5}


def _path(p):
    return os.fspath(p).encode()


def df_dims(path):
    dims = (C.c_int64 * 3)()
    check(_lib.lib().svr_df_dims(_path(path), dims), "df_dims")
    return tuple(int(d) for d in dims)


def df_read_payload(path, out=None):
    """The raw float32 payload (x fastest) of a .df file as a flat array; `out`: a flat float32 numpy array (e.g. the
    numpy view of a pinned tensor) of the right length."""
    X, Y, Z = df_dims(path)
    n = X * Y * Z
    if out is None:
        out = np.empty(n, dtype=np.float32)
    assert out.dtype == np.float32 and out.size == n and out.flags["C_CONTIGUOUS"]
    check(_lib.lib().svr_df_read(_path(path), out.ctypes.data_as(C.c_void_p), n), "df_read")
    return out, (X, Y, Z)


def npz_member_info(path, member):
    dt, nd, fo = C.c_int32(), C.c_int32(), C.c_int32()
    shape = (C.c_int64 * 8)()
    check(_lib.lib().svr_npz_member_info(_path(path), member.encode(), C.byref(dt), C.byref(nd), shape, C.byref(fo)),
          "npz_member_info")
    return _NP[dt.value], tuple(int(shape[i]) for i in range(nd.value)), bool(fo.value)


def npz_load(path, member, out=None):
    """np.load(path)[member] through the native reader (stored or deflated member)."""
    dtype, shape, fortran = npz_member_info(path, member)
    n = int(np.prod(shape, dtype=np.int64)) if shape else 1
    if out is None:
        out = np.empty(n, dtype=dtype)
    assert out.dtype == dtype and out.size == n and out.flags["C_CONTIGUOUS"]
    check(_lib.lib().svr_npz_member_read(_path(path), member.encode(), out.ctypes.data_as(C.c_void_p), out.nbytes),
          "npz_member_read")
    return out.reshape(shape, order="F" if fortran else "C")


# ---- device side -------------------------------------------------------------------------------------------------
def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def df_to_grid(payload, dims):
    """device float32 payload (x fastest) -> (X, Y, Z) C-order float32 tensor."""
    X, Y, Z = dims
    out = torch.empty(X, Y, Z, device=payload.device, dtype=torch.float32)
    check(_lib.lib().svr_df_to_grid(C.c_void_p(payload.data_ptr()), C.c_void_p(out.data_ptr()), X, Y, Z, _stream()), "df_to_grid")
    return out


def cast_to_f32(t):
    t = t.contiguous()
    out = torch.empty(t.shape, device=t.device, dtype=torch.float32)
    check(_lib.lib().svr_cast_to_f32(C.c_void_p(t.data_ptr()), _CODE[t.dtype], C.c_void_p(out.data_ptr()), t.numel(), _stream()),
          "cast_to_f32")
    return out


def subsample_rows(rows, idx):
    """out[i] = float32(rows[idx[i]]) for a (n_rows, cols) or (n_rows,) device tensor of float64 / float32 / bool."""
    rows = rows.contiguous()
    cols = 1 if rows.dim() == 1 else rows.shape[1]
    idx = idx.to(device=rows.device, dtype=torch.int64).contiguous()
    out = torch.empty((idx.numel(),) if rows.dim() == 1 else (idx.numel(), cols), device=rows.device, dtype=torch.float32)
    bad = torch.zeros(1, device=rows.device, dtype=torch.int32)
    check(_lib.lib().svr_subsample_rows(C.c_void_p(rows.data_ptr()), _CODE[rows.dtype], rows.shape[0], cols,
                                        C.c_void_p(idx.data_ptr()), idx.numel(), C.c_void_p(out.data_ptr()),
                                        C.c_void_p(bad.data_ptr()), _stream()), "subsample_rows")
    return out, bad
