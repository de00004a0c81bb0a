"""check_mesh_contains on MI355X: mirror of the reference's data_processing/libmesh/inside_mesh.py:5-110.

Same call surface -- ``check_mesh_contains(mesh, points, hash_resolution=512) -> (contains, hole_points)`` and
``MeshIntersector(mesh, resolution).query(points)``, where ``mesh`` is anything with ``.vertices`` (V,3) and ``.faces``
(F,3) (a trimesh.Trimesh, or the ``Mesh`` of ..mesh_occupancies) -- with two differences: the 2-D triangle hash is built by
the library's C++ host code (the reference's Cython TriangleHash, triangle_hash.pyx) and uploaded once per mesh, and the
query runs in ONE HIP kernel with the points staying on the device.  ``points``: a CUDA tensor (float32 or float64;
results are CUDA bool tensors) or a numpy array (copied to the device; results are numpy bool arrays, like the
reference).  float64 arithmetic in the reference's operation order: the booleans match bit for bit."""
import ctypes as C

import numpy as np
import torch

from ... import _lib
from ..._lib import check


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class MeshIntersector:
    def __init__(self, mesh, resolution=512, device=None):
        verts = np.ascontiguousarray(np.asarray(mesh.vertices), dtype=np.float64)   # .astype(np.float64): exact for f32
        faces = np.ascontiguousarray(np.asarray(mesh.faces), dtype=np.int32)
        if verts.ndim != 2 or verts.shape[1] != 3 or faces.ndim != 2 or faces.shape[1] != 3 or len(faces) == 0:
            raise ValueError(f"mesh: vertices {verts.shape}, faces {faces.shape}")
        l = _lib.lib()
        self.resolution = int(resolution)
        st = np.zeros(6, dtype=np.float64)
        vp, fp = verts.ctypes.data_as(C.c_void_p), faces.ctypes.data_as(C.c_void_p)
        entries = l.svr_mesh_hash_entries(vp, len(verts), fp, len(faces), self.resolution, st.ctypes.data_as(C.c_void_p))
        if entries < 0:
            check(int(entries), "mesh_hash_entries")
        tri = np.empty((len(faces), 3, 3), dtype=np.float64)
        cell_start = np.empty(self.resolution * self.resolution + 1, dtype=np.int32)
        tri_ids = np.empty(max(int(entries), 1), dtype=np.int32)
        check(l.svr_mesh_hash_build(vp, len(verts), fp, len(faces), self.resolution, tri.ctypes.data_as(C.c_void_p),
                                    cell_start.ctypes.data_as(C.c_void_p), tri_ids.ctypes.data_as(C.c_void_p), int(entries)),
              "mesh_hash_build")
        self.scale, self.translate = st[:3].copy(), st[3:].copy()
        self._st = st
        dev = torch.device(device if device is not None else "cuda")
        self._tri = torch.from_numpy(tri).to(dev)
        self._cell_start = torch.from_numpy(cell_start).to(dev)
        self._tri_ids = torch.from_numpy(tri_ids).to(dev)
        self.n_entries = int(entries)

    def query(self, points):
        as_numpy = not torch.is_tensor(points)
        if as_numpy:
            arr = np.ascontiguousarray(points)
            if arr.dtype not in (np.float32, np.float64):
                arr = arr.astype(np.float64)
            points = torch.from_numpy(arr).to(self._tri.device)
        if not points.is_cuda:
            raise RuntimeError("check_mesh_contains HIP path needs GPU tensors (no CPU fallback)")
        if points.dtype not in (torch.float32, torch.float64):
            points = points.double()
        pts = points.reshape(-1, 3).contiguous()
        n = pts.shape[0]
        contains = torch.empty(n, device=pts.device, dtype=torch.uint8)
        holes = torch.empty(n, device=pts.device, dtype=torch.uint8)
        check(_lib.lib().svr_mesh_contains(C.c_void_p(pts.data_ptr()), int(pts.dtype == torch.float64), n,
                                           C.c_void_p(self._tri.data_ptr()), C.c_void_p(self._cell_start.data_ptr()),
                                           C.c_void_p(self._tri_ids.data_ptr()), self.resolution,
                                           self._st.ctypes.data_as(C.c_void_p), C.c_void_p(contains.data_ptr()),
                                           C.c_void_p(holes.data_ptr()), _stream()), "mesh_contains")
        contains, holes = contains.bool(), holes.bool()
        if as_numpy:
            return contains.cpu().numpy(), holes.cpu().numpy()
        return contains, holes


def check_mesh_contains(mesh, points, hash_resolution=512):
    intersector = MeshIntersector(mesh, hash_resolution, device=points.device if torch.is_tensor(points) else None)
    return intersector.query(points)
