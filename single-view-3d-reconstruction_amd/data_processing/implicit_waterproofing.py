"""Mirror of the reference's data_processing/implicit_waterproofing.py:6-48: points whose two z-ray directions disagree
(holes in the mesh) are re-tested against the mesh rotated by 90 degrees about y, x and z in turn.  The containment tests
run on the device (libmesh.inside_mesh); only the `holes.any()` flag crosses to the host per round."""
import math

import numpy as np
import torch

from .libmesh.inside_mesh import check_mesh_contains


def to_rotation_matrix(euler_angles):
    a = euler_angles
    R_x = np.array([[1, 0, 0], [0, math.cos(a[0]), -math.sin(a[0])], [0, math.sin(a[0]), math.cos(a[0])]])
    R_y = np.array([[math.cos(a[1]), 0, math.sin(a[1])], [0, 1, 0], [-math.sin(a[1]), 0, math.cos(a[1])]])
    R_z = np.array([[math.cos(a[2]), -math.sin(a[2]), 0], [math.sin(a[2]), math.cos(a[2]), 0], [0, 0, 1]])
    return np.dot(R_z, np.dot(R_y, R_x))


class _Rotated:
    def __init__(self, mesh, r):
        self.vertices = np.dot(r, np.asarray(mesh.vertices, dtype=np.float64).T).T     # mesh.apply_transform(r)
        self.faces = mesh.faces


def implicit_waterproofing(mesh_source, query_points, hash_resolution=512):
    """-> (occupancies, holes) bool, numpy for numpy points / CUDA tensors for CUDA points."""
    as_numpy = not torch.is_tensor(query_points)
    pts = torch.from_numpy(np.ascontiguousarray(query_points)).cuda() if as_numpy else query_points
    pts = pts.reshape(-1, 3)
    occ, holes = check_mesh_contains(mesh_source, pts, hash_resolution)
    for euler in np.array([[0, np.pi / 2, 0], [np.pi / 2, 0, 0], [0, 0, np.pi / 2]]):
        if not bool(holes.any()):
            break
        r = to_rotation_matrix(euler)
        hp = pts[holes].double()
        rt = torch.from_numpy(r).to(hp.device)
        rot = torch.stack([rt[i, 0] * hp[:, 0] + rt[i, 1] * hp[:, 1] + rt[i, 2] * hp[:, 2] for i in range(3)], dim=1)
        occ_rot, holes_rot = check_mesh_contains(_Rotated(mesh_source, r), rot, hash_resolution)
        occ = occ.clone()
        occ[holes] = occ_rot
        upd = torch.zeros_like(holes)
        upd[holes] = holes_rot
        holes = upd
    if as_numpy:
        return occ.cpu().numpy(), holes.cpu().numpy()
    return occ, holes
