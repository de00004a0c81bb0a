"""CPU oracle: on-the-fly occupancy labelling of query points against a triangle mesh, restated in numpy.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Follows, statement for statement:

* check_mesh_contains / MeshIntersector ............ data_processing/libmesh/inside_mesh.py:5-8, :11-110
  (rescale to [0.5, res-0.5]^3 :20-24,108-110; AABB cull :39-47; 2-D candidates + exact 2-D test :50;
   intersection depth :55-56,76-106; parity in both z directions :59-74)
* TriangleIntersector2d.query / check_triangles ..... inside_mesh.py:113-155
* TriangleHash (the reference's only native code) ... data_processing/libmesh/triangle_hash.pyx:8-85
  (cell = resolution * x + y, every cell of a triangle's integer bounding box, triangles in index order)
* implicit_waterproofing ............................. data_processing/implicit_waterproofing.py:6-48
* determine_occupancy (incl. its assignment quirk) ... data_processing/mesh_occupancies.py:24-53

All arithmetic is float64 with one rounding per operation, in the reference's operation order (numpy does not
contract multiply-add), so the boolean outputs can be compared bit for bit.

Parity: check_mesh_contains is PINNED by tests/golden/mesh_*.npz -- outputs of the reference's own inside_mesh.py run
with its triangle_hash.pyx compiled into oracle/_ref/ (oracle/build_ref.py, oracle/gen_golden_mesh.py).
implicit_waterproofing / determine_occupancy call trimesh (mesh.copy / apply_transform / trimesh.load), which is not
installed in this image: their mesh transform is restated as plain `vertices @ R.T` and that step is "parity unpinned".
"""
from __future__ import annotations

import math

import numpy as np


class TriangleHash:
    """triangle_hash.pyx:8-85 as a CSR table: start[cell] .. start[cell+1] index `tris` (ascending triangle index)."""

    def __init__(self, triangles2d: np.ndarray, resolution: int):
        r = int(resolution)
        self.resolution = r
        t = np.asarray(triangles2d, dtype=np.float64)
        lo = np.clip(t.min(axis=1).astype(np.int32), 0, r - 1)     # <int> min(...): truncation toward zero, then clamp
        hi = np.clip(t.max(axis=1).astype(np.int32), 0, r - 1)
        cells, tris = [], []
        for i in range(t.shape[0]):
            xs = np.arange(lo[i, 0], hi[i, 0] + 1)
            ys = np.arange(lo[i, 1], hi[i, 1] + 1)
            c = (r * xs[:, None] + ys[None, :]).reshape(-1)
            cells.append(c)
            tris.append(np.full(c.shape, i, dtype=np.int64))
        cells = np.concatenate(cells) if cells else np.zeros(0, np.int64)
        tris = np.concatenate(tris) if tris else np.zeros(0, np.int64)
        order = np.argsort(cells, kind="stable")                   # stable: triangles stay in index order inside a cell
        self.tris = tris[order]
        self.start = np.searchsorted(cells[order], np.arange(r * r + 1))

    def query(self, points2d: np.ndarray):
        p = np.asarray(points2d, dtype=np.float64)
        x = p[:, 0].astype(np.int64)                               # int(): truncation toward zero
        y = p[:, 1].astype(np.int64)
        ok = (0 <= x) & (x < self.resolution) & (0 <= y) & (y < self.resolution)
        cell = np.where(ok, self.resolution * x + y, 0)
        cnt = np.where(ok, self.start[cell + 1] - self.start[cell], 0)
        pidx = np.repeat(np.arange(p.shape[0]), cnt)
        offs = np.arange(int(cnt.sum())) - np.repeat(np.cumsum(cnt) - cnt, cnt)
        tidx = self.tris[np.repeat(self.start[cell], cnt) + offs]
        return pidx, tidx


def check_triangles(points: np.ndarray, triangles: np.ndarray) -> np.ndarray:
    """inside_mesh.py:130-155 (strict inequalities: points on an edge belong to no triangle)."""
    contains = np.zeros(points.shape[0], dtype=bool)
    A = triangles[:, :2] - triangles[:, 2:]
    A = A.transpose([0, 2, 1])
    y = points - triangles[:, 2]
    detA = A[:, 0, 0] * A[:, 1, 1] - A[:, 0, 1] * A[:, 1, 0]
    mask = (np.abs(detA) != 0.)
    A, y, detA = A[mask], y[mask], detA[mask]
    s_detA = np.sign(detA)
    abs_detA = np.abs(detA)
    u = (A[:, 1, 1] * y[:, 0] - A[:, 0, 1] * y[:, 1]) * s_detA
    v = (-A[:, 1, 0] * y[:, 0] + A[:, 0, 0] * y[:, 1]) * s_detA
    sum_uv = u + v
    contains[mask] = ((0 < u) & (u < abs_detA) & (0 < v) & (v < abs_detA) & (0 < sum_uv) & (sum_uv < abs_detA))
    return contains


def compute_intersection_depth(points: np.ndarray, triangles: np.ndarray):
    """inside_mesh.py:76-106: depth * |n_z| of the z ray through the point on the triangle's plane (NaN for n_z == 0)."""
    t1, t2, t3 = triangles[:, 0, :], triangles[:, 1, :], triangles[:, 2, :]
    v1 = t3 - t1
    v2 = t2 - t1
    normals = np.cross(v1, v2)
    alpha = np.sum(normals[:, :2] * (t1[:, :2] - points[:, :2]), axis=1)
    n_2 = normals[:, 2]
    s_n_2 = np.sign(n_2)
    abs_n_2 = np.abs(n_2)
    mask = (abs_n_2 != 0)
    depth = np.full(points.shape[0], np.nan)
    depth[mask] = t1[:, 2][mask] * abs_n_2[mask] + alpha[mask] * s_n_2[mask]
    return depth, abs_n_2


class MeshIntersector:
    def __init__(self, vertices, faces, resolution=512):
        triangles = np.asarray(vertices)[np.asarray(faces)].astype(np.float64)
        n_tri = triangles.shape[0]
        self.resolution = resolution
        self.bbox_min = triangles.reshape(3 * n_tri, 3).min(axis=0)
        self.bbox_max = triangles.reshape(3 * n_tri, 3).max(axis=0)
        self.scale = (resolution - 1) / (self.bbox_max - self.bbox_min)
        self.translate = 0.5 - self.scale * self.bbox_min
        self._triangles = self.rescale(triangles)
        self._tri2d = self._triangles[:, :, :2]
        self._hash = TriangleHash(self._tri2d, resolution)

    def rescale(self, array):
        return self.scale * array + self.translate

    def query(self, points):
        points = self.rescale(points)
        contains = np.zeros(len(points), dtype=bool)
        hole_points = np.zeros(len(points), dtype=bool)
        inside_aabb = np.all((0 <= points) & (points <= self.resolution), axis=1)
        if not inside_aabb.any():
            return contains, hole_points
        mask = inside_aabb
        points = points[mask]
        pidx, tidx = self._hash.query(points[:, :2])
        keep = check_triangles(points[pidx][:, :2], self._tri2d[tidx])
        pidx, tidx = pidx[keep], tidx[keep]
        tri = self._triangles[tidx]
        pint = points[pidx]
        depth, abs_n_2 = compute_intersection_depth(pint, tri)
        with np.errstate(invalid="ignore"):
            smaller = depth >= pint[:, 2] * abs_n_2
            bigger = depth < pint[:, 2] * abs_n_2
        n0 = np.bincount(pidx[smaller], minlength=points.shape[0])
        n1 = np.bincount(pidx[bigger], minlength=points.shape[0])
        c1 = (np.mod(n0, 2) == 1)
        c2 = (np.mod(n1, 2) == 1)
        contains[mask] = (c1 & c2)
        hole_points[mask] = np.logical_xor(c1, c2)
        return contains, hole_points


def check_mesh_contains(vertices, faces, points, hash_resolution=512):
    return MeshIntersector(vertices, faces, hash_resolution).query(np.asarray(points))


def to_rotation_matrix(euler_angles):
    """implicit_waterproofing.py:6-24."""
    a = euler_angles
    R_x = np.array([[1, 0, 0], [0, math.cos(a[0]), -math.sin(a[0])], [0, math.sin(a[0]), math.cos(a[0])]])
    R_y = np.array([[math.cos(a[1]), 0, math.sin(a[1])], [0, 1, 0], [-math.sin(a[1]), 0, math.cos(a[1])]])
    R_z = np.array([[math.cos(a[2]), -math.sin(a[2]), 0], [math.sin(a[2]), math.cos(a[2]), 0], [0, 0, 1]])
    return np.dot(R_z, np.dot(R_y, R_x))


ROTATIONS = np.array([[0, np.pi / 2, 0], [np.pi / 2, 0, 0], [0, 0, np.pi / 2]])


def implicit_waterproofing(vertices, faces, query_points, hash_resolution=512):
    """implicit_waterproofing.py:27-48: points whose two ray directions disagree (holes in the mesh) are re-tested
    against the mesh rotated by 90 degrees about y, x, z in turn.  (mesh.apply_transform restated as vertices @ R.T:
    parity unpinned, trimesh is not installed.)"""
    query_points = np.asarray(query_points)
    vertices = np.asarray(vertices, dtype=np.float64)
    occ, holes = check_mesh_contains(vertices, faces, query_points, hash_resolution)
    for euler in ROTATIONS:
        if not holes.any():
            break
        r = to_rotation_matrix(euler)
        v_rot = np.dot(r, vertices.T).T
        pts = np.dot(r, query_points[holes].T).T
        occ_rot, holes_rot = check_mesh_contains(v_rot, faces, pts, hash_resolution)
        occ[holes] = occ_rot
        upd = np.full(len(query_points), False)
        upd[holes] = holes_rot
        holes = upd
    return occ, holes


def determine_occupancy(meshes, points, dims=(139, 104, 112), reference_quirk=True, points_normalized=False):
    """mesh_occupancies.py:24-53.  `meshes`: one (vertices, faces) pair per batch sample, in grid units (what
    trimesh.load(path) would return); they are translated by -dims/2 and scaled by 1/dims (:40-43).  With
    `reference_quirk` the points are overwritten like the reference does (:29-31 ASSIGN dims[i] instead of dividing by
    it, so every point becomes (dims) -- outside every normalised mesh -- and all occupancies are 0);
    reference_quirk=False applies the normalisation the comment above those lines describes."""
    points = np.array(points, copy=True)
    if reference_quirk or not points_normalized:
        points[:, :, 0] -= (dims[0] / 2)
        points[:, :, 1] -= (dims[1] / 2)
        points[:, :, 2] -= (dims[2] / 2)
        if reference_quirk:
            points[:, :, 0] = dims[0]
            points[:, :, 1] = dims[1]
            points[:, :, 2] = dims[2]
        else:
            points[:, :, 0] /= dims[0]
            points[:, :, 1] /= dims[1]
            points[:, :, 2] /= dims[2]
    occs = np.zeros((len(meshes), points.shape[1]))
    size = np.array(dims)
    for i, (v, f) in enumerate(meshes):
        v = (np.asarray(v, dtype=np.float64) + (-size / 2)) * (1 / size)
        occs[i] = implicit_waterproofing(v, f, points[i])[0]
    return points, occs.astype(np.float32)


# ------------------------------------------------------------------------------------------------------------------
# synthetic meshes for fixtures and tests (closed, open, degenerate)
# ------------------------------------------------------------------------------------------------------------------
def icosphere(subdiv=2, radius=1.0, center=(0.0, 0.0, 0.0)):
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                  [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], dtype=np.float64)
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6],
                  [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5], [2, 4, 11], [6, 2, 10],
                  [8, 6, 7], [9, 8, 1]], dtype=np.int64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    for _ in range(subdiv):
        cache, nf, vl = {}, [], list(v)

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                m = vl[a] + vl[b]
                vl.append(m / np.linalg.norm(m))
                cache[key] = len(vl) - 1
            return cache[key]
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [[a, ab, ca], [b, bc, ab], [c, ca, bc], [ab, bc, ca]]
        v, f = np.array(vl), np.array(nf, dtype=np.int64)
    return v * radius + np.asarray(center, dtype=np.float64), f


def box(lo=(-1.0, -1.0, -1.0), hi=(1.0, 1.0, 1.0), drop_faces=0):
    lo, hi = np.asarray(lo, np.float64), np.asarray(hi, np.float64)
    v = np.array([[x, y, z] for x in (lo[0], hi[0]) for y in (lo[1], hi[1]) for z in (lo[2], hi[2])])
    f = np.array([[0, 1, 3], [0, 3, 2], [4, 6, 7], [4, 7, 5], [0, 4, 5], [0, 5, 1], [2, 3, 7], [2, 7, 6], [0, 2, 6], [0, 6, 4],
                  [1, 5, 7], [1, 7, 3]], dtype=np.int64)
    return v, f[: len(f) - drop_faces]            # drop_faces > 0: an open box (hole points appear)


def torus(R=1.0, r=0.35, nu=24, nv=12):
    u = np.linspace(0, 2 * np.pi, nu, endpoint=False)
    w = np.linspace(0, 2 * np.pi, nv, endpoint=False)
    U, W = np.meshgrid(u, w, indexing="ij")
    v = np.stack([(R + r * np.cos(W)) * np.cos(U), (R + r * np.cos(W)) * np.sin(U), r * np.sin(W)], -1).reshape(-1, 3)
    f = []
    for i in range(nu):
        for j in range(nv):
            a, b = i * nv + j, ((i + 1) % nu) * nv + j
            c, d = i * nv + (j + 1) % nv, ((i + 1) % nu) * nv + (j + 1) % nv
            f += [[a, b, d], [a, d, c]]
    return v, np.array(f, dtype=np.int64)
