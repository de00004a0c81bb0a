"""Mirror of the reference's data_processing/mesh_occupancies.py:24-53 (determine_occupancy: the labelling that runs
inside SceneNetTrainer.training_step when subsample_points > 0, trainer/trainer_scene_net.py:112,128).

``determine_occupancy(mesh_path, points, dims)``: `mesh_path` is the batch's list of .obj paths (dataset/
scene_net_data.py:87,97) -- or already loaded meshes (anything with .vertices / .faces); `points` (B,M,3) stays on the
device.  The reference does a D2H copy, a trimesh.load and a Cython hash build per sample and step; here meshes are
parsed once (small LRU cache keyed by path + mtime), hashed by the library's C++ host code, and tested in one kernel.

Reference quirk, preserved by default (`reference_quirk=True`): lines :29-31 ASSIGN dims[i] to the point coordinates
instead of dividing by it, so every query point becomes (dims[0], dims[1], dims[2]) -- outside every normalised mesh --
and the returned occupancies are all zero.  `reference_quirk=False` applies the normalisation the comment describes
(points - dims/2, / dims)."""
import functools
import os
from types import SimpleNamespace

import numpy as np
import torch

from .implicit_waterproofing import implicit_waterproofing


def load_obj(path):
    """Minimal Wavefront .obj reader: `v x y z` and `f a b c ...` (1-based, negative = relative, a/b/c forms, polygons
    fanned into triangles) -> Mesh(vertices float64 (V,3), faces int32 (F,3)).  What determine_occupancy needs of
    trimesh.load(path): the triangle soup (vertex merging does not change containment)."""
    verts, faces = [], []
    with open(path, "r") as fh:
        for line in fh:
            if line.startswith("v "):
                p = line.split()
                verts.append((float(p[1]), float(p[2]), float(p[3])))
            elif line.startswith("f "):
                idx = []
                for tok in line.split()[1:]:
                    i = int(tok.split("/")[0])
                    idx.append(i - 1 if i > 0 else len(verts) + i)
                for k in range(1, len(idx) - 1):
                    faces.append((idx[0], idx[k], idx[k + 1]))
    return SimpleNamespace(vertices=np.asarray(verts, dtype=np.float64).reshape(-1, 3),
                           faces=np.asarray(faces, dtype=np.int32).reshape(-1, 3))


@functools.lru_cache(maxsize=64)
def _load_cached(path, mtime):
    return load_obj(path)


def _as_mesh(m):
    if isinstance(m, (str, os.PathLike)):
        p = os.fspath(m)
        return _load_cached(p, os.path.getmtime(p))
    if isinstance(m, (tuple, list)) and len(m) == 2:
        return SimpleNamespace(vertices=np.asarray(m[0]), faces=np.asarray(m[1]))
    return m


def determine_occupancy(mesh_path, points, dims=(139, 104, 112), reference_quirk=True, points_normalized=False):
    """-> (points as the reference returns them, occupancies (B,M) float32), both on the device of `points`.
    reference_quirk=False: points are normalised as the reference's comment intends ((p - dims/2) / dims), or taken as
    they are when `points_normalized` (the trainer passes the already normalised point cloud)."""
    if not torch.is_tensor(points):
        points = torch.from_numpy(np.asarray(points)).cuda()
    if not points.is_cuda:
        raise RuntimeError("determine_occupancy HIP path needs GPU tensors (no CPU fallback)")
    pts = points.detach().clone()
    if reference_quirk or not points_normalized:
        for a in range(3):
            pts[:, :, a] -= (dims[a] / 2)
        for a in range(3):
            if reference_quirk:
                pts[:, :, a] = dims[a]
            else:
                pts[:, :, a] /= dims[a]
    size = np.array(dims)
    occs = torch.zeros(len(mesh_path), pts.shape[1], device=pts.device, dtype=torch.float32)
    for i, m in enumerate(mesh_path):
        mesh = _as_mesh(m)
        # mesh.apply_translation(-size / 2); mesh.apply_scale(1 / size)   (:40-43)
        norm = SimpleNamespace(vertices=(np.asarray(mesh.vertices, dtype=np.float64) + (-size / 2)) * (1 / size),
                               faces=mesh.faces)
        occs[i] = implicit_waterproofing(norm, pts[i])[0].float()
    return pts, occs
