"""Generate tests/golden/mesh_*.npz by running the REFERENCE's check_mesh_contains on CPU (build container only).

TEST INFRASTRUCTURE ONLY.  data_processing/libmesh/inside_mesh.py is imported from /root/reference unmodified; its
native dependency (triangle_hash.pyx) is the build in oracle/_ref/ (oracle/build_ref.py).  Two environment facts:
inside_mesh.py uses the alias np.bool, which numpy >= 1.24 no longer has, so `np.bool = bool` is set before the call;
and its `mesh` argument only needs .vertices / .faces, so no trimesh is involved.  Only inputs and outputs are stored.

    python oracle/gen_golden_mesh.py
"""
import importlib.util
import os
import sys
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import build_ref  # noqa: E402
from oracle import mesh_oracle as M  # noqa: E402

REF = os.environ.get("SVR_REFERENCE", "/root/reference")
OUT = os.path.join(REPO, "tests", "golden")


def load_reference():
    so = build_ref.build()
    assert so, "oracle/_ref/triangle_hash*.so missing and /root/reference absent"
    spec = importlib.util.spec_from_file_location("data_processing.libmesh.triangle_hash", so)
    th = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(th)
    for name in ("data_processing", "data_processing.libmesh"):
        pkg = types.ModuleType(name)
        pkg.__path__ = [os.path.join(REF, *name.split("."))]
        sys.modules[name] = pkg
    sys.modules["data_processing.libmesh.triangle_hash"] = th
    if not hasattr(np, "bool"):
        np.bool = bool                      # alias removed from numpy; the reference was written for numpy < 1.24
    sys.dont_write_bytecode = True
    spec = importlib.util.spec_from_file_location("data_processing.libmesh.inside_mesh",
                                                  os.path.join(REF, "data_processing", "libmesh", "inside_mesh.py"))
    im = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = im
    spec.loader.exec_module(im)
    return im


def cases():
    rng = np.random.default_rng(7)
    v, f = M.icosphere(2, 0.37, (0.05, -0.02, 0.01))
    yield "sphere", v, f, rng.uniform(-0.5, 0.5, size=(4000, 3)), 512
    v, f = M.torus(0.3, 0.11, 40, 16)
    yield "torus", v, f, rng.uniform(-0.5, 0.5, size=(4000, 3)) * [1.0, 1.0, 0.4], 256
    v, f = M.box((-0.31, -0.2, -0.27), (0.3, 0.22, 0.25), drop_faces=2)       # open box: hole points
    pts = rng.uniform(-0.45, 0.45, size=(3000, 3))
    pts[:200] = np.round(pts[:200], 1)                                          # points on cell / face boundaries
    yield "openbox", v, f, pts, 128
    v, f = M.icosphere(1, 0.4)
    f = np.concatenate([f, [[0, 0, 1], [2, 3, 3]]])                             # degenerate triangles (zero area)
    v32 = v.astype(np.float32)
    yield "f32pts", v32, f, rng.uniform(-0.6, 0.6, size=(2500, 3)).astype(np.float32), 64   # float32 inputs (trainer path)


def main():
    im = load_reference()
    os.makedirs(OUT, exist_ok=True)
    for tag, v, f, pts, res in cases():
        mesh = types.SimpleNamespace(vertices=v, faces=f)
        contains, holes = im.check_mesh_contains(mesh, pts, res)
        oc, oh = M.check_mesh_contains(v, f, pts, res)
        assert np.array_equal(oc, contains) and np.array_equal(oh, holes), tag     # the restatement, checked on the spot
        path = os.path.join(OUT, f"mesh_{tag}.npz")
        np.savez_compressed(path, vertices=v, faces=f.astype(np.int32), points=pts, resolution=np.int64(res),
                            contains=np.packbits(contains), holes=np.packbits(holes), n=np.int64(len(pts)))
        print(f"{path}: {len(f)} triangles, {len(pts)} points, inside {contains.mean():.3f}, holes {holes.mean():.4f}, "
              f"{os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
