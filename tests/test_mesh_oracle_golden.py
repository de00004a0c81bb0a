"""CPU: the numpy restatement of the reference's occupancy labelling (oracle/mesh_oracle.py) against the outputs of the
reference's own inside_mesh.py (tests/golden/mesh_*.npz, made by oracle/gen_golden_mesh.py with the reference's
triangle_hash.pyx compiled into oracle/_ref/) -- bit for bit -- and, when that build is present, the CSR triangle hash
against the reference's Cython TriangleHash directly."""
import glob
import importlib.util
import os

import numpy as np
import pytest

from oracle import mesh_oracle as M

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["sphere", "torus", "openbox", "f32pts"]


def load(tag):
    z = np.load(os.path.join(GOLD, f"mesh_{tag}.npz"), allow_pickle=False)
    n = int(z["n"])
    return z, np.unpackbits(z["contains"])[:n].astype(bool), np.unpackbits(z["holes"])[:n].astype(bool)


@pytest.mark.parametrize("tag", CASES)
def test_oracle_matches_reference_outputs(tag):
    z, contains, holes = load(tag)
    c, h = M.check_mesh_contains(z["vertices"], z["faces"], z["points"], int(z["resolution"]))
    assert np.array_equal(c, contains) and np.array_equal(h, holes)
    if tag == "openbox":
        assert holes.any() and not contains.any()          # an open box: every interior point is a hole point
    else:
        assert contains.any() and not holes.any()


def test_triangle_hash_equals_reference_cython_build():
    so = glob.glob(os.path.join(os.path.dirname(GOLD), "..", "oracle", "_ref", "triangle_hash*.so"))
    if not so:
        pytest.skip("oracle/_ref/triangle_hash*.so not built (needs /root/reference: oracle/build_ref.py)")
    spec = importlib.util.spec_from_file_location("triangle_hash", so[0])          # PyInit_triangle_hash
    th = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(th)
    rng = np.random.default_rng(3)
    tri = rng.uniform(-2, 34, size=(300, 3, 2))                     # some triangles stick out of the 32 x 32 grid
    pts = rng.uniform(-3, 35, size=(2000, 2))
    ref_p, ref_t = th.TriangleHash(tri, 32).query(pts)
    got_p, got_t = M.TriangleHash(tri, 32).query(pts)
    assert np.array_equal(np.asarray(ref_p), got_p) and np.array_equal(np.asarray(ref_t), got_t)


def test_waterproofing_closes_the_holes_of_an_open_box():
    v, f = M.box((-0.31, -0.2, -0.27), (0.3, 0.22, 0.25), drop_faces=2)
    rng = np.random.default_rng(11)
    pts = rng.uniform(-0.45, 0.45, size=(2000, 3))
    occ0, holes0 = M.check_mesh_contains(v, f, pts, 128)
    occ, holes = M.implicit_waterproofing(v, f, pts, 128)
    inside = np.all((pts > [-0.31, -0.2, -0.27]) & (pts < [0.3, 0.22, 0.25]), axis=1)
    assert holes0.sum() > 100 and holes.sum() < holes0.sum()
    assert np.array_equal(occ[~holes], inside[~holes])              # resolved points agree with the analytic box


def test_determine_occupancy_quirk_and_fix():
    v, f = M.icosphere(2, 30.0, (139 / 2, 104 / 2, 112 / 2))        # a sphere in grid units
    rng = np.random.default_rng(5)
    pts = rng.uniform(0, 1, size=(2, 500, 3)) * [139, 104, 112]
    _, occ_q = M.determine_occupancy([(v, f), (v, f)], pts)
    assert occ_q.shape == (2, 500) and not occ_q.any()              # the reference's assignment quirk: all points -> dims
    p2, occ = M.determine_occupancy([(v, f), (v, f)], pts, reference_quirk=False)
    d = np.linalg.norm((pts - [139 / 2, 104 / 2, 112 / 2]), axis=-1)
    sure = np.abs(d - 30.0) > 1.0                                   # away from the faceted surface
    assert np.array_equal(occ[sure] > 0.5, (d < 30.0)[sure])
